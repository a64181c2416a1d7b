#!/bin/bash
# usage: tools/pmc_any.sh <tag> <command...>   (GPU box, from the repo root)
# Three rocprofv3 --pmc passes (kernel-trace only, as the pool requires) around <command>, then a per-kernel table:
# duration, clock, MFMA-pipe busy, where the wave cycles go, LDS conflicts, VMEM, fabric-side FETCH/WRITE bytes.
export TMPDIR=/tmp
R=$PWD
tag="$1"; shift
out=$R/gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAIT_INST_LDS --output-format csv -d $out/a -o p -- "$@" >/dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU --output-format csv -d $out/b -o p -- "$@" >/dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/c -o p -- "$@" >/dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/d -o p -- "$@" >/dev/null 2>&1
cd $R
python tools/pmc_any_summary.py $out "${PMC_FILTER:-conv}" | tee $out/summary.txt
