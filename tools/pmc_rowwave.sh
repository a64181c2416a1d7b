#!/bin/bash
# usage: tools/pmc_rowwave.sh <tag> [bench args...]     (GPU box, from the repo root)
# What limits the sparse row-wave kernels: separate rocprofv3 --pmc passes (kernel-trace only, as the pool
# requires) of the default bench, one counter block per pass, summarised per kernel into gpurun_out/<tag>_rowwave_pmc.json
# (copy to profiles/). Counter names: /opt/rocm/share/rocprofiler-sdk/counter_defs.yaml (gfx950 entries).
export TMPDIR=/tmp
R=$PWD
tag="$1"; shift
out=$R/gpurun_out/pmc_rw_$tag; [ -n "$PASSES" ] || rm -rf $out; mkdir -p $out
CMD="python $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline --no-extras $*"
cd /tmp
# every pass under its own timeout: a counter set the hardware cannot schedule makes rocprofv3 abort and then HANG (round 3: 25 min lost)
pass() { n=$1; shift; if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $n "; then return; fi
         timeout -k 5 ${PASS_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$n -o p -- $CMD >$out/$n.log 2>&1 || echo "pass $n failed" >&2; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
pass sq3 SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
pass tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
pass tcp3 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_READ_sum TCP_GATE_EN2_sum
pass ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass td TD_TD_BUSY_sum TD_TC_STALL_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass derived MemUnitStalled OccupancyPercent
cd $R
python tools/pmc_rowwave_summary.py $out gpurun_out/${tag}_rowwave_pmc.json
