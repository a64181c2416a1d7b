#!/bin/bash
# Diagnostic libraries tools/probe/libcpd_ablN.so: gather_conv.hip built with -DCPD_GC_ABLATE=N
# (row-wave split kernel: 1 no weight loads, 2 no row gathers, 4 no MFMAs, 8 no barriers; window kernel: 16 no border masks,
# 64 no weight stages, 128 no window loads/splits/stores, 256 fragment LDS reads in the first stage only, 512 no MFMAs, 1024 no epilogue;
# both: 2048 two of the three split-fp16 products; row-wave: 4096 no split (gathered bits used as fragments); sums combine),
# staged row-wave kernel (rowplan): 65536 no weight loads, 131072 no row loads, 262144 no fragment reads / MFMAs, 524288 no epilogue;
# rest of the library unchanged.
set -e
cd "$(dirname "$0")/../cpd_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/probe
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DCPD_GC_ABLATE=$n -c gather_conv.hip -o /tmp/gc_abl$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probe/libcpd_abl$n.so voxelize.o site_index.o row_plan.o /tmp/gc_abl$n.o decode.o iou3d_nms.o train_ops.o roi_pool.o atss.o
done
ls -la ../../tools/probe/*.so
