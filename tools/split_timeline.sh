#!/bin/bash
# Per-workgroup timelines of the split-arithmetic kernels (conv_gemm_split.hip built with -DSAR_SPLIT_TL).
#   build here: tools/split_timeline.sh build ;  on the GPU box: tools/split_timeline.sh run [g_fwd g_dgate t_fwd t_dgrad]
set -e
cd "$(dirname "$0")/.."
C=skeleton-action-recognition_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/bin
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DSAR_SPLIT_TL -c $C/conv_gemm_split.hip -o tools/bin/split_tl.o
  OTHERS=$(ls $C/*.o | grep -v "/conv_gemm_split.o\|\.lds")
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libsar_split_tl.so tools/bin/split_tl.o $OTHERS
else
  shift || true
  SAR_HIP_LIB=$PWD/tools/bin/libsar_split_tl.so python tools/split_timeline.py "$@" | grep -v "^/opt"
fi
