#!/bin/bash
# Per-workgroup timelines of the fp32 MFMA conv-GEMM kernels (conv_gemm.hip built with -DSAR_FP32_TL).
#   build here: tools/fp32_timeline.sh build ;  on the GPU box: tools/fp32_timeline.sh run [g_fwd g_dgate t_fwd t_dgrad]
set -e
cd "$(dirname "$0")/.."
C=skeleton-action-recognition_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/bin
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSAR_FP32_TL -c $C/conv_gemm.hip -o tools/bin/fp32_tl.o
  OTHERS=$(ls $C/*.o | grep -v "/conv_gemm.o\|\.lds")
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libsar_fp32_tl.so tools/bin/fp32_tl.o $OTHERS
else
  shift || true
  SAR_HIP_LIB=$PWD/tools/bin/libsar_fp32_tl.so python tools/fp32_timeline.py "$@" | grep -v "^/opt"
fi
