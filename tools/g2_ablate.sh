#!/bin/bash
# Ablation timings of conv_graph_split2_kernel (csrc/conv_gemm_split.hip built with -DSAR_G2_ABLATE): which phase costs what, without
# the perturbation of in-kernel stamps.   build here: tools/g2_ablate.sh build ; on the GPU box: tools/g2_ablate.sh run
set -e
cd "$(dirname "$0")/.."
C=skeleton-action-recognition_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/bin
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DSAR_G2_ABLATE $G2_DEFS -c $C/conv_gemm_split.hip -o tools/bin/g2_ablate.o
  OTHERS=$(ls $C/*.o | grep -v "/conv_gemm_split.o\|\.lds")
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libsar_g2_ablate.so tools/bin/g2_ablate.o $OTHERS
else
  for bits in ${G2_BITS:-0 1 2 4 6 8 16 32 38 46 63}; do
    echo "== SAR_G2_ABLATE_BITS=$bits (1 raw DMA, 2 convert, 4 virtual convert, 8 MFMA, 16 W DMA, 32 epilogue)"
    SAR_GRAPH_SPLIT2=${SAR_GRAPH_SPLIT2:-1} SAR_G2_ABLATE_BITS=$bits SAR_HIP_LIB=$PWD/tools/bin/libsar_g2_ablate.so python tools/kernel_bench.py --split f16x3a --only gcn_fwd,gcn_dgrad --layers ${G2_LAYERS:-2,6,9} --reps 7 2>&1 | grep "^L"
  done
fi
