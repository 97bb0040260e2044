#!/bin/bash
# Ablation timings of graph_wgrad_split_kernel (csrc/conv_wgrad_split.hip built with -DSAR_GW_ABLATE): which phase costs what.
#   build here: tools/gw_ablate.sh build ; on the GPU box: tools/gw_ablate.sh run
set -e
cd "$(dirname "$0")/.."
C=skeleton-action-recognition_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/bin
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DSAR_GW_ABLATE -c $C/conv_wgrad_split.hip -o tools/bin/gw_ablate.o
  OTHERS=$(ls $C/*.o | grep -v "/conv_wgrad_split.o\|\.lds")
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libsar_gw_ablate.so tools/bin/gw_ablate.o $OTHERS
else
  for bits in ${GW_BITS:-0 1 2 3 4 8 12 15 16 31}; do
    echo "== SAR_GW_ABLATE_BITS=$bits (1 gathered slices, 2 slice-0 images, 4 dout conversion, 8 MFMA, 16 slab stores)"
    SAR_GW_ABLATE_BITS=$bits SAR_HIP_LIB=$PWD/tools/bin/libsar_gw_ablate.so python tools/kernel_bench.py --split f16x3a --only gcn_wgrad --layers ${GW_LAYERS:-2,6,9} --reps 7 2>&1 | grep "^L"
  done
fi
