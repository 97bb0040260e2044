#!/bin/bash
# Timeline of ONE launch of the bf16 graph convolution (conv_graph_cn8.hip built with -DSAR_G2_TIMELINE): per workgroup start / end
# (100 MHz), CU id, cycles per phase -> workgroups resident per CU, lifetimes, dispatch rate.
#   build here: tools/g2_timeline.sh build [ablate mask] ;  on the GPU box: tools/g2_timeline.sh run
set -e
cd "$(dirname "$0")/.."
C=skeleton-action-recognition_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/bin
  for m in ${MODES:-0 23}; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DSAR_G2_TIMELINE -DSAR_G2_ABLATE=$m -c $C/conv_graph_cn8.hip -o tools/bin/g2_tl$m.o
    OTHERS=$(ls $C/*.o | grep -v "/conv_graph_cn8.o\|\.lds")
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libsar_g2tl$m.so tools/bin/g2_tl$m.o $OTHERS
  done
else
  for m in ${MODES:-0 23}; do
    echo "== SAR_G2_ABLATE=$m"; SAR_HIP_LIB=$PWD/tools/bin/libsar_g2tl$m.so python tools/g2_timeline.py | grep -v "^/opt"
  done
fi
