#!/bin/bash
# Diagnostic builds of conv_graph_cn8.hip with parts removed (SAR_G2_ABLATE bit mask: 1 no MFMA, 2 global loads of stage 0 only,
# 4 no epilogue, 8 no mini-builder, 16 LDS stores of stage 0 only) -> where does the time of the bf16 graph convolution go?
# Build here: tools/ablate_g2.sh build ; run on the GPU box: tools/ablate_g2.sh run
set -e
cd "$(dirname "$0")/.."
C=skeleton-action-recognition_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/bin
  for m in ${MODES:-1 2 4 8 18 19 23}; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DSAR_G2_ABLATE=$m -c $C/conv_graph_cn8.hip -o tools/bin/g2_a$m.o
    OTHERS=$(ls $C/*.o | grep -v "/conv_graph_cn8.o\|\.lds")
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libsar_g$m.so tools/bin/g2_a$m.o $OTHERS
  done
else
  echo "== full"; python tools/kbench8.py g_fwd,g_dgrad | grep -v "^/opt"
  for m in ${MODES:-1 2 4 8 18 19 23}; do
    echo "== SAR_G2_ABLATE=$m"; SAR_HIP_LIB=$PWD/tools/bin/libsar_g$m.so python tools/kbench8.py g_fwd,g_dgrad | grep -v "^/opt"
  done
fi
