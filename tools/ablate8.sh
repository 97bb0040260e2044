#!/bin/bash
# Diagnostic builds of conv_gemm_cn8.hip with parts of the temporal kernel removed (SAR_ABLATE8 bit mask: 1 no MFMA, 2 global
# loads of stage 0 only, 4 no epilogue, 8 LDS stores of stage 0 only) -> where does the time of the bf16 temporal GEMM go?
# Build here: tools/ablate8.sh build ; run on the GPU box: tools/ablate8.sh run
set -e
cd "$(dirname "$0")/.."
C=skeleton-action-recognition_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/bin
  for m in ${MODES:-1 2 4 8 10 11}; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DSAR_ABLATE8=$m -c $C/conv_gemm_cn8.hip -o tools/bin/cn8_a$m.o
    OTHERS=$(ls $C/*.o | grep -v "/conv_gemm_cn8.o")
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libsar_c$m.so tools/bin/cn8_a$m.o $OTHERS
  done
else
  echo "== full"; python tools/kbench8.py t_fwd,t_dgrad | grep -v "^/opt"
  for m in ${MODES:-1 2 4 8 10 11}; do
    echo "== SAR_ABLATE8=$m"; SAR_HIP_LIB=$PWD/tools/bin/libsar_c$m.so python tools/kbench8.py t_fwd,t_dgrad | grep -v "^/opt"
  done
fi
