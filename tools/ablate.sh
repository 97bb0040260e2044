#!/bin/bash
# Diagnostic builds of conv_gemm.hip with parts of the kernel removed (SAR_ABLATE bit mask: 1 no epilogue,
# 2 stage the first K-slab only, 4 no barriers) -> where does the time of the temporal GEMM go?
# Build here (no GPU needed): tools/ablate.sh build ; run on the GPU box: tools/ablate.sh run
set -e
cd "$(dirname "$0")/.."
C=skeleton-action-recognition_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/bin
  for m in ${MODES:-1 2 3}; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSAR_ABLATE=$m -c $C/conv_gemm.hip -o tools/bin/conv_gemm_a$m.o
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSAR_ABLATE=$m -c $C/conv_wgrad.hip -o tools/bin/conv_wgrad_a$m.o
    # every other object of the regular build (make -C $C first): the loader requires every symbol of include/sar_hip.h
    OTHERS=$(ls $C/*.o | grep -v -e "/conv_gemm.o" -e "/conv_wgrad.o")
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libsar_a$m.so tools/bin/conv_gemm_a$m.o tools/bin/conv_wgrad_a$m.o $OTHERS
  done
else
  K=${KERNELS:-tconv_fwd,tconv_dgrad,gcn_fwd}
  echo "== full"; python tools/kernel_bench.py --only $K --layers 2,6,9 | grep -v TOTAL
  for m in ${MODES:-1 2 3}; do
    echo "== SAR_ABLATE=$m"; SAR_HIP_LIB=$PWD/tools/bin/libsar_a$m.so python tools/kernel_bench.py --only $K --layers 2,6,9 | grep -v TOTAL
  done
fi
