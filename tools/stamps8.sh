#!/bin/bash
# In-kernel phase stamps of the 9-tap CN8 kernel (conv_gemm_cn8.hip, -DSAR_CN8_STAMPS): where do a workgroup's cycles go?
#   build here: tools/stamps8.sh build ;  on the GPU box: tools/stamps8.sh run
set -e
cd "$(dirname "$0")/.."
C=skeleton-action-recognition_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/bin
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DSAR_CN8_STAMPS=${STAMPS:-1} -c $C/conv_gemm_cn8.hip -o tools/bin/cn8_stamps.o
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSAR_CN8_STAMPS=1 -c $C/conv_wgrad_cn8.hip -o tools/bin/wg8_stamps.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libsar_stamps.so tools/bin/cn8_stamps.o tools/bin/wg8_stamps.o $(ls $C/*.o | grep -v "/conv_gemm_cn8.o\|/conv_wgrad_cn8.o\|\.lds")
else
  SAR_HIP_LIB=$PWD/tools/bin/libsar_stamps.so python tools/kbench8.py ${KB:-t_fwd,t_dgrad} | grep -v "^/opt"
fi
