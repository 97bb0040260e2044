#!/bin/bash
# A/B: KC=8 (47 KB LDS/WG) vs KC=4 (24 KB LDS/WG) on the temporal GEMMs
cd $GRAFT_REPO_ROOT
echo "=== KC=8"; python tools/kernel_bench.py --only tconv_fwd,tconv_dgrad --layers 1,5,8 2>&1 | grep -v amdgpu
cp skeleton-action-recognition_amd/sar_amd/libsar_hip.so /tmp/keep.so
cp skeleton-action-recognition_amd/sar_amd/libsar_hip_kc4.so skeleton-action-recognition_amd/sar_amd/libsar_hip.so
echo "=== KC=4"; python tools/kernel_bench.py --only tconv_fwd,tconv_dgrad --layers 1,5,8 2>&1 | grep -v amdgpu
cp /tmp/keep.so skeleton-action-recognition_amd/sar_amd/libsar_hip.so
