for v in 0 1; do echo "== SAR_COMPACT_SKIP=$v"; SAR_COMPACT_SKIP=$v SAR_WGRAD_STREAM=0 python tools/step_table.py --mfma f32_split 2>&1 | grep -v amdgpu; done
