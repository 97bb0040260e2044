mkdir -p gpurun_out/r6
SAR_WGRAD_STREAM=0 python tools/step_table.py --mfma f32_split 2>&1 | tee gpurun_out/r6/step_table_split_single.txt
SAR_WGRAD_STREAM=0 SAR_GRAPH_SPLIT2=0 python tools/step_table.py --mfma f32_split 2>&1 | tee gpurun_out/r6/step_table_split_single_graphv1.txt
python tools/step_table.py --mfma f32_split 2>&1 | head -3
SAR_GRAPH_SPLIT2=0 python tools/step_table.py --mfma f32_split 2>&1 | head -3
