timeout 1500 python -m pytest tests/test_gpu_stgcn_kernels.py tests/test_gpu_split.py tests/test_gpu_conv2d_kernels.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for lib in base a2; do echo "== $lib"; SAR_HIP_LIB=$PWD/tools/bin/libsar_$lib.so timeout 300 python tools/kernel_bench.py --split f16x3a --only tconv_wgrad,gcn_wgrad --reps 7 2>&1 | grep "TOTAL"; done
done
for lib in base a2 base a2; do echo "== $lib"; SAR_HIP_LIB=$PWD/tools/bin/libsar_$lib.so python bench.py --mfma f32_split --steps 60 --warmup 5 --warm-seconds 3 --no-cpu-baseline --no-isolated-pass --no-secondary 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
for lib in base a2 base a2; do echo "== pathB $lib"; SAR_HIP_LIB=$PWD/tools/bin/libsar_$lib.so python bench.py --workload spectrogram --mfma f32_split --steps 200 --warmup 5 --warm-seconds 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
