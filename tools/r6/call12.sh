for rep in 1 2; do
for lib in base a2; do echo "== $lib"; SAR_HIP_LIB=$PWD/tools/bin/libsar_$lib.so timeout 300 python tools/kernel_bench.py --split f16x3a --only tconv_fwd,tconv_dgrad --reps 7 2>&1 | grep "TOTAL\|^L3 \|^L7 \|^L10"; done
done
timeout 900 python -m pytest tests/test_gpu_stgcn_kernels.py tests/test_gpu_split.py -x -q -m gpu -k "temporal or f16 or scale" 2>&1 | tail -3
for lib in base a2 base a2; do echo "== $lib"; SAR_HIP_LIB=$PWD/tools/bin/libsar_$lib.so python bench.py --mfma f32_split --steps 60 --warmup 5 --warm-seconds 3 --no-cpu-baseline --no-isolated-pass --no-secondary 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
