mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ops.py -m gpu -q -x -k "upfirdn or resampl or op" > gpurun_out/r02_tests_d.log 2>&1; grep -E "passed|failed" gpurun_out/r02_tests_d.log; grep -E "^E " gpurun_out/r02_tests_d.log | head -5
python tools/bench_kernels.py --only misc 2>&1 | grep -E "upfirdn"
python tools/bench_kernels.py --only gn 2>&1 | grep -E "gn_"
