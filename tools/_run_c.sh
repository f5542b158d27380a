mkdir -p gpurun_out
python -m pytest tests/test_planes.py -m gpu -q -k "wgrad" 2>&1 | tail -8
python tools/bench_kernels.py --only conv --shapes 8 2>&1 | grep -E "wgrad"
for mb in 32 128; do echo "== STK_WGRAD_SLAB_MB=$mb"; STK_WGRAD_SLAB_MB=$mb python tools/bench_kernels.py --only conv --planes-only --shapes 3 2>&1 | grep -E "wgrad.planes"; done
