python -m pytest tests/test_planes.py -m gpu -q 2>&1 | tail -5
python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_planes.json 2> gpurun_out/bench_planes.err; python - <<'PY'
import json
r=json.loads(open('gpurun_out/bench_planes.json').read().strip().splitlines()[-1])
print('planes default:', r['value'], r['ms_per_step'], r['config']['loss_mean'])
for k,v in r.get('kernels',{}).items(): print(' ', k, v)
PY
