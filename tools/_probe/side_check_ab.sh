# same box, alternating: side stream picked by the overlap check vs the first pooled stream
B="--steps 30 --warmup 8 --no-cpu-baseline --no-exchange-proxy --no-extra-workloads --no-parity-probe --sampler-steps 0 --no-kernel-timer --prof-steps 0"
for i in 1 2 3; do
  for v in 1 0; do
    STK_SIDE_CHECK=$v python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STK_SIDE_CHECK=$v', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
