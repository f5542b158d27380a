# which stage of the default bench run slows the 256x256 net's extra-workload measurement (39.5 ms stand-alone, 44 ms inside)?
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), {k:(round(v['value'],1),round(v['ms_per_step'],2)) for k,v in d['workloads'].items()})"; }
B="--steps 10 --warmup 3 --no-cpu-baseline"
python bench.py $B 2>/dev/null | p all_stages
python bench.py $B --no-exchange-proxy 2>/dev/null | p no_exchange_proxy
python bench.py $B --no-parity-probe 2>/dev/null | p no_parity_probe
python bench.py $B --sampler-steps 0 2>/dev/null | p no_sampler
python bench.py $B --no-kernel-timer --prof-steps 0 2>/dev/null | p no_kernel_timer
