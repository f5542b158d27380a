#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel stats, and the two HBM-traffic counter passes of
# the same command.  Outputs under gpurun_out/round/, summarised by tools/profile_summary.py into profiles/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
python bench.py --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err
# The profiled passes run the backward on ONE stream (STK_WGRAD_STREAM=0): with the weight gradients on the side stream kernels
# of two streams overlap, and a kernel's traced duration then includes the time it shares the chip (x2w::wgrad_kernel<32> 232 us
# against 174 alone).  The benchmark line itself (bench.json) is the default engine; bench.py's own event brackets
# (roofline object) are taken with one stream as well (engine/profile.KernelTimer).
export STK_WGRAD_STREAM_PROFILE=0
BENCH="env STK_WGRAD_STREAM=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exchange-proxy --no-extra-workloads --detail /tmp/profile_round_detail.json"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- $BENCH --sampler-steps 0 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- $BENCH --prof-steps 0 --no-kernel-timer --sampler-steps 0 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- $BENCH --prof-steps 0 --no-kernel-timer --sampler-steps 0 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o r -- $BENCH --prof-steps 0 --no-kernel-timer --sampler-steps 0 > /dev/null 2> $OUT/pmc_mfma.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exchange-proxy --no-extra-workloads --no-parity-probe --sampler-steps 0 --no-kernel-timer --prof-steps 0 --detail /tmp/profile_round_detail.json > /dev/null 2> $OUT/stats2.err
python tools/profile_summary.py $OUT > $OUT/summary.txt 2>&1
python - $OUT/stats2 > $OUT/two_stream_kernel_stats.txt <<'PY'
import csv, glob, os, re, sys
f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_stats.csv'), recursive=True)
print('default engine (weight gradients on the side stream): traced durations include the time a kernel shares the chip')
for r in list(csv.DictReader(open(f[0])))[:30] if f else []:
  n = re.sub(r'\(anonymous namespace\)::|void ', '', r['Name']).split('(')[0][:90]
  print(f"{n:92s} {int(r['Calls']):8d} {float(r['AverageNs']) / 1e3:9.1f} {float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['Percentage']):6.2f}")
PY
# the raw traces are large (gpurun merges at most 64 MiB back): keep the per-kernel stats and the summaries only
rm -f $OUT/stats/*kernel_trace.csv $OUT/stats2/*kernel_trace.csv $OUT/stats2/*/*kernel_trace.csv $OUT/pmc_*/*counter_collection.csv
tail -5 $OUT/summary.txt
ls $OUT $OUT/stats | head -30
