#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel stats, and the two HBM-traffic counter passes of
# the same command.  Outputs under gpurun_out/round/, summarised by tools/profile_summary.py into profiles/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
BENCH="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exchange-proxy"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- $BENCH --prof-steps 0 --no-kernel-timer --sampler-steps 0 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- $BENCH --prof-steps 0 --no-kernel-timer --sampler-steps 0 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o r -- $BENCH --prof-steps 0 --no-kernel-timer --sampler-steps 0 > /dev/null 2> $OUT/pmc_mfma.err
python tools/profile_summary.py $OUT > $OUT/summary.txt 2>&1
# the raw traces are large (gpurun merges at most 64 MiB back): keep the per-kernel stats and the summaries only
rm -f $OUT/stats/*kernel_trace.csv $OUT/pmc_*/*counter_collection.csv
tail -5 $OUT/summary.txt
ls $OUT $OUT/stats | head -30
