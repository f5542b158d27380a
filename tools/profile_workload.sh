#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats and the HBM-traffic / MFMA counter passes of ONE bench.py workload
# other than the headline (celeba64, celebahq256: BASELINE configs[2] / configs[4]), so that every `frac` and `traffic` of the
# bench line can be recomputed from a file of the same round.  Outputs under gpurun_out/round_<workload>/; copy
#   summary.txt -> profiles/rNN_<workload>_kernel_stats.txt, traffic.json -> profiles/rNN_<workload>_traffic.json
# (bench.py: traffic_file_for).  Counter passes are separate runs with --pmc only (MI355X_MICROARCH.md; gpurun refuses --pmc
# together with the trace domains).  One stream (STK_WGRAD_STREAM=0) in the profiled passes, as in tools/profile_round.sh.
#   tools/profile_workload.sh celeba64
set -u
W="$1"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/round_$W
rm -rf $OUT; mkdir -p $OUT
Q="--workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-exchange-proxy --no-extra-workloads --no-parity-probe --sampler-steps 0"
python bench.py $Q --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err
BENCH="env STK_WGRAD_STREAM=0 python bench.py $Q --detail /tmp/pw_detail.json"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- $BENCH --prof-steps 0 --no-kernel-timer > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- $BENCH --prof-steps 0 --no-kernel-timer > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o r -- $BENCH --prof-steps 0 --no-kernel-timer > /dev/null 2> $OUT/pmc_mfma.err
python tools/profile_summary.py $OUT "--workload $W" > $OUT/summary.txt 2>&1
rm -f $OUT/stats/*kernel_trace.csv $OUT/stats/*/*kernel_trace.csv $OUT/pmc_*/*counter_collection.csv $OUT/pmc_*/*/*counter_collection.csv
tail -3 $OUT/summary.txt
