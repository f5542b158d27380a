#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats + PMC passes of the attention micro-benchmark (tools/bench_kernels.py --only attn)
set -u
cd /tmp && export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/attn_prof"
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python $GRAFT_REPO_ROOT/tools/bench_kernels.py --only attn256 --reps 10"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ap_stats -o r -- $CMD > "$OUT/stats.log" 2>&1
cp $(find /tmp/ap_stats -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d /tmp/ap_$TAG -o r -- $CMD > "$OUT/pmc_$TAG.log" 2>&1
  F=$(find /tmp/ap_$TAG -name "*counter_collection.csv" | head -1)
  python - "$F" "$OUT/pmc_$TAG.txt" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
  k = r['Kernel_Name']
  if 'attn' not in k: continue
  import re
  k = 'attn' + ''.join(re.findall(r'attn_kernel<\d, \d+>|amax', k))
  agg[k][r['Counter_Name']] += float(r['Counter_Value'])
  cnt[(k, r['Counter_Name'])] += 1
with open(sys.argv[2], 'w') as f:
  for k, d in agg.items():
    for c, v in d.items():
      f.write(f'{k:<42} {c:<28} {v / cnt[(k, c)]:16.1f} per launch ({cnt[(k, c)]} launches)\n')
PY
done
head -12 "$OUT/kernel_stats.csv" | cut -c1-200
cat "$OUT"/pmc_*.txt
