cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_abl; rm -rf $OUT; mkdir -p $OUT
for k in 1 16 112; do
  STK_PL_KERNEL=$k rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/k$k -o r -- python tools/bench_kernels.py --only conv --shapes 1 --planes-only --reps 10 > $OUT/k$k.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob('$OUT/k$k/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
  k = row['Kernel_Name']
  if 'gemm' not in k: continue
  agg[k][row['Counter_Name']] += float(row['Counter_Value']); 
  if row['Counter_Name']=='SQ_WAVES': cnt[k]+=1
for k,v in agg.items():
  n=cnt[k]
  print('k$k', k[:70], 'launches', n, {c: round(x/n) for c,x in v.items()})
t = glob.glob('$OUT/k$k/**/*kernel_trace.csv', recursive=True)
d = collections.defaultdict(list)
for row in csv.DictReader(open(t[0])):
  if 'gemm' in row['Kernel_Name']: d[row['Kernel_Name']].append(int(row['End_Timestamp'])-int(row['Start_Timestamp']))
for k,v in d.items(): print('k$k dur_us', k[:70], sum(v)/len(v)/1e3)
PY
done
rm -rf $OUT/k*/
