#!/bin/bash
# development aid: PMC counters for one conv shape.  usage: pmc_conv.sh B H W cin cout k
mkdir -p gpurun_out; rm -rf gpurun_out/pmc1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py "$@" 3 ${PATCHMODE:-nopatch} > $GRAFT_REPO_ROOT/gpurun_out/pmc1.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/pmc1.txt
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc1/**/*counter_collection.csv',recursive=True)
if not f: print('no counter file', glob.glob('gpurun_out/pmc1/**/*',recursive=True)[:10]); raise SystemExit
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    n=r['Kernel_Name'].replace('void (anonymous namespace)::','').split('(')[0][:50]
    agg[n][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[n]+=1
for n,c in agg.items():
    if 'conv' not in n or 'pack' in n: continue
    print(n, 'dispatches',cnt[n])
    for k,v in sorted(c.items()): print('    %-28s %.4g'%(k,v/max(cnt[n],1)))
PY
