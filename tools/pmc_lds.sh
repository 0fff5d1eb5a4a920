#!/bin/bash
# development aid: LDS bank-conflict ratio per kernel over 2 training steps
mkdir -p gpurun_out; rm -rf /tmp/pmcL
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmcL -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/pmcL_run.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > gpurun_out/pmc_lds.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('/tmp/pmcL/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:56]
        agg[n][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': cnt[n] += 1
rows = []
for n, c in agg.items():
    act = c.get('SQ_LDS_IDX_ACTIVE', 0.0)
    rows.append((c.get('SQ_LDS_BANK_CONFLICT', 0.0), act, c.get('SQ_BUSY_CYCLES', 0.0), cnt[n], n))
for conf, act, busy, k, n in sorted(rows, reverse=True)[:30]:
    print("%-58s calls %4d  conflict %.3e  lds_active %.3e  ratio %.2f  busy %.3e" % (n, k, conf, act, conf / act if act else 0, busy))
PY
cat gpurun_out/pmc_lds.txt
