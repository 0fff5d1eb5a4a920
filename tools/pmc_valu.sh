#!/bin/bash
# development aid: where the wave cycles of each kernel go (VALU / waiting), over 2 training steps
mkdir -p gpurun_out; rm -rf /tmp/pmcV
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d /tmp/pmcV -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/pmcV_run.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > gpurun_out/pmc_valu.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('/tmp/pmcV/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:50]
        agg[n][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': cnt[n] += 1
print("%-52s %5s %10s %6s %6s %6s %6s %6s" % ("kernel", "calls", "busy", "valu%", "lds%", "wait%", "winst%", "act%"))
for n, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0))[:34]:
    wc = c.get('SQ_WAVE_CYCLES', 1.0)
    print("%-52s %5d %10.3e %6.1f %6.1f %6.1f %6.1f %6.1f" % (n, cnt[n], c.get('SQ_BUSY_CYCLES', 0), 100 * c.get('SQ_ACTIVE_INST_VALU', 0) / wc,
          100 * c.get('SQ_ACTIVE_INST_LDS', 0) / wc, 100 * c.get('SQ_WAIT_ANY', 0) / wc, 100 * c.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc))
PY
cat gpurun_out/pmc_valu.txt
