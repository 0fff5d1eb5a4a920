#!/bin/bash
# development aid: per-kernel times of one conv shape with the LDS-patch kernels vs the generic kernels (rocprofv3 stats)
# usage: wgrad_ab.sh B H W cin cout k
mkdir -p gpurun_out
for mode in patch nopatch; do
  rm -rf /tmp/wab_$mode
  (cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wab_$mode -- python3 $GRAFT_REPO_ROOT/tools/conv_probe.py "$@" 10 $mode > /dev/null 2>&1 < /dev/null)
  f=$(find /tmp/wab_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode $*"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:70]
    if any(k in n for k in ("conv_","unpack_wgrad","colsum")): print("   %-72s calls %4s avg %9.1f us"%(n, r["Calls"], float(r["AverageNs"])/1e3))
PY
done
