#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for v in s1v1 s1v2; do
  export DNAGPU_LIB_PATH=$GRAFT_REPO_ROOT/build_ab/libdnagpu_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 $GRAFT_REPO_ROOT/tools/sk_once.py 3e9 31 2 > /tmp/$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; tail -2 /tmp/$v.log | cut -c1-200
  grep -E "sk_scatter1|sk_regroup|sk_hist1" $f | cut -d, -f1-4 | cut -c1-60,200-
  python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if any(x in r["Name"] for x in ("sk_scatter1","sk_regroup","sk_hist1","sk_scatter0")):
        print(r["Name"][:30], r["Calls"], round(float(r["AverageNs"])/1e6,3))
PY
done
