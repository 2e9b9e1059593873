#!/bin/bash
# per-kernel times of the one-call operator for A/B builds (run via gpurun): tools/prepass.sh "<variants>" "<workloads>"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in $2; do for v in $1; do
  rm -rf gpurun_out/kt_$v
  LBFA_LIB_PATH=$PWD/variants/lib_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$v -- python3 bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-sweep --no-c5 --no-fa2 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/kt_$v/*/*_kernel_stats.csv"):
    tot = 0
    for r in csv.DictReader(open(f)):
        if "lbfa" in r["Name"]:
            tot += float(r["AverageNs"]) / 1000
            print("$v $wl", r["Name"][:64], r["Calls"], round(float(r["AverageNs"]) / 1000, 2))
    print("$v $wl  sum of kernel averages", round(tot, 2), "us")
PY
done; done
