#!/bin/bash
# L2 -> fabric bytes per attention launch for one bench workload (run via gpurun): the two TCC passes of profile_round.sh only.
# usage: tools/traffic.sh <tag> <workload> [lib]   -> prints (2 * FETCH_SIZE + WRITE_SIZE) * 1024 for the low-bit attention kernel
tag=$1; wl=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/traffic_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc2 -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-c5 --no-fa2 > $out/pmc2.json 2> $out/pmc2.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc3 -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-c5 --no-fa2 > $out/pmc3.json 2> $out/pmc3.err
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for d in ("pmc2", "pmc3"):
    for f in glob.glob("$out/" + d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if ("attn_fwd16_kernel<" in k and k.split("<", 1)[1].split(",")[1].strip() == "3") or "attn_fwd_kernel<" in k:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
f = sum(acc["FETCH_SIZE"]) / max(len(acc["FETCH_SIZE"]), 1)
w = sum(acc["WRITE_SIZE"]) / max(len(acc["WRITE_SIZE"]), 1)
print("$tag $wl traffic_bytes_per_launch", round((2 * f + w) * 1024), "fetch_x2", round(2 * f * 1024), "write", round(w * 1024))
PY
