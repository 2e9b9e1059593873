#!/bin/bash
# fp16 vs bf16 storage on one box: whole operator + attention kernel, C2 / D128 S4K / S8K D128 (via --workload d128) / C3
for wl in c2 d128 c3; do
  for dt in fp16 bf16; do
    timeout -k 10 200 python bench.py --workload $wl --dtype $dt --no-sweep --no-c5 --no-cpu-baseline --no-fa2 --steps 20 --warmup 5 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$wl', '$dt', 'whole', round(d['value'],1), 'kernel', round(d['roofline']['achieved'],1), 'ms', d['ms_per_step'])" || echo "FAILED $wl $dt"
  done
done
