#!/bin/bash
# round-3 baseline: C2 on N(0,1) vs the reference's bench distribution randint(-100,100) (utils/benchmark.py:215-230)
for d in normal randint; do
  for wl in c2 s16k d128; do
    timeout -k 10 200 python bench.py --workload $wl --dist $d --no-sweep --no-c5 --no-cpu-baseline --no-fa2 --steps 20 --warmup 5 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$wl', '$d', 'whole', round(d['value'],1), 'kernel', round(d['roofline']['achieved'],1), 'frac', round(d['roofline']['frac'],4), 'ms', d['ms_per_step'])" || echo "FAILED $wl $d"
  done
done
