#!/bin/bash
# A/B of the waves-per-workgroup choice of the D = 128 int8 kernels (LBFA_NW=4|8) - for the experiment build of commit c7408af only (the shipped kernels have no such switch): tools/nw_ab.sh "<workloads>" [rounds]
wls=${1:-"d128 c3"}; rounds=${2:-2}
for r in $(seq $rounds); do
for nw in 4 8; do
  for wl in $wls; do
    LBFA_NW=$nw timeout -k 10 200 python bench.py --workload $wl --no-sweep --no-c5 --no-cpu-baseline --no-fa2 --steps 20 --warmup 5 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('NW=$nw', '$wl', 'whole', round(d['value'],1), 'kernel', round(d['roofline']['achieved'],1), 'frac', round(d['roofline']['frac'],4), 'ms', d['ms_per_step'], 'mse', d['accuracy']['mse'])" || echo "FAILED NW=$nw $wl"
  done
done
done
