#!/bin/bash
# A/B of builds on bf16 storage: tools/bf16_ab.sh "<variants>" "<workloads>" [rounds]
vars=${1:-"cur"}; wls=${2:-"c2 c3"}; rounds=${3:-2}
for r in $(seq $rounds); do
for v in $vars; do
  lib=$PWD/variants/lib_$v.so; [ "$v" = cur ] && lib=$PWD/lowbit_quant_fa2_paddle_amd/liblowbit_fa_hip.so
  for wl in $wls; do
    LBFA_LIB_PATH=$lib timeout -k 10 200 python bench.py --workload $wl --dtype bf16 --no-sweep --no-c5 --no-cpu-baseline --no-fa2 --steps 20 --warmup 5 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', '$wl', 'bf16 whole', round(d['value'],1), 'kernel', round(d['roofline']['achieved'],1), 'ms', d['ms_per_step'], 'mse', d['accuracy']['mse'])" || echo "FAILED $v $wl"
  done
done
done
