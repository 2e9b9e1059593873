#!/bin/bash
# round-3 A/B: tools/r3_ab.sh "<variants>" "<workloads>" "<dists>"  -> whole / kernel TFLOP/s per (variant, workload, distribution)
vars=${1:-"cur"}; wls=${2:-"c2 s16k d128"}; dists=${3:-"normal randint"}; rounds=${4:-1}
for r in $(seq $rounds); do
for v in $vars; do
  lib=$PWD/variants/lib_$v.so; [ "$v" = cur ] && lib=$PWD/lowbit_quant_fa2_paddle_amd/liblowbit_fa_hip.so
  for d in $dists; do
    for wl in $wls; do
      LBFA_LIB_PATH=$lib timeout -k 10 200 python bench.py --workload $wl --dist $d --no-sweep --no-c5 --no-cpu-baseline --no-fa2 --steps 20 --warmup 5 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', '$wl', '$d', 'whole', round(d['value'],1), 'kernel', round(d['roofline']['achieved'],1), 'frac', round(d['roofline']['frac'],4), 'ms', d['ms_per_step'])" || echo "FAILED $v $wl $d"
    done
  done
done
done
