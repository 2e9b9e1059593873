#!/bin/bash
# A/B of BASELINE configs[4] itself (B = 32 on one GPU, `c5_strong` of the bench line): tools/c5s_ab.sh "<variants>" [rounds]
vars=${1:-"cur"}; rounds=${2:-1}
for r in $(seq $rounds); do
for v in $vars; do
  lib=$PWD/variants/lib_$v.so; [ "$v" = cur ] && lib=$PWD/lowbit_quant_fa2_paddle_amd/liblowbit_fa_hip.so
  LBFA_LIB_PATH=$lib timeout -k 10 300 python bench.py --no-sweep --no-cpu-baseline --no-fa2 --steps 5 --warmup 2 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['c5_strong']; print('$v', 'c5_strong', c['tflops_total'], 'TFLOP/s', c['ms'], 'ms')" || echo "FAILED $v"
done
done
