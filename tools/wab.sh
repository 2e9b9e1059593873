#!/bin/bash
# whole-operator A/B on one GPU box: tools/wab.sh "<variant names>" [rounds]   (bench.py's C2 line per variants/lib_<name>.so)
vars=${1:-"cur"}; rounds=${2:-2}
for r in $(seq $rounds); do
  for v in $vars; do
    LBFA_LIB_PATH=$PWD/variants/lib_$v.so timeout -k 10 300 python bench.py --no-sweep --no-c5 --no-cpu-baseline --no-fa2 --steps 60 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', 'whole', round(d['value'],1), 'kernel', round(d['roofline']['achieved'],1), 'frac', round(d['roofline']['frac'],4))" || echo "FAILED $v"
  done
done
