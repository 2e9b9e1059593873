#!/bin/bash
# A/B kernel builds on one GPU box: tools/ab.sh "<variant names>" "<kbench cfgs>"   (variants/lib_<name>.so, see build_variant.sh)
# Round-robin over the variants, several rounds, so that clock / box drift hits all of them alike.
vars=${1:-"base"}; cfgs=${2:-"c2 d128"}; rounds=${3:-2}
for r in $(seq $rounds); do
  for v in $vars; do
    echo "== $v (round $r)"
    LBFA_LIB_PATH=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/kbench.py --cfg $cfgs --iters 20 --check || echo "FAILED $v"
  done
done
