#!/bin/bash
# A/B kernel builds on both input distributions: tools/ab2.sh "<variant names>" "<kbench cfgs>" [rounds]
vars=${1:-"cur"}; cfgs=${2:-"c5 f8d64"}; rounds=${3:-2}
for r in $(seq $rounds); do
  for v in $vars; do
    for d in normal randint; do
      echo "== $v $d (round $r)"
      LBFA_LIB_PATH=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/kbench.py --cfg $cfgs --iters 12 --dist $d || echo "FAILED $v"
    done
  done
done
