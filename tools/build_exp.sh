#!/bin/bash
# usage: tools/build_exp.sh <name> "<extra flags>" [base]  -> variants/lib_<name>.so: only attn_fwd16.hip is recompiled, the other
# objects come from an earlier tools/build_variant.sh <base> build (development experiments)
set -e
cd "$(dirname "$0")/.."
name=$1; extra=$2; base=${3:-vraw}
C=lowbit_quant_fa2_paddle_amd/csrc
mkdir -p /tmp/lbfa_exp_$name
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize $extra"
/opt/rocm/bin/hipcc $FLAGS -c $C/attn_fwd16.hip -o /tmp/lbfa_exp_$name/attn_fwd16.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$name.so /tmp/lbfa_exp_$name/attn_fwd16.o \
  /tmp/lbfa_var_$base/lbfa_api.o /tmp/lbfa_var_$base/quant_kernels.o /tmp/lbfa_var_$base/attn_fwd.o
echo built variants/lib_$name.so
