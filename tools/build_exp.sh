#!/bin/bash
# usage: tools/build_exp.sh <name> "<extra flags>"  -> variants/lib_<name>.so: only attn_fwd16.hip is recompiled (both head-dim
# units), the other objects come from the last `make` in lowbit_quant_fa2_paddle_amd/csrc (development experiments)
set -e
cd "$(dirname "$0")/.."
name=$1; extra=$2
C=lowbit_quant_fa2_paddle_amd/csrc
mkdir -p variants /tmp/lbfa_exp_$name
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize $extra"
/opt/rocm/bin/hipcc $FLAGS -mllvm -enable-post-misched=0 -DLBFA_D16=64 -c $C/attn_fwd16.hip -o /tmp/lbfa_exp_$name/attn_fwd16_d64.o &
p1=$!
/opt/rocm/bin/hipcc $FLAGS -DLBFA_D16=128 -c $C/attn_fwd16.hip -o /tmp/lbfa_exp_$name/attn_fwd16_d128.o &
p2=$!
wait $p1; wait $p2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$name.so /tmp/lbfa_exp_$name/attn_fwd16_d64.o \
  /tmp/lbfa_exp_$name/attn_fwd16_d128.o $C/lbfa_api.o $C/quant_kernels.o $C/attn_fwd.o
echo built variants/lib_$name.so
