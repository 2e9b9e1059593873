#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"   -> gpurun_variants/lib_<name>.so  (development A/B builds)
set -e
cd "$(dirname "$0")/.."
name=$1; extra=$2
out=variants/lib_$name.so
mkdir -p variants /tmp/lbfa_var_$name
C=lowbit_quant_fa2_paddle_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize $extra"
rm -f /tmp/lbfa_var_$name/*.o $out
pids=""
for f in lbfa_api quant_kernels attn_fwd; do
  /opt/rocm/bin/hipcc $FLAGS -c $C/$f.hip -o /tmp/lbfa_var_$name/$f.o &
  pids="$pids $!"
done
for d in 64 128; do  # attn_fwd16.hip: one translation unit per head dim (as the Makefile does)
  f16=""; [ $d = 64 ] && f16="-mllvm -enable-post-misched=0"  # as the Makefile
  /opt/rocm/bin/hipcc $FLAGS $f16 -DLBFA_D16=$d -c $C/attn_fwd16.hip -o /tmp/lbfa_var_$name/attn_fwd16_d$d.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done   # set -e: a failed compile stops here
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out /tmp/lbfa_var_$name/*.o
echo built $out
