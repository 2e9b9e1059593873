#!/bin/bash
# spill / register summary of the attn_fwd16 instances of one head dim: tools/regs16.sh <64|128> [extra flags]
d=$1; shift
f=""; [ "$d" = 64 ] && f="-mllvm -enable-post-misched=0"
"$(dirname "$0")/regs.sh" "$(dirname "$0")/../lowbit_quant_fa2_paddle_amd/csrc/attn_fwd16.hip" "$f -DLBFA_D16=$d $*" | python3 -c "
import sys, ast
for l in sys.stdin:
    d = ast.literal_eval(l)
    print(d['fn'][6:], 'vgpr', d['VGPRs'], 'spill', d['VGPRssp'], 'sgpr_spill', d['SGPRssp'])
"
