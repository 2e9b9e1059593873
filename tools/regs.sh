#!/bin/bash
# prints VGPR / occupancy / spill per kernel of one HIP source (development tool)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize $2 -c $1 -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur={}
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass', l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'):
        if cur: print(cur)
        cur={'fn':t.split(':',1)[1].strip().replace('_ZN4lbfa15attn_fwd_kernel','attn').replace('_ZN4lbfa17attn_fwd16_kernel','attn16').replace('EEvNS_10AttnParamsE','')}
    else:
        k,v=t.split(':',1)
        if k.strip() in ('VGPRs','SGPRs','Occupancy [waves/SIMD]','VGPRs Spill','SGPRs Spill','LDS Size [bytes/block]'): cur[k.strip().split(' ')[0]+('sp' if 'Spill' in k else '')]=v.strip()
if cur: print(cur)
"
