#!/bin/bash
# bench.py over every workload (one JSON line each) -> gpurun_out/sweep_<tag>.jsonl   (run via gpurun)
tag=${1:-r}
out=gpurun_out/sweep_$tag.jsonl
: > $out
for wl in c2 c2c s8k s16k s32k d128 c3 c4 c4m c5; do
  timeout -k 10 280 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-c5 >> $out 2>> gpurun_out/sweep_$tag.err || echo "{\"workload\": \"$wl\", \"failed\": true}" >> $out
done
python - <<PY
import json
for l in open("$out"):
    d = json.loads(l)
    if d.get("failed"):
        print(d); continue
    f = d.get("fa2_reference") or {}
    own = (f.get("own_fp16_kernel") or {}).get("tflops")
    print(f'{d["config"]["workload"][:58]:58s} whole {d["value"]:8.1f}  kernel {d["roofline"]["achieved"]:8.1f}  frac {d["roofline"]["frac"]:.3f}  ms {d["ms_per_step"]:8.4f}  torchFA2 {f.get("tflops")}  ownfp16 {own}')
PY
