#!/bin/bash
# Collect the rocprofv3 evidence for one bench workload on the GPU box (run via gpurun):
#   kernel-trace stats + three separate PMC passes (SQ counters, FETCH_SIZE, WRITE_SIZE+GRBM) as the
#   MI355X guide prescribes (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share a pass; no --pmc with sys-trace).
# usage: tools/profile_round.sh <tag> [workload] ["extra bench.py arguments", e.g. "--dist randint"]
set -e
tag=$1; wl=${2:-c2}; extra=${3:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-c5 $extra > $out/kt.json 2> $out/kt.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc1 -- python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-sweep --no-c5 $extra > $out/pmc1.json 2> $out/pmc1.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc2 -- python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-sweep --no-c5 $extra > $out/pmc2.json 2> $out/pmc2.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc3 -- python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-sweep --no-c5 $extra > $out/pmc3.json 2> $out/pmc3.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $out/pmc4 -- python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-sweep --no-c5 $extra > $out/pmc4.json 2> $out/pmc4.err || true
echo done $tag
