#!/bin/bash
# Round-end evidence on the GPU box: tests, trace + PMC passes, bench lines of configs 3 / 2 / 4, work counters.
# Usage (through gpurun): bash scripts/final_evidence.sh   -> gpurun_out/final2/, gpurun_out/prof_r2final2/
set -u
mkdir -p gpurun_out/final2
export GRAFT_COMMIT=${GRAFT_COMMIT:-unknown}   # the snapshot has no .git: pass the commit in (GRAFT_COMMIT=$(git rev-parse --short HEAD))
python -m pytest tests -m gpu -q > gpurun_out/final2/tests.log 2>&1; tail -1 gpurun_out/final2/tests.log
bash scripts/profile_gpu.sh r2final2 "trace sq1 fetch write" --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/final2/prof.log 2>&1
cp gpurun_out/prof_r2final2/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
python bench.py > gpurun_out/final2/bench3.json 2> gpurun_out/final2/bench3.err
python bench.py --config 2 > gpurun_out/final2/bench2.json 2> gpurun_out/final2/bench2.err
python bench.py --config 4 > gpurun_out/final2/bench4.json 2> gpurun_out/final2/bench4.err
python scratch/stats_probe.py 10000000 > gpurun_out/final2/stats.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/final2/pmc_traffic.json
