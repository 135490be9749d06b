#!/bin/bash
# Round-end evidence on the GPU box: tests, trace + PMC passes of the DRIVER's bench command, bench lines of configs
# 3 / 2 / 4 (+ the rejector / reciprocal lines), work counters, per-stage profile of the stand-off search.
# Usage (through gpurun): GRAFT_COMMIT=$(git rev-parse --short HEAD) bash scripts/final_evidence.sh [tag]
#   -> gpurun_out/<tag>/, gpurun_out/prof_<tag>/
set -u
TAG=${1:-r3final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GRAFT_COMMIT=${GRAFT_COMMIT:-unknown}   # the snapshot has no .git: pass the commit in
python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; tail -1 $OUT/tests.log
# EVIDENCE_SHORT=1: what fits a few GPU-minutes -- tests, trace + FETCH/WRITE passes, the bench lines, k-NN timings
SHORT=${EVIDENCE_SHORT:-0}
PASSES="trace sq1 sq2 fetch write"; [ "$SHORT" = 1 ] && PASSES="trace fetch write"
bash scripts/profile_gpu.sh $TAG "$PASSES" > $OUT/prof.log 2>&1
cp gpurun_out/prof_$TAG/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
cp gpurun_out/prof_$TAG/summary.txt $OUT/prof_summary.txt 2>/dev/null
python bench.py > $OUT/bench3.json 2> $OUT/bench3.err
python bench.py --config 2 > $OUT/bench2.json 2> $OUT/bench2.err
python bench.py --config 4 > $OUT/bench4.json 2> $OUT/bench4.err
python bench.py --no-cpu-baseline --no-host-align --rejectors median,trimmed > $OUT/bench3_rejectors.json 2> $OUT/bench3_rejectors.err
python bench.py --no-cpu-baseline --no-host-align --reciprocal > $OUT/bench3_reciprocal.json 2> $OUT/bench3_reciprocal.err
python scratch/knn_probe.py 10000000 1 8 > $OUT/knn_probe.log 2>&1
if [ "$SHORT" != 1 ]; then
PCLHIP_LIB=pcl_amd/variants/libpclhip_stats.so python scratch/stats_probe.py 10000000 > $OUT/stats.log 2>&1
PCLHIP_LIB=pcl_amd/variants/libpclhip_prof.so python scratch/stats_probe.py 10000000 2>&1 | grep "it0" > $OUT/standoff_stage_ticks.log
PCLHIP_LIB=pcl_amd/variants/libpclhip_why.so python scratch/stats_probe.py 10000000 2>&1 | grep "it0" > $OUT/standoff_exits.log
# tick profiles of the normals kernel (pass 1 / pass 2 / plane fit) and of the seeded search body (before / traversal / after)
PCLHIP_LIB=pcl_amd/variants/libpclhip_nrmprof.so python scratch/stats_probe.py 10000000 2>&1 | grep "normals" > $OUT/normals_stage_ticks.log
PCLHIP_LIB=pcl_amd/variants/libpclhip_icpprof.so python scratch/stats_probe.py 10000000 2>&1 | grep "icp it" > $OUT/seeded_stage_ticks.log
PCLHIP_LIB=pcl_amd/variants/libpclhip_lanes.so python scratch/stats_probe.py 10000000 2>&1 | grep -E "normals|icp it" > $OUT/active_lanes_per_round.log
# the normals' fallback paths (records overflowing, exact policy): a build with one record per lane, same tests
PCLHIP_LIB=pcl_amd/variants/libpclhip_rec1.so python -m pytest tests -m gpu -q -k "normal or Normal or fuzz" > $OUT/tests_rec1.log 2>&1; tail -1 $OUT/tests_rec1.log
# randomised rounds against the oracle on the final build (scratch/fuzz_*.py <seed> <rounds>)
for seed in 11 12; do timeout 200 python scratch/fuzz_knn.py $seed 30 2>&1 | tail -1; done > $OUT/fuzz_knn.log
for seed in 11 12; do timeout 200 python scratch/fuzz_filters.py $seed 25 2>&1 | tail -1; done > $OUT/fuzz_filters.log
cat $OUT/fuzz_knn.log $OUT/fuzz_filters.log
bash scripts/profile_iter.sh $TAG > $OUT/prof_iter.log 2>&1; cp gpurun_out/prof_$TAG/per_iter.txt $OUT/per_iter.txt 2>/dev/null
python scratch/first_probe.py > $OUT/first_call.log 2>&1
python scratch/misc_probe.py > $OUT/misc.log 2>&1
fi
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
for f in $OUT/bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], "ms_per_step", d["ms_per_step"], "value %.4g" % d["value"], "frac", d.get("roofline", {}).get("frac"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
