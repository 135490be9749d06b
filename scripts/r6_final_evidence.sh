#!/bin/bash
# Round-6 evidence of ONE commit on the GPU box (~25 GPU-minutes): the -m gpu tests, trace + SQ + FETCH/WRITE passes of
# the driver's bench command (-> pmc_traffic.json stamped with the commit and the kernel sources' hash), the bench lines
# (configs 3 / 2 / 4, rejectors, reciprocal, the cloud families, config 5: ranks 0 / 3 / 6 of eight at 100M points and the
# single slab WITH its CPU baseline), the instrumented builds (scripts/build_evidence_variants.sh first), per-iteration
# PMC passes.
#   GRAFT_COMMIT=$(git rev-parse --short HEAD) bash scripts/r6_final_evidence.sh [tag]  -> gpurun_out/<tag>/, gpurun_out/prof_<tag>/
set -u
TAG=${1:-r6final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GRAFT_COMMIT=${GRAFT_COMMIT:-unknown}
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/tests.log 2>&1; grep -E "passed|failed|real" $OUT/tests.log | tail -2
bash scripts/profile_gpu.sh $TAG "trace sq1 sq2 fetch write" > $OUT/prof.log 2>&1
cp gpurun_out/prof_$TAG/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
cp gpurun_out/prof_$TAG/summary.txt $OUT/prof_summary.txt 2>/dev/null
timeout 600 python bench.py > $OUT/bench3.json 2> $OUT/bench3.err
timeout 300 python bench.py --config 2 > $OUT/bench2.json 2> $OUT/bench2.err
timeout 300 python bench.py --config 4 > $OUT/bench4.json 2> $OUT/bench4.err
timeout 300 python bench.py --no-cpu-baseline --no-host-align --rejectors median,trimmed > $OUT/bench3_rejectors.json 2> $OUT/bench3_rejectors.err
timeout 300 python bench.py --no-cpu-baseline --no-host-align --reciprocal > $OUT/bench3_reciprocal.json 2> $OUT/bench3_reciprocal.err
for c in cube layers clusters; do
  timeout 300 python bench.py --cloud $c --no-cpu-baseline --no-host-align > $OUT/bench3_cloud_$c.json 2> $OUT/bench3_cloud_$c.err
done
for r in 0 3 6; do
  timeout 300 python bench.py --config 5 --points 100000000 --virtual-world 8 --virtual-rank $r --no-cpu-baseline > $OUT/bench5_100M_rank$r.json 2> $OUT/bench5_100M_rank$r.err
done
timeout 900 python bench.py --config 5 --points 100000000 > $OUT/bench5_100M_1gpu.json 2> $OUT/bench5_100M_1gpu.err
N=10000000 bash scratch/prof_build.sh ${TAG}_build > $OUT/index_build_dispatches.log 2>&1
if [ -f pcl_amd/variants/libpclhip_kdbticks.so ]; then
PCLHIP_LIB=pcl_amd/variants/libpclhip_kdbticks.so timeout 300 python scratch/kdb_ticks.py 2>&1 | grep -v amdgpu.ids > $OUT/kd_block_kernel_ticks.log
fi
timeout 300 python scratch/knn_probe.py 10000000 1 8 > $OUT/knn_probe.log 2>&1
timeout 300 python scratch/misc_probe.py > $OUT/misc.log 2>&1
if [ -f pcl_amd/variants/libpclhip_stats.so ]; then
PCLHIP_LIB=pcl_amd/variants/libpclhip_stats.so timeout 300 python scratch/stats_probe.py 10000000 > $OUT/stats.log 2>&1
PCLHIP_LIB=pcl_amd/variants/libpclhip_prof.so timeout 300 python scratch/stats_probe.py 10000000 2>&1 | grep "it0" > $OUT/standoff_stage_ticks.log
PCLHIP_LIB=pcl_amd/variants/libpclhip_icpprof.so timeout 300 python scratch/stats_probe.py 10000000 2>&1 | grep "icp it" > $OUT/seeded_stage_ticks.log
PCLHIP_LIB=pcl_amd/variants/libpclhip_lanes.so timeout 300 python scratch/stats_probe.py 10000000 2>&1 | grep -E "normals|icp it" > $OUT/active_lanes_per_round.log
fi
bash scripts/profile_iter.sh $TAG > $OUT/prof_iter.log 2>&1; cp gpurun_out/prof_$TAG/per_iter.txt $OUT/per_iter.txt 2>/dev/null
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
for f in $OUT/bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    its = {}
    for s in d.get("per_step", []):
        its.setdefault(s["iteration"], []).append(s["search_ms"])
    print(sys.argv[1].split('/')[-1], "ms_per_step", d["ms_per_step"], "value %.4g" % d["value"], "frac", d.get("roofline", {}).get("frac"),
          [round(sum(v) / len(v), 3) for k, v in sorted(its.items())][:5], "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
