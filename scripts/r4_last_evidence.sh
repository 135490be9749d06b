#!/bin/bash
# The short evidence run of the round's LAST commit (the cell-shape change of the index build came after
# scripts/r4_final_evidence.sh had run): the -m gpu tests, the trace + FETCH/WRITE passes of the driver's bench command
# (-> pmc_traffic.json stamped with the commit and the hash of the kernel + index-build sources), and the bench lines of
# configs 3 / 2 / 4, the rejector chain and reciprocal correspondences.  ~7 GPU-minutes.
#   GRAFT_COMMIT=$(git rev-parse --short HEAD) bash scripts/r4_last_evidence.sh [tag]
set -u
TAG=${1:-r4last}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GRAFT_COMMIT=${GRAFT_COMMIT:-unknown}
timeout 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; grep -E "passed|failed" $OUT/tests.log | tail -1
bash scripts/profile_gpu.sh $TAG "trace fetch write" > $OUT/prof.log 2>&1
cp gpurun_out/prof_$TAG/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
cp gpurun_out/prof_$TAG/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null
cp gpurun_out/prof_$TAG/summary.txt $OUT/prof_summary.txt 2>/dev/null
timeout 600 python bench.py > $OUT/bench3.json 2> $OUT/bench3.err
timeout 300 python bench.py --config 2 --no-cpu-baseline > $OUT/bench2.json 2> $OUT/bench2.err
timeout 300 python bench.py --config 4 --no-cpu-baseline > $OUT/bench4.json 2> $OUT/bench4.err
timeout 300 python bench.py --no-cpu-baseline --no-host-align --rejectors median,trimmed > $OUT/bench3_rejectors.json 2> $OUT/bench3_rejectors.err
timeout 300 python bench.py --no-cpu-baseline --no-host-align --reciprocal > $OUT/bench3_reciprocal.json 2> $OUT/bench3_reciprocal.err
for f in $OUT/bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    its = {}
    for s in d.get("per_step", []):
        its.setdefault(s["iteration"], []).append(s["search_ms"])
    print(sys.argv[1].split('/')[-1], "ms_per_step", d["ms_per_step"], "value %.4g" % d["value"], "frac", d.get("roofline", {}).get("frac"),
          "traffic", d.get("roofline", {}).get("traffic"), [round(sum(v) / len(v), 3) for k, v in sorted(its.items())][:5])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
