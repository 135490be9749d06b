#!/bin/bash
# Round 4, second GPU call: the trimmed default build (greedy seeds in, variants out), the new rejector selection, the
# fused radius normals, the device-side partition at 100M points.  ~8 GPU-minutes.
#   GRAFT_COMMIT=$(git rev-parse --short HEAD) bash scripts/r4_call2.sh  -> gpurun_out/r4call2/
set -u
export GRAFT_COMMIT=${GRAFT_COMMIT:-unknown}
OUT=gpurun_out/r4call2
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
timeout 300 python bench.py --no-cpu-baseline --no-host-align > $OUT/bench3.json 2> $OUT/bench3.err
timeout 300 python bench.py --no-cpu-baseline --no-host-align --rejectors median,trimmed > $OUT/bench3_rejectors.json 2> $OUT/bench3_rejectors.err
timeout 300 python bench.py --no-cpu-baseline --no-host-align --reciprocal > $OUT/bench3_reciprocal.json 2> $OUT/bench3_reciprocal.err
timeout 300 python bench.py --config 2 --no-cpu-baseline --no-host-align > $OUT/bench2.json 2> $OUT/bench2.err
timeout 300 python bench.py --config 4 --no-cpu-baseline > $OUT/bench4.json 2> $OUT/bench4.err
timeout 300 python scratch/radius_probe.py 10000000 > $OUT/radius_probe.log 2>&1; grep -v "^/opt" $OUT/radius_probe.log
# kernel trace of the rejector line (where the chain's time goes now)
bash scripts/profile_gpu.sh r4rej "trace" --steps 20 --warmup 5 --no-cpu-baseline --no-host-align --rejectors median,trimmed > $OUT/prof_rej.log 2>&1
head -30 gpurun_out/prof_r4rej/summary.txt > $OUT/prof_rej_summary.txt
# config 5, one rank of eight of the 100M-point job, clouds generated and cut on the device
timeout 600 python bench.py --config 5 --points 100000000 --virtual-world 8 --virtual-rank 3 --steps 20 --warmup 5 > $OUT/bench5_r3.json 2> $OUT/bench5_r3.err
tail -3 $OUT/bench5_r3.err
for f in $OUT/bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    its = {}
    for s in d.get("per_step", []):
        its.setdefault(s["iteration"], []).append((s["search_ms"], s["step_ms"]))
    print(sys.argv[1].split('/')[-1], "ms_per_step", d["ms_per_step"], "frac", d.get("roofline", {}).get("frac"),
          [(round(sum(x[0] for x in v) / len(v), 3), round(sum(x[1] for x in v) / len(v), 3)) for k, v in sorted(its.items())][:6],
          {k: v for k, v in d.get("setup", {}).items() if k in ("shard_setup_s", "synth_gen_s", "index_points")}, d.get("stages_ms"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
