#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT
export GRAFT_COMMIT=${GRAFT_COMMIT:-unknown}
timeout 300 python bench.py --config 5 --points 100000000 --virtual-world 8 --virtual-rank 3 > $OUT/bench5_100M_rank3.json 2> $OUT/bench5_100M_rank3.err
timeout 400 python bench.py --config 5 --points 100000000 > $OUT/bench5_100M_1gpu.json 2> $OUT/bench5_100M_1gpu.err
for f in $OUT/bench5*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
its = {}
for s in d.get("per_step", []):
    its.setdefault(s["iteration"], []).append(s["search_ms"])
print(sys.argv[1].split('/')[-1], "ms_per_step", d["ms_per_step"], "value %.4g" % d["value"], [round(sum(v) / len(v), 3) for k, v in sorted(its.items())][:5], d.get("setup", {}).get("index_build_ms"))
PY
done
