"""Compact A/B line: runs bench.py (config 3, no CPU baseline) and prints per-iteration mean search_ms + setup."""
import json, subprocess, sys, collections
steps = sys.argv[1] if len(sys.argv) > 1 else "10"
out = subprocess.run([sys.executable, "bench.py", "--steps", steps, "--warmup", "5", "--no-cpu-baseline", "--no-host-align"] + sys.argv[2:],
                     capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print("bench failed:", out.stdout[-800:], out.stderr[-1500:]); sys.exit(1)
d = json.loads(line[-1])
by = collections.defaultdict(list)
for s in d.get("per_step", []):
    by[s["iteration"]].append(s["search_ms"])
print("ms_per_step %.4f  search/iter: %s  avg_kernel %.4f frac %.4f" % (
    d["ms_per_step"], " ".join("%d:%.3f" % (k, sum(v) / len(v)) for k, v in sorted(by.items())),
    d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"]))
ov = [s["step_ms"] - s["search_ms"] for s in d.get("per_step", [])]
if ov:
    print("accumulate+finalize per iteration: %.4f ms" % (sum(ov) / len(ov)))
print("setup", {k: v for k, v in d["setup"].items() if k.endswith("_ms")})
