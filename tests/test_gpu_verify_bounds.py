"""The two INEXACT bounds of the search -- the per-lane disc bound and the reach filters of the stand-off launch -- checked
where they cull, not by their outcome.

pcl_amd/variants/libpclhip_verify.so is the library with search.hip compiled -DPCLHIP_VERIFY_BOUNDS (built by
__graft_entry__.build(); scripts/build_variant.sh verify "-DPCLHIP_VERIFY_BOUNDS"): every time a disc bound or a reach
filter drops a leaf for a lane, that lane evaluates the leaf's true minimum distance and the claim ("nothing of this leaf
within my bound") is counted, a broken claim separately (traverse.hpp: verify_culled_leaf).  The worker
(tests/verify_bounds_worker.py) runs in a subprocess with PCLHIP_LIB pointing at that build: the product library of this
process is never swapped.  Mutation check (done by hand on the emulation, profiles/r05_verify_bounds.txt): the same build
with the disc bound inflated by 1.3 reports thousands of broken claims -- also in a scenario whose results still equal the
oracle's, which is the point of checking the bound itself.

Sizes: the four cloud families of pcl_amd/synth.py at 10M points (the depth of tree and the stand-offs the bench has; one
of them at 2.5M: the other parity of four-way rounds, 256-point cells are strips there), and the degenerate inputs the
allowances of the disc bound exist for at 200k points with the results compared with the oracle as well.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VERIFY_LIB = os.path.join(ROOT, "pcl_amd", "variants", "libpclhip_verify.so")


def run_worker(lib, n, families, timeout=1500, extra_env=None):
    env = dict(os.environ, PCLHIP_LIB=lib)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "verify_bounds_worker.py"), str(n)] + list(families),
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(rows) == 5 * len(families), r.stdout[-2000:]
    return rows


def check(rows, need_oracle):
    total = sum(r["checks"] for r in rows)
    broken = [r for r in rows if r["violations"] != 0]
    print("verify-bounds: %d claims checked over %d scenarios, %d broken" % (total, len(rows), sum(r["violations"] for r in rows)))
    assert not broken, broken
    for fam in {r["family"] for r in rows}:
        assert sum(r["checks"] for r in rows if r["family"] == fam) > 0, fam     # the instrumented paths did run
    if need_oracle:
        assert all(r["matches_vs_oracle"] is True for r in rows), [r for r in rows if not r["matches_vs_oracle"]]


@pytest.fixture(scope="module")
def verify_lib():
    if not os.path.exists(VERIFY_LIB):
        pytest.fail("pcl_amd/variants/libpclhip_verify.so is missing: __graft_entry__.build() builds it")
    return VERIFY_LIB


def test_no_culled_leaf_holds_a_needed_point_degenerate_inputs(verify_lib):
    check(run_worker(verify_lib, 200_000, ["sheet", "collinear", "coincident", "far", "mm"]), need_oracle=True)


@pytest.mark.parametrize("n,families", [(10_000_000, ["sheet", "cube"]), (10_000_000, ["layers"]), (2_500_000, ["clusters", "sheet"])])
def test_no_culled_leaf_holds_a_needed_point_at_bench_size(verify_lib, n, families):
    check(run_worker(verify_lib, n, families), need_oracle=False)
