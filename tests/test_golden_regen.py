"""Generator drift guard: the committed fixtures must be exactly what tests/golden/make_golden.py produces at HEAD.

Round 1 changed scenarios.make_scenario after the fixtures were written, and nothing noticed.  When the upstream
checkout is present (build container only -- it never travels to the GPU box) this regenerates a few small cases
into a temp dir by running the REFERENCE's modules again and compares every array with the committed file."""
import os
import subprocess
import sys

import numpy as np
import pytest

import _golden_io as gio

REF = os.environ.get("UPSTREAM_REFERENCE", "/root/reference")
CASES = ["all_s0_n16", "all_s1_n64", "modes_n32", "c1_n64", "coincident_n8", "zspread_n64"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="upstream checkout not present (GPU box)")
def test_committed_fixtures_are_what_the_generator_writes_today(tmp_path):
    gen = os.path.join(gio.GOLDEN_DIR, "make_golden.py")
    subprocess.check_call([sys.executable, gen, "--out", str(tmp_path)] + CASES, stdout=subprocess.DEVNULL)
    for name in CASES:
        new = np.load(tmp_path / (name + ".npz"), allow_pickle=False)
        old = np.load(os.path.join(gio.GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        assert sorted(new.files) == sorted(old.files), name
        for k in new.files:
            a, b = new[k], old[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (name, k)
            assert np.array_equal(a, b, equal_nan=(a.dtype.kind == "f")), (name, k)


@pytest.mark.skipif(not os.path.isdir(REF), reason="upstream checkout not present (GPU box)")
def test_baseline_c1_fixture_is_the_c1_the_bench_builds():
    """c1_n64.npz must hold the inputs baseline_scenario('c1') builds today (what bench.py --workload c1 runs)."""
    from carla_social_force_model_amd import scenarios
    sc, _ = scenarios.baseline_scenario("c1")
    case = gio.Case(os.path.join(gio.GOLDEN_DIR, "c1_n64.npz"))
    assert np.array_equal(case.loc, sc.loc) and np.array_equal(case.vel, sc.vel)
    for (c0, r0), (c1, r1) in zip(case.dynamic_obstacles, sc.dynamic_obstacles):
        assert np.array_equal(c0, c1) and np.array_equal(r0, r1)
