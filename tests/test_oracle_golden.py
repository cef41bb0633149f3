"""The oracle reproduces its committed closest-hit vectors (tests/golden/oracle_golden.npz, written by
tests/golden/make_oracle_golden.py).  A regression guard for the checker itself: the vectors are the oracle's own
earlier outputs, not the reference's -- the reference ships none."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_oracle_golden", os.path.join(ROOT, "tests", "golden", "make_oracle_golden.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)


@pytest.fixture(scope="module")
def golden():
    return np.load(gen.OUT, allow_pickle=False)


@pytest.mark.parametrize("name,seed,small,large", gen.CASES)
def test_oracle_reproduces_committed_hits(oracle, golden, name, seed, small, large):
    got = gen.records(name, seed, small, large)
    for key, value in got.items():
        want = golden[f"{name}/{key}"]
        assert value.shape == want.shape and np.array_equal(value, want), f"{name}/{key} drifted"


def test_benchmark_scene_fixture_is_reproduced():
    """The seeded generator and the host flattener still produce the committed benchmark scene (tests/golden/bouncing_seed1.npz:
    the scene as it crosses the C ABI, and the camera constants at 1920x1080), bit for bit."""
    sp = importlib.util.spec_from_file_location("make_scene_fixture", os.path.join(ROOT, "tests", "golden", "make_scene_fixture.py"))
    fx = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(fx)
    want = np.load(fx.OUT, allow_pickle=False)
    got = fx.records()
    assert sorted(want.files) == sorted(got)
    for key, value in got.items():
        assert value.shape == want[key].shape and np.array_equal(value, want[key]), f"{key} drifted"
    assert int(want["n_spheres"][0]) == 484


def test_packaged_earth_raster_is_the_golden_one():
    a = open(os.path.join(ROOT, "tests", "golden", "earthmap_rgb8.png"), "rb").read()
    b = open(os.path.join(ROOT, "ray-tracer_amd", "assets", "earthmap_rgb8.png"), "rb").read()
    assert a == b
