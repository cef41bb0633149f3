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


def test_packaged_earth_raster_is_the_golden_one():
    a = open(os.path.join(ROOT, "tests", "golden", "earthmap_rgb8.png"), "rb").read()
    b = open(os.path.join(ROOT, "ray-tracer_amd", "assets", "earthmap_rgb8.png"), "rb").read()
    assert a == b
