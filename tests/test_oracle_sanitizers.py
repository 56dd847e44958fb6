"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the race / memory
checks of the reference's tooling have no counterpart in a Python code base; the oracle is C, so it
gets them).  oracle/selftest.c drives every entry point on problems taken from the golden fixtures;
the sanitized build must exit cleanly and print exactly what the plain build prints."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import REPO, load_golden

BUILD = os.path.join(REPO, "oracle", "_build")


def _write_problem(path, d, g, anchored=False):
    N, k = d["Y_init"][g].shape
    rng = np.random.RandomState(g)
    W = rng.randn(N, k)
    lower = np.where(np.isnan(d["G_lower"]), np.nan, d["G_lower"])
    upper = np.where(np.isnan(d["G_upper"]), np.nan, d["G_upper"])
    # goal edges: both bounds equal the goal distance wherever the goal graph has an equality edge
    goal = (d["omega"] != 0) & np.isnan(lower)
    lower = np.where(goal, np.sqrt(d["D_goal"][g]), lower)
    upper = np.where(goal, np.sqrt(d["D_goal"][g]), upper)
    at_node = np.zeros(0, dtype=np.int32)
    at_pos, at_tgt, at_kind = np.zeros((0, 3)), np.zeros(0), np.zeros(0, dtype=np.int32)
    if anchored:                      # a few synthetic point-to-anchor terms of every kind
        n_at = 9
        at_node = rng.randint(0, N, n_at).astype(np.int32)
        at_pos = rng.randn(n_at, 3)
        at_tgt = rng.rand(n_at) + 0.2
        at_kind = (1 + np.arange(n_at) % 3).astype(np.int32)
    with open(path, "wb") as f:
        f.write(struct.pack("4i", N, k, int(d["use_limits"]), len(at_node)))
        for a in (d["Y_init"][g], W, d["D_goal"][g], d["omega"], d["psi_L"], d["psi_U"], lower, upper, at_pos, at_tgt):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        f.write(at_node.tobytes())
        f.write(at_kind.tobytes())


@pytest.mark.parametrize("name,anchored", [("planar10_limits_halfpi", False), ("lwa4d", False), ("ur10", True)])
def test_oracle_is_clean_under_asan_ubsan(tmp_path, name, anchored):
    r = subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "-s", "sanitizers"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable here: " + r.stderr[-300:])
    d = load_golden(name)
    prob = tmp_path / "p.bin"
    _write_problem(prob, d, 1, anchored)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_NUM_THREADS="1")
    plain = subprocess.run([os.path.join(BUILD, "selftest"), str(prob)], capture_output=True, text=True, timeout=600)
    san = subprocess.run([os.path.join(BUILD, "selftest_san"), str(prob)], capture_output=True, text=True,
                         timeout=900, env=env)
    assert plain.returncode == 0, plain.stderr
    assert san.returncode == 0, san.stderr[-3000:]
    assert "runtime error" not in san.stderr and "AddressSanitizer" not in san.stderr, san.stderr[-3000:]
    assert san.stdout == plain.stdout and len(plain.stdout.splitlines()) == (10 if anchored else 9)
