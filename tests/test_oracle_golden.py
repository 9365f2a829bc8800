"""The oracle's rollout/return/sampler/pool restatement vs vectors captured from the reference's
own train() loop (tests/golden/make_golden.py).  Bit-exact for u8/int, exact for the f64 scans."""
import glob
import hashlib
import os

import numpy as np
import pytest

from golden_env import GoldenEnv
from oracle import preprocess, rollout, sampler

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "rollout_*.npz")))


def replay_oracle(g, sample_mode="restated"):
    N, T, A = int(g["N"]), int(g["T"]), int(g["A"])
    envs = [GoldenEnv(i, A, preprocess.FramePoolOracle, preprocess.ObservationPoolOracle,
                      preprocess.max_resize, float(g["terminal_p"])) for i in range(N)]
    rs = np.random.RandomState(int(g["seed"]))
    calls = {"act": 0, "cycle": 0}

    def policy_fn(states):
        c, t = calls["cycle"], calls["act"]
        if t < T:
            calls["act"] += 1
            return g["v"][c, t], g["pi"][c, t]
        calls["act"] = 0
        calls["cycle"] += 1
        return g["v_boot"][c], None

    if sample_mode == "restated":
        sample_fn = lambda pi: sampler.sample_mt_restated(pi, rs)[0]
    else:
        sample_fn = lambda pi: sampler.sample_numpy_reference(pi, rs)
    ro = rollout.OracleRollout(envs, A, T, float(g["gamma"]), float(g["initial_lr"]),
                               int(g["lr_annealing_steps"]), policy_fn, sample_fn)
    cycles = [ro.cycle() for _ in range(int(g["cycles"]))]
    return ro, cycles, rs


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
@pytest.mark.parametrize("mode", ["restated", "numpy"])
def test_oracle_matches_reference_capture(path, mode):
    g = np.load(path)
    ro, cycles, rs = replay_oracle(g, mode)
    for c, cyc in enumerate(cycles):
        assert np.array_equal(cyc["states"], g["states"][c]), "stacked u8 states differ (cycle %d)" % c
        assert np.array_equal(cyc["actions"], g["actions"][c]), "sampled actions differ"
        assert np.array_equal(cyc["y"], g["y"][c]), "float64 n-step returns differ"
        assert np.array_equal(cyc["adv"], g["adv"][c]), "float64 advantages differ"
        assert cyc["lr"] == float(g["lr"][c])
        assert cyc["global_step"] == int(g["global_step"][c])
    st = rs.get_state()
    assert int(st[2]) == int(g["mt_pos"])
    assert hashlib.sha256(np.asarray(st[1], dtype=np.uint32).tobytes()).hexdigest() == str(g["mt_key_sha256"])
    eps = np.array(ro.finished_episodes, dtype=np.float64).reshape(-1, 3)
    assert np.array_equal(eps, g["episodes"])


def test_golden_present():
    assert len(GOLDEN) >= 5      # incl. the SURVEY Appendix B matrix: (32,5,4,8) and (16,20,18,4)
