"""Ragged shapes through the device loop against the oracle (the checks of tests/test_learner_gpu.py at sizes the test
suite does not list).  usage: python tools/stress_shapes.py"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_learner_gpu as T

CASES = [("breakout", 1, 1, 2, "NATURE"), ("pong", 3, 2, 2, "NIPS"), ("seaquest", 5, 7, 2, "NATURE"),
         ("qbert", 7, 5, 2, "NIPS"), ("breakout", 17, 3, 2, "NATURE"), ("qbert", 33, 5, 2, "NATURE"),
         ("breakout", 64, 5, 1, "NATURE"), ("breakout", 65, 5, 1, "NATURE"), ("seaquest", 60, 4, 1, "NIPS"),
         ("seaquest", 61, 2, 1, "NATURE"), ("pong", 100, 2, 1, "NIPS"), ("breakout", 16, 1, 2, "NIPS"),
         ("qbert", 2, 20, 1, "NATURE"),
         # around the batch classes of the paired weight-gradient launches (65 .. 512 training rows)
         ("breakout", 13, 5, 1, "NATURE"), ("breakout", 14, 5, 2, "NATURE"), ("qbert", 100, 5, 1, "NATURE"),
         ("breakout", 103, 5, 1, "NATURE"), ("seaquest", 27, 3, 2, "NATURE")]
# round 4: every case also on path B (raw screen pairs + the preprocess launch), plus shapes around the large-shard routes
# (heads finished inside the sampler workgroups up to 8 actions; doubles made ahead; kept rows at 65 .. 256 environments)
CASES = [c + (False,) for c in CASES] + [c + (True,) for c in CASES[::3]] + [
    ("breakout", 128, 2, 1, "NATURE", False), ("qbert", 129, 2, 1, "NATURE", False), ("seaquest", 66, 3, 1, "NATURE", False),
    ("pong", 200, 2, 1, "NIPS", False), ("breakout", 255, 1, 1, "NATURE", True), ("qbert", 97, 2, 1, "NIPS", True)]
bad = 0
for c in CASES:
    try:
        T.test_device_loop_matches_oracle(*c)
        print("ok  ", c, flush=True)
    except Exception:
        bad += 1
        print("FAIL", c, flush=True)
        traceback.print_exc()
sys.exit(1 if bad else 0)
