#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE's own AtariEmulator (atari_emulator.py:15-118) stepped by the reference's
own EmulatorRunner._run (emulator_runner.py:18-33).

Runs ONLY in the build container (needs /root/reference); the .npz it writes is what travels.  Two third-party
modules the reference imports are absent here (ordinary ModuleNotFoundError / ImportError) and are stood in for:
  * `ale_python_interface.ALEInterface` -> tests/fake_ale.FakeALE (a deterministic object with ALE's Python
    interface: screens, rewards, lives and game-over are pure functions of (seed, emulator frames));
  * `scipy.misc.imresize(img, (84, 84), interp='nearest')` (removed from scipy >= 1.3) -> PIL Image.resize NEAREST, the
    code path imresize itself took (the same stand-in make_golden.py uses).
Everything else -- random-start no-ops (:60-67), action repeat + reward sum + 2-frame pool (:77-86), the 4-step initial
fill (:88-96), life-loss terminals (:108-112), FramePool / ObservationPool (environment.py:42-75), the runner's
auto-reset on terminal (emulator_runner.py:24-31) -- is the reference's code, executed.

Per case (single_life_episodes, random_start, actor_id): the Python `random` seed, the action indices, and per runner
step the shared observation (sha256; the first few and every post-reset one in full), reward, terminal flag.
"""
import argparse
import hashlib
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(HERE))       # tests/ (fake_ale)
sys.path.insert(0, REFERENCE)

STEPS = 150
EPISODE_FRAMES = 150        # FakeALE: game over after this many emulator frames
CASES = [(False, True), (True, False), (True, True)]        # (single_life_episodes, random_start)
ACTORS = (0, 2)
FULL_FIRST = 3              # observations stored in full: the first FULL_FIRST and the first FULL_RESETS post-reset ones
FULL_RESETS = 2


def install_stubs():
    from fake_ale import FakeALE
    from PIL import Image
    ale = types.ModuleType("ale_python_interface")
    ale.ALEInterface = lambda: FakeALE(episode_frames=EPISODE_FRAMES)
    sys.modules["ale_python_interface"] = ale

    def imresize(img, size, interp="nearest"):
        assert interp == "nearest" and tuple(size) == (84, 84)
        return np.asarray(Image.fromarray(img).resize((size[1], size[0]), Image.NEAREST))

    misc = types.ModuleType("scipy.misc")
    misc.imresize = imresize
    sys.modules["scipy.misc"] = misc


class ScriptedQueue(object):
    """Stands in for the worker's multiprocessing.Queue: every get() first publishes the next scripted action into the
    shared action array (what the learner does before runners.update_environments()), then hands out the token."""

    def __init__(self, actions, shared_actions):
        self.actions, self.shared, self.k = actions, shared_actions, 0

    def get(self):
        if self.k >= len(self.actions):
            return None
        self.shared[0] = np.eye(self.shared.shape[1], dtype=np.float32)[self.actions[self.k]]
        self.k += 1
        return True


class RecordingBarrier(object):
    def __init__(self, variables):
        self.variables, self.log = variables, []

    def put(self, _):
        obs, reward, over = self.variables[0][0], self.variables[1][0], self.variables[2][0]
        self.log.append((np.array(obs, dtype=np.uint8), float(reward), bool(over)))


def capture_case(single_life, random_start, actor, seed):
    import atari_emulator            # the reference module
    import emulator_runner           # the reference module
    args = argparse.Namespace(random_seed=3, rom_path="roms", game="breakout", random_start=random_start,
                              single_life_episodes=single_life, visualize=False)
    random.seed(seed)
    emu = atari_emulator.AtariEmulator(actor, args)
    A = len(emu.get_legal_actions())
    actions = np.random.RandomState(100 + actor).randint(0, A, STEPS)
    first = emu.get_initial_state()
    variables = [np.zeros((1, 84, 84, 4), dtype=np.uint8), np.zeros(1, dtype=np.float32), np.zeros(1, dtype=np.float32),
                 np.zeros((1, A), dtype=np.float32)]
    variables[0][0] = first
    barrier = RecordingBarrier(variables)
    runner = emulator_runner.EmulatorRunner(0, [emu], variables, ScriptedQueue(actions, variables[3]), barrier)
    runner._run()                    # in-process: the loop body of the worker, emulator_runner.py:18-33
    obs = [first] + [o for o, _, _ in barrier.log]
    rewards = np.array([r for _, r, _ in barrier.log], dtype=np.float32)
    overs = np.array([t for _, _, t in barrier.log], dtype=np.bool_)
    sha = np.array([hashlib.sha256(o.tobytes()).hexdigest() for o in obs])
    after_reset = [k + 1 for k in np.nonzero(overs)[0][:FULL_RESETS]]
    full_idx = np.array(sorted(set(list(range(FULL_FIRST)) + after_reset)), dtype=np.int64)
    return dict(seed=np.int64(seed), actions=actions.astype(np.int32), rewards=rewards, terminals=overs, obs_sha256=sha,
                full_idx=full_idx, full_obs=np.stack([obs[k] for k in full_idx]), noop=np.array(emu.get_noop()),
                legal_actions=np.asarray(emu.get_legal_actions()))


if __name__ == "__main__":
    install_stubs()
    out = dict(steps=np.int64(STEPS), episode_frames=np.int64(EPISODE_FRAMES))
    for single_life, random_start in CASES:
        for actor in ACTORS:
            key = "sl%d_rs%d_a%d" % (single_life, random_start, actor)
            rec = capture_case(single_life, random_start, actor, seed=11 + actor)
            for k, v in rec.items():
                out["%s/%s" % (key, k)] = v
            print(key, "resets:", int(rec["terminals"].sum()), "reward sum:", float(rec["rewards"].sum()))
    path = os.path.join(HERE, "atari_emulator_flow.npz")
    np.savez_compressed(path, **out)
    print("->", path, os.path.getsize(path) // 1024, "KiB")
