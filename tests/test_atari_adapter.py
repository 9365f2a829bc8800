"""The Atari adapter (paac_amd/atari_emulator.py) against the reference's episode semantics
(atari_emulator.py:60-112) restated with the oracle's pools, and its raw-screen / device-preprocessing path against
its host path.  ALE is absent: tests/fake_ale.py stands in for the emulator."""
import argparse
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fake_ale import FakeALE
from oracle import preprocess as opre


def emu_args(**kw):
    d = dict(random_seed=3, rom_path="roms", game="breakout", random_start=True, single_life_episodes=False,
             visualize=False)
    d.update(kw)
    return argparse.Namespace(**d)


class ReferenceFlow(object):
    """atari_emulator.py:60-112 restated step by step on the oracle's FramePool / ObservationPool / max_resize."""

    def __init__(self, actor_id, args, ale=None):
        self.ale = FakeALE() if ale is None else ale
        self.ale.setInt(b"random_seed", args.random_seed * (actor_id + 1))
        self.legal = self.ale.getMinimalActionSet()
        self.args = args
        self.lives = self.ale.lives()
        self.frames = opre.FramePoolOracle()
        self.obs = opre.ObservationPoolOracle()

    def _screen(self):
        g = np.zeros((210, 160, 1), dtype=np.uint8)
        self.ale.getScreenGrayscale(g)
        return g[..., 0]

    def _repeat(self, a):
        r = 0
        for _ in range(2):
            r += self.ale.act(self.legal[a])
        for _ in range(2):
            r += self.ale.act(self.legal[a])
            self.frames.new_frame(self._screen())
        return r

    def _terminal(self):
        if self.args.single_life_episodes:
            return self.ale.game_over() or self.lives > self.ale.lives()
        return self.ale.game_over()

    def initial(self):
        self.ale.reset_game()
        self.lives = self.ale.lives()
        if self.args.random_start:
            for _ in range(random.randint(0, 30)):
                self.ale.act(self.legal[0])
        for _ in range(4):
            self._repeat(0)
            self.obs.new_observation(self.frames.get_processed_frame())
        return self.obs.get_pooled_observations()

    def next(self, a):
        r = self._repeat(a)
        self.obs.new_observation(self.frames.get_processed_frame())
        term = self._terminal()
        self.lives = self.ale.lives()
        return self.obs.get_pooled_observations(), r, term


@pytest.mark.parametrize("single_life,random_start", [(False, True), (True, False), (True, True)])
def test_adapter_follows_reference_episode_semantics(single_life, random_start):
    from paac_amd.atari_emulator import AtariEmulator
    args = emu_args(single_life_episodes=single_life, random_start=random_start)
    for actor in (0, 2):
        random.seed(11 + actor)
        emu = AtariEmulator(actor, args, ale=FakeALE())
        got = [emu.get_initial_state()]
        rs = np.random.RandomState(actor)
        actions = rs.randint(0, 4, 120)
        trace = []
        for a in actions:
            o, r, t = emu.next(np.eye(4)[a])
            trace.append((r, t))
            got.append(emu.get_initial_state() if t else o)
        random.seed(11 + actor)
        ref = ReferenceFlow(actor, args)
        want = [ref.initial()]
        for a, (r_got, t_got) in zip(actions, trace):
            o, r, t = ref.next(a)
            assert (r, t) == (r_got, t_got)
            want.append(ref.initial() if t else o)
        assert sum(t for _, t in trace) >= 2, "the trace must contain resets"
        for k, (g, w) in enumerate(zip(got, want)):
            assert g.dtype == np.uint8 and g.shape == (84, 84, 4)
            assert np.array_equal(g, w), "observation %d differs" % k
        assert emu.ale.options[b"repeat_action_probability"] == 0.0 and emu.ale.options[b"frame_skip"] == 1
        assert emu.ale.options[b"color_averaging"] is False and emu.get_noop() == [1.0, 0.0]


def test_raw_screens_rebuild_the_host_observations():
    from paac_amd.atari_emulator import AtariEmulator
    from paac_amd.environment import max_resize_84
    args = emu_args(random_start=False)
    a_host = AtariEmulator(1, args, ale=FakeALE())
    a_raw = AtariEmulator(1, args, ale=FakeALE())
    want = a_host.get_initial_state()
    stack = np.zeros((84, 84, 4), dtype=np.uint8)
    for pair in a_raw.initial_raw():
        stack = opre.push_observation(stack, max_resize_84(pair))
    assert np.array_equal(stack, want)
    for a in np.random.RandomState(0).randint(0, 4, 30):
        o, r, t = a_host.next(np.eye(4)[a])
        pair, r2, t2 = a_raw.next_raw(np.eye(4)[a])
        stack = opre.push_observation(stack, opre.max_resize(pair))
        assert (r, t) == (r2, t2) and np.array_equal(stack, o)


def test_runner_protocol_for_raw_screens():
    from paac_amd.atari_emulator import AtariEmulator
    from paac_amd.runners import RawEmulatorRunner, Runners
    args = emu_args(random_start=False)
    N, A = 4, 4
    emus = [AtariEmulator(i, args, ale=FakeALE(episode_frames=70)) for i in range(N)]
    first = np.stack([e.initial_raw() for e in emus])
    variables = [first, np.full(N, 4, dtype=np.float32), np.zeros(N, dtype=np.float32), np.zeros(N, dtype=np.float32),
                 np.zeros((N, A), dtype=np.float32)]
    runners = Runners(RawEmulatorRunner, emus, 0, variables)
    raw, counts, rewards, overs, actions = runners.get_shared_variables()
    seen_reset = False
    for step in range(24):
        actions[...] = np.eye(A, dtype=np.float32)[np.random.RandomState(step).randint(0, A, N)]
        runners.update_environments()
        runners.wait_updated()
        for i in range(N):
            assert counts[i] == (4 if overs[i] else 1)
            seen_reset |= bool(overs[i])
    assert seen_reset


@pytest.mark.gpu
def test_device_preprocessing_loop_equals_host_loop():
    """PAACLearner's host loop with --device_preprocess (raw screens -> pinned staging -> paac_preprocess_stack) sees
    the same observations, actions and returns as the plain host loop on the same emulators."""
    torch = pytest.importorskip("torch")
    import tempfile
    from oracle import network as onet
    from paac_amd import train
    from paac_amd.atari_emulator import AtariEmulator
    from paac_amd.paac import PAACLearner
    N, T, cycles = 4, 5, 6

    class Creator(object):
        num_actions = 4

        def __init__(self, args):
            self.create_environment = lambda i: AtariEmulator(i, args, ale=FakeALE(episode_frames=110))

    feeds = {}
    for mode in (False, True):
        args = train.get_arg_parser().parse_args([])
        args.debugging_folder = tempfile.mkdtemp(prefix="paac_atari_")
        args.game, args.arch = "breakout", "NATURE"
        args.emulator_counts, args.emulator_workers, args.max_local_steps = N, 0, T
        args.max_global_steps = cycles * N * T
        args.random_start, args.single_life_episodes, args.visualize = True, True, False
        args.device_preprocess, args.record_feeds = mode, True
        feeds[mode] = []
        args.feed_callback = feeds[mode].append
        network_creator, _ = train.get_network_and_environment_creator(args)
        args.num_actions = 4
        random.seed(5)
        np.random.seed(7)
        learner = PAACLearner(network_creator, Creator(args), args)
        learner.network.set_parameters(onet.init_params("NATURE", 4, np.random.RandomState(0), dtype=np.float32))
        learner.network.init = lambda folder, saver, session: 0
        learner.train()
    assert len(feeds[True]) == len(feeds[False]) == cycles
    # 110-frame episodes: every emulator is reset at least once inside the 6 cycles (16-46 reset frames + 4 per step)
    for c, (a, b) in enumerate(zip(feeds[False], feeds[True])):
        assert np.array_equal(a["states"], b["states"]), "cycle %d: observations differ" % c
        assert np.array_equal(a["actions"], b["actions"]) and np.array_equal(a["y"], b["y"])
    # ... and against the ORACLE: the reference's episode flow (atari_emulator.py:60-112, emulator_runner.py:24-31)
    # restated on the oracle's FramePool / ObservationPool / PIL-nearest LUT, driven by the recorded actions -- every
    # observation the device-preprocessing loop trained on, bit for bit, resets included
    random.seed(5)
    ref_args = emu_args(random_start=True, single_life_episodes=True)
    flows = [ReferenceFlow(i, ref_args, ale=FakeALE(episode_frames=110)) for i in range(N)]
    shared = [f.initial() for f in flows]
    resets = 0
    for c, feed in enumerate(feeds[True]):
        states = feed["states"].reshape(T, N, 84, 84, 4)
        actions = feed["actions"].reshape(T, N)
        for t in range(T):
            for e in range(N):
                assert np.array_equal(states[t, e], shared[e]), "cycle %d step %d env %d" % (c, t, e)
            for e in range(N):                   # emulator_runner.py:24-31: step, auto-reset on terminal
                obs, _, over = flows[e].next(int(actions[t, e]))
                shared[e] = flows[e].initial() if over else obs
                resets += bool(over)
    assert resets >= N


def test_evaluation_loop_freezes_finished_environments(monkeypatch):
    """paac_amd.test.evaluate: every environment plays ONE episode; its score stops changing once it is over (the
    reference's loop, test.py:77-83, keeps stepping and adding until all happen to finish on the same step)."""
    import paac_amd.test as harness
    from paac_amd.atari_emulator import AtariEmulator

    class Creator(object):
        num_actions = 4

        def __init__(self):
            self.made = []

        def create_environment(self, i):
            env = AtariEmulator(i, emu_args(random_start=False), ale=FakeALE(episode_frames=60 + 24 * i))
            self.made.append(env)
            return env

    steps = {"n": 0}

    def fake_choose(network, num_actions, states, session):
        steps["n"] += 1
        assert states.shape == (3, 84, 84, 4) and states.dtype == np.uint8
        idx = np.full(3, steps["n"] % num_actions)
        return np.eye(num_actions)[idx], np.zeros(3), np.full((3, num_actions), 0.25)

    monkeypatch.setattr(harness.PAACLearner, "choose_next_actions", staticmethod(fake_choose))
    random.seed(2)
    creator = Creator()
    rewards = harness.evaluate(network=None, env_creator=creator, session=None, test_count=3, noops=4)
    assert rewards.shape == (3,) and rewards.dtype == np.float32
    # episodes of 60, 84, 108 emulator frames: 16 reset frames, then 4 per step (no-ops included) -> the longest one
    # decides how many policy steps were taken, the shorter ones were not stepped past their end
    assert all(env.ale.game_over() for env in creator.made)
    frames = [env.ale.t - env.ale.start for env in creator.made]
    assert frames == [60, 84, 108], frames
    assert 10 <= steps["n"] <= 23
