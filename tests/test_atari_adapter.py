"""The Atari adapter (paac_amd/atari_emulator.py) against captures of the reference's own AtariEmulator class
(atari_emulator.py:15-118) stepped by the reference's own EmulatorRunner._run (tests/golden/atari_emulator_flow.npz,
made by tests/golden/make_golden_atari.py), and its raw-screen / device-preprocessing path against its host path and
against the same captures.  ALE is absent: tests/fake_ale.py stands in for the emulator, here and in the capture."""
import argparse
import hashlib
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fake_ale import FakeALE
from oracle import preprocess as opre
from oracle.atari_flow import ReferenceFlow


def emu_args(**kw):
    d = dict(random_seed=3, rom_path="roms", game="breakout", random_start=True, single_life_episodes=False,
             visualize=False)
    d.update(kw)
    return argparse.Namespace(**d)


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "atari_emulator_flow.npz")
CASES = [(False, True), (True, False), (True, True)]        # (single_life_episodes, random_start), actors 0 and 2


def golden_case(single_life, random_start, actor):
    """One capture of the reference's own AtariEmulator + EmulatorRunner._run (tests/golden/make_golden_atari.py)."""
    z = np.load(GOLDEN)
    key = "sl%d_rs%d_a%d/" % (single_life, random_start, actor)
    rec = {k[len(key):]: z[k] for k in z.files if k.startswith(key)}
    rec["episode_frames"] = int(z["episode_frames"])
    return rec


def sha(o):
    return hashlib.sha256(np.ascontiguousarray(o).tobytes()).hexdigest()


def check_against_capture(rec, observations, rewards, terminals):
    """observations: the initial one + one per runner step (fresh initial state after a terminal)."""
    assert np.array_equal(np.asarray(rewards, dtype=np.float32), rec["rewards"])
    assert np.array_equal(np.asarray(terminals, dtype=np.bool_), rec["terminals"])
    assert rec["terminals"].sum() >= 2, "the capture must contain resets"
    assert len(observations) == len(rec["obs_sha256"])
    for k, o in enumerate(observations):
        assert o.dtype == np.uint8 and o.shape == (84, 84, 4)
        assert sha(o) == str(rec["obs_sha256"][k]), "observation %d differs from the reference's" % k
    for k, full in zip(rec["full_idx"], rec["full_obs"]):
        assert np.array_equal(observations[int(k)], full)


@pytest.mark.parametrize("single_life,random_start", CASES)
def test_oracle_flow_reproduces_the_reference_capture(single_life, random_start):
    """oracle/atari_flow.py (the restatement the -m gpu loop test checks against) is pinned to the reference's class."""
    for actor in (0, 2):
        rec = golden_case(single_life, random_start, actor)
        random.seed(int(rec["seed"]))
        flow = ReferenceFlow(actor, emu_args(single_life_episodes=single_life, random_start=random_start),
                             FakeALE(episode_frames=rec["episode_frames"]))
        obs, rewards, terminals = [flow.initial()], [], []
        for a in rec["actions"]:
            o, r, t = flow.runner_step(int(a))
            obs.append(o), rewards.append(r), terminals.append(t)
        check_against_capture(rec, obs, rewards, terminals)


@pytest.mark.parametrize("single_life,random_start", CASES)
def test_adapter_follows_reference_episode_semantics(single_life, random_start):
    """The product adapter, driven by the captured actions on the same FakeALE and `random` seed, reproduces what the
    reference's AtariEmulator + EmulatorRunner produced: every shared observation, reward and terminal flag."""
    from paac_amd.atari_emulator import AtariEmulator
    args = emu_args(single_life_episodes=single_life, random_start=random_start)
    for actor in (0, 2):
        rec = golden_case(single_life, random_start, actor)
        random.seed(int(rec["seed"]))
        emu = AtariEmulator(actor, args, ale=FakeALE(episode_frames=rec["episode_frames"]))
        assert np.array_equal(np.asarray(emu.get_legal_actions()), rec["legal_actions"])
        obs, rewards, terminals = [emu.get_initial_state()], [], []
        for a in rec["actions"]:
            o, r, t = emu.next(np.eye(4)[a])
            obs.append(emu.get_initial_state() if t else o)      # emulator_runner.py:24-31
            rewards.append(r), terminals.append(t)
        check_against_capture(rec, obs, rewards, terminals)
        assert emu.ale.options[b"repeat_action_probability"] == 0.0 and emu.ale.options[b"frame_skip"] == 1
        assert emu.ale.options[b"color_averaging"] is False and emu.get_noop() == list(rec["noop"]) == [1.0, 0.0]


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
@pytest.mark.parametrize("single_life,random_start", CASES)
def test_device_preprocessing_reproduces_the_reference_capture(single_life, random_start):
    """Raw screen pairs of the product adapter -> pinned staging -> paac_preprocess_stack (DeviceObservations, the
    --device_preprocess path) == the observations the reference's AtariEmulator + EmulatorRunner produced, resets
    included; both actors of a case are stepped as one batch of two environments."""
    torch = pytest.importorskip("torch")
    from paac_amd.atari_emulator import AtariEmulator
    from paac_amd.paac import DeviceObservations
    args = emu_args(single_life_episodes=single_life, random_start=random_start)
    recs = [golden_case(single_life, random_start, actor) for actor in (0, 2)]
    emus, firsts = [], []
    for actor, rec in zip((0, 2), recs):       # the capture seeded `random` per actor: draw each actor's first reset alone
        random.seed(int(rec["seed"]))
        emus.append(AtariEmulator(actor, args, ale=FakeALE(episode_frames=rec["episode_frames"])))
        firsts.append(emus[-1].initial_raw())
        rec["rnd"] = random.getstate()
    dobs = DeviceObservations(2, 4, torch.device("cuda", 0))
    raw = np.stack(firsts)                                         # [2, 4 slots, 2, 210, 160]
    obs = [[o] for o in dobs.update(raw, [4, 4]).cpu().numpy()]
    rewards, terminals = [[], []], [[], []]
    for k in range(len(recs[0]["actions"])):
        counts = []
        for e, (emu, rec) in enumerate(zip(emus, recs)):
            random.setstate(rec["rnd"])
            pair, r, t = emu.next_raw(np.eye(4)[rec["actions"][k]])
            if t:                                                  # emulator_runner.py:24-31: fresh initial state
                raw[e] = emu.initial_raw()
            else:
                raw[e, 0] = pair
            rec["rnd"] = random.getstate()
            counts.append(4 if t else 1)
            rewards[e].append(r), terminals[e].append(t)
        for e, o in enumerate(dobs.update(raw, counts).cpu().numpy()):
            obs[e].append(o)
    for e, rec in enumerate(recs):
        check_against_capture(rec, obs[e], rewards[e], terminals[e])


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
    # ... and against the ORACLE (oracle/atari_flow.py, pinned to the reference's own AtariEmulator by
    # test_oracle_flow_reproduces_the_reference_capture), driven by the recorded actions -- every observation the
    # device-preprocessing loop trained on, bit for bit, resets included
    random.seed(5)
    ref_args = emu_args(random_start=True, single_life_episodes=True)
    flows = [ReferenceFlow(i, ref_args, FakeALE(episode_frames=110)) for i in range(N)]
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
