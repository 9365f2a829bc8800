#!/usr/bin/env python3
"""Capture golden vectors by running the REFERENCE's own PAACLearner.train() loop.

Runs ONLY in the build container (needs /root/reference); the .npz it writes are what travels.
Recipe = SURVEY.md Appendix B:
  * `tensorflow` is absent, so a stub module exposing only tf.Summary / tf.summary.merge_all is
    registered; nothing in paac.py / actor_learner.py touches tf at import time.
  * the learner is built with object.__new__(paac.PAACLearner) + hand-set attributes; the TF session is
    replaced by a fake whose (v, pi) come from golden_env.FakePolicy and which records the train feed.
  * real reference Runners / EmulatorRunner worker processes step golden_env.GoldenEnv instances that
    are built on the reference's own environment.FramePool / ObservationPool; the frame-pool operation
    restates atari_emulator.py:69-75 with PIL NEAREST in place of the removed scipy.misc.imresize
    (atari_emulator itself is not importable: ale_python_interface is absent).
What the fixtures pin: rollout order, auto-reset on terminal, reward clipping, masks, the float64
n-step return scan, t-major flattening, lr schedule / global_step accounting, the sampler's MT19937
consumption, FramePool/ObservationPool ordering, max + PIL-nearest resize.
"""
import os
import sys
import types
import hashlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
sys.path.insert(0, REFERENCE)


def install_tf_stub():
    tf = types.ModuleType("tensorflow")

    class Summary(object):
        class Value(object):
            def __init__(self, **kw):
                self.kw = kw

        def __init__(self, value=None):
            self.value = value

    tf.Summary = Summary
    tf.summary = types.SimpleNamespace(merge_all=lambda: "SUMMARIES")
    sys.modules["tensorflow"] = tf


class FakeNetwork(object):
    input_ph = "input_ph"
    output_layer_v = "output_layer_v"
    output_layer_pi = "output_layer_pi"
    critic_target_ph = "critic_target_ph"
    selected_action_ph = "selected_action_ph"
    adv_actor_ph = "adv_actor_ph"


class FakeSession(object):
    def __init__(self, net, policy, learner):
        self.net, self.policy, self.learner = net, policy, learner
        self.pi_log, self.v_log, self.boot_log, self.feeds = [], [], [], []

    def run(self, fetches, feed_dict=None):
        net = self.net
        if fetches == [net.output_layer_v, net.output_layer_pi]:
            v, pi = self.policy(np.asarray(feed_dict[net.input_ph]))
            self.v_log.append(v.copy())
            self.pi_log.append(pi.copy())
            return v, pi
        if fetches is net.output_layer_v:
            v, _ = self.policy(np.asarray(feed_dict[net.input_ph]))
            self.boot_log.append(v.copy())
            return v
        if isinstance(fetches, list) and fetches[0] == "train_step":
            rec = {k: np.array(v, copy=True) for k, v in feed_dict.items()}
            rec["global_step"] = self.learner.global_step
            self.feeds.append(rec)
            return None, "S"
        raise AssertionError("unexpected fetches %r" % (fetches,))

    def close(self):
        pass


class FakeWriter(object):
    def __init__(self):
        self.episodes = []

    def add_summary(self, summary, step):
        if hasattr(summary, "value") and summary.value:
            kw = {v.kw["tag"]: v.kw["simple_value"] for v in summary.value}
            self.episodes.append((step, kw["rl/reward"], kw["rl/episode_length"]))

    def flush(self):
        pass


def process_frame_pool(frame_pool):
    """atari_emulator.py:69-75 with PIL standing in for scipy.misc.imresize(interp='nearest')."""
    from PIL import Image
    img = np.amax(frame_pool, axis=0)
    img = np.asarray(Image.fromarray(img).resize((84, 84), Image.NEAREST))
    return img.astype(np.uint8)


def capture(N, T, A, W, cycles, seed, terminal_p=0.1):
    import paac                      # the reference module
    import environment as ref_env    # the reference module
    from golden_env import GoldenEnv, FakePolicy

    L = object.__new__(paac.PAACLearner)
    net = FakeNetwork()
    policy = FakePolicy(A)
    sess = FakeSession(net, policy, L)
    writer = FakeWriter()
    L.network = net
    L.session = sess
    L.emulators = np.asarray([GoldenEnv(i, A, ref_env.FramePool, ref_env.ObservationPool, process_frame_pool,
                                        terminal_p) for i in range(N)])
    L.num_actions = A
    L.emulator_counts = N
    L.max_local_steps = T
    L.workers = W
    L.gamma = 0.99
    L.max_global_steps = cycles * N * T
    L.global_step = 0
    L.initial_lr = 0.0224
    L.lr_annealing_steps = 80000000
    L.learning_rate = "learning_rate"
    L.train_step = "train_step"
    L.init_network = lambda: 0
    L.save_vars = lambda force=False: None
    L.summary_writer = writer
    np.random.seed(seed)
    L.train()
    for r in L.runners.runners:
        r.join(timeout=10)
    state = np.random.get_state()
    out = dict(N=N, T=T, A=A, W=W, cycles=cycles, seed=seed, terminal_p=terminal_p,
               gamma=0.99, initial_lr=0.0224, lr_annealing_steps=80000000,
               pi=np.stack(sess.pi_log).reshape(cycles, T, N, A),
               v=np.stack(sess.v_log).reshape(cycles, T, N),
               v_boot=np.stack(sess.boot_log),
               states=np.stack([f[net.input_ph] for f in sess.feeds]).astype(np.uint8),
               y=np.stack([f[net.critic_target_ph] for f in sess.feeds]),
               adv=np.stack([f[net.adv_actor_ph] for f in sess.feeds]),
               actions=np.stack([f[net.selected_action_ph] for f in sess.feeds]),
               lr=np.array([float(f["learning_rate"]) for f in sess.feeds], dtype=np.float64),
               global_step=np.array([f["global_step"] for f in sess.feeds], dtype=np.int64),
               episodes=np.array(writer.episodes, dtype=np.float64).reshape(-1, 3),
               mt_pos=np.int64(state[2]),
               mt_key_sha256=hashlib.sha256(np.asarray(state[1], dtype=np.uint32).tobytes()).hexdigest())
    assert out["states"].dtype == np.uint8 and out["y"].dtype == np.float64
    return out


CONFIGS = [
    # name, N, T, A, W, cycles, seed
    ("pong_n8_t5_a6", 8, 5, 6, 2, 3, 42),
    ("breakout_n8_t5_a4", 8, 5, 4, 2, 3, 43),
    ("seaquest_n4_t20_a18", 4, 20, 18, 2, 2, 44),
    # SURVEY.md Appendix B matrix: the headline shard (32 envs, 8 workers) and a long-rollout A=18 case
    ("breakout_n32_t5_a4", 32, 5, 4, 8, 3, 45),
    ("seaquest_n16_t20_a18", 16, 20, 18, 4, 3, 46),
]

if __name__ == "__main__":
    import warnings
    warnings.simplefilter("ignore", DeprecationWarning)
    install_tf_stub()
    only = set(sys.argv[1:])
    for name, N, T, A, W, cycles, seed in CONFIGS:
        if only and name not in only:
            continue
        out = capture(N, T, A, W, cycles, seed)
        path = os.path.join(HERE, "rollout_%s.npz" % name)
        np.savez_compressed(path, **out)
        print(name, "->", path, os.path.getsize(path) // 1024, "KiB; episodes finished:", len(out["episodes"]))
