"""Deterministic fake environment + fake policy used to capture golden vectors from the reference's
own train() loop (make_golden.py) and to replay them against the oracle / the HIP path (tests).

This file is the build's own code (no reference source).  The frame / observation pool classes are
INJECTED: make_golden.py passes the reference's environment.FramePool / ObservationPool, the tests
pass the oracle's or the product's restatement, so the fixture pins those classes' behaviour.
"""
import numpy as np

RAW_H, RAW_W = 210, 160


def blocky_frame(rs):
    """Atari-like raw gray frame: flat background, a few rectangles, a sprinkle of single pixels
    (compresses well, still exercises every row/column of the nearest-resize LUT over many frames)."""
    f = np.full((RAW_H, RAW_W), rs.randint(0, 256), dtype=np.uint8)
    for _ in range(6):
        y0, x0 = rs.randint(0, RAW_H - 4), rs.randint(0, RAW_W - 4)
        h, w = rs.randint(1, 70), rs.randint(1, 70)
        f[y0:y0 + h, x0:x0 + w] = rs.randint(0, 256)
    ys = rs.randint(0, RAW_H, 64)
    xs = rs.randint(0, RAW_W, 64)
    f[ys, xs] = rs.randint(0, 256, 64).astype(np.uint8)
    return f


class GoldenEnv(object):
    """BaseEnvironment-shaped (duck typed) env mirroring AtariEmulator's call pattern
    (atari_emulator.py:77-106): action repeat pushes the last 2 raw frames into the frame pool, the
    processed frame goes into the observation pool, get_initial_state does 4 no-op action repeats."""

    def __init__(self, i, num_actions, frame_pool_cls, observation_pool_cls, process_op, terminal_p=0.1):
        self.rs = np.random.RandomState(1000 + i)
        self.num_actions = num_actions
        self.terminal_p = terminal_p
        self.observation_pool = observation_pool_cls(np.zeros((84, 84, 4), dtype=np.uint8))
        self.frame_pool = frame_pool_cls(np.empty((2, RAW_H, RAW_W), dtype=np.uint8), process_op)
        self.raw_log = None     # set to a list to record raw frame pairs

    def _action_repeat(self, a):
        reward = 0.0
        pair = []
        for _ in range(2):
            reward += float(self.rs.choice([-2.0, 0.0, 0.0, 1.0, 3.0])) * (1.0 if (a % 2 == 0) else 0.5)
            fr = blocky_frame(self.rs)
            pair.append(fr)
            self.frame_pool.new_frame(fr)
        if self.raw_log is not None:
            self.raw_log.append(np.stack(pair))
        return reward

    def get_initial_state(self):
        for _ in range(4):
            self._action_repeat(0)
            self.observation_pool.new_observation(self.frame_pool.get_processed_frame())
        return self.observation_pool.get_pooled_observations()

    def next(self, action):
        reward = self._action_repeat(int(np.argmax(action)))
        self.observation_pool.new_observation(self.frame_pool.get_processed_frame())
        terminal = bool(self.rs.rand() < self.terminal_p)
        return self.observation_pool.get_pooled_observations(), reward, terminal

    def get_legal_actions(self):
        return np.arange(self.num_actions)

    def get_noop(self):
        return [1.0, 0.0]


class FakePolicy(object):
    """Deterministic stand-in for the TF session's (v, pi): a tiny linear net on a strided
    subsample of the state.  Its OUTPUTS are stored in the fixture (they are inputs of the path
    under test), so replay never depends on BLAS rounding."""

    def __init__(self, num_actions, seed=7):
        rs = np.random.RandomState(seed)
        self.w = (rs.randn(12 * 12 * 4, num_actions) * 0.15).astype(np.float32)
        self.wv = (rs.randn(12 * 12 * 4) * 0.05).astype(np.float32)

    def __call__(self, states):
        x = np.asarray(states)[:, ::7, ::7, :].reshape(len(states), -1).astype(np.float32) / np.float32(255.0)
        logits = x @ self.w
        logits = logits - logits.max(axis=1, keepdims=True)
        e = np.exp(logits)
        pi = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
        v = (x @ self.wv).astype(np.float32)
        return v, pi
