"""Environment plugin surface (mirrors reference environment.py:4-75).

`BaseEnvironment` is the interface a user environment implements; `FramePool` / `ObservationPool` are the
host-side helpers an emulator wrapper uses (atari_emulator.py:37-43).  On the device the same two pools are
one kernel (paac_preprocess_stack): the 4-frame history is kept already rotated (channel 0 oldest,
channel 3 newest) and shifted on every push, which is what ObservationPool's ring + rotated read-out
returns (environment.py:66-71).
"""
import numpy as np


def pil_nearest_lut(src, dst=84):
    """Source index of every destination pixel under PIL's NEAREST resize, which is what
    scipy.misc.imresize(img, (84, 84), interp='nearest') (atari_emulator.py:73) evaluates: the source position is
    ACCUMULATED in double precision (xo += scale), not recomputed per pixel -- 160 -> 84 columns 52 and 73 land on
    99 / 139."""
    scale = src / float(dst)
    xo = 0.0 + scale * 0.5
    lut = np.empty(dst, dtype=np.int64)
    for x in range(dst):
        lut[x] = int(xo)
        xo += scale
    return lut


_LUTS = {}


def max_resize_84(frames):
    """atari_emulator.py:69-75: element-wise max over the pooled screens [k,H,W], then nearest resize to 84x84 u8."""
    frames = np.asarray(frames)
    h, w = frames.shape[-2:]
    if (h, w) not in _LUTS:
        _LUTS[(h, w)] = (pil_nearest_lut(h), pil_nearest_lut(w))
    rows, cols = _LUTS[(h, w)]
    return np.amax(frames, axis=0)[rows][:, cols].astype(np.uint8)


class BaseEnvironment(object):
    def get_initial_state(self):
        """Sets the environment to its initial state; returns uint8 [84,84,4]."""
        raise NotImplementedError()

    def next(self, action):
        """Applies `action` (one-hot vector); returns (observation uint8[84,84,4], reward, is_terminal)."""
        raise NotImplementedError()

    def get_legal_actions(self):
        raise NotImplementedError()

    def get_noop(self):
        raise NotImplementedError()

    def on_new_frame(self, frame):
        pass


class FramePool(object):
    """Round-robin pool of the last raw frames reduced by `operation` (environment.py:42-55)."""

    def __init__(self, frame_pool, operation):
        self.frame_pool = frame_pool
        self.frame_pool_index = 0
        self.frames_in_pool = frame_pool.shape[0]
        self.operation = operation

    def new_frame(self, frame):
        self.frame_pool[self.frame_pool_index] = frame
        self.frame_pool_index = (self.frame_pool_index + 1) % self.frames_in_pool

    def get_processed_frame(self):
        return self.operation(self.frame_pool)


class ObservationPool(object):
    """History of the last `pool_size` processed frames, oldest first (environment.py:58-75)."""

    def __init__(self, observation_pool):
        self.observation_pool = observation_pool
        self.pool_size = observation_pool.shape[-1]

    def new_observation(self, observation):
        self.observation_pool[..., :-1] = self.observation_pool[..., 1:]
        self.observation_pool[..., -1] = observation

    def get_pooled_observations(self):
        return np.copy(self.observation_pool)
