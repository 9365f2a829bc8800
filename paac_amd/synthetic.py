"""Synthetic environments: the metric's "synthetic 84x84x4 uint8 frames".

This is the SPEC of the synthetic environment family; the same numbers are produced
  * on the host by `SyntheticEnvironment` (a BaseEnvironment plugin, numpy), and
  * on the device by paac_synth_reset / paac_synth_step (csrc/misc.hip), N envs per launch.
The reference has no synthetic environment (it drives ALE, atari_emulator.py); this one exists so the hot
path can be measured and parity-tested without an emulator.  Call pattern mirrors AtariEmulator:
get_initial_state() / next(one_hot) -> (obs, reward, terminal)  (atari_emulator.py:88-106).

Spec (all arithmetic uint32, wrapping):
  lowbias32(x): x ^= x>>16; x *= 0x7feb352d; x ^= x>>15; x *= 0x846ca68b; x ^= x>>16
  key(seed, env, id) = lowbias32( lowbias32(seed_lo ^ lowbias32(env + 0x9E3779B9)) ^ seed_hi
                                  ^ lowbias32(id_lo*0x85EBCA6B + id_hi + 0x7F4A7C15) )
  word(key, w)       = lowbias32(key + w*0x9E3779B9 + 0x165667B1)        (4 little-endian bytes = 4 pixels)
  frame id           = number of next() calls so far (id 0 = the frame of the very first initial state)
  path A plane       pixel (y,x) = byte (x&3) of word(key, y*21 + (x>>2))
  path B raw frames  frame f, pixel (ry,rx) = byte (rx&3) of word(key ^ 0x5bd1e995, f*8400 + ry*40 + (rx>>2));
                     plane = nearest-resize(max(frame0, frame1))   (atari_emulator.py:69-75)
  reward             = [-2,0,0,1,3][(lowbias32(key ^ 0xA511E9B3) % 5 + action) % 5]   (unclipped)
  terminal           = lowbias32(key ^ 0x3C6EF372) < terminal_threshold
  observation        = previous stack shifted by one channel with the new plane as channel 3; the initial
                       state (construction and after a terminal) is [0, 0, 0, current plane].
"""
import numpy as np

from .environment import BaseEnvironment, pil_nearest_lut

M32 = np.uint64(0xFFFFFFFF)
REWARD_TABLE = np.array([-2.0, 0.0, 0.0, 1.0, 3.0], dtype=np.float32)


def lowbias32(x):
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7feb352d)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846ca68b)) & M32
    x ^= x >> np.uint64(16)
    return x


def lowbias32_int(x):
    """lowbias32 on a Python int (the per-step scalars: several times faster than numpy scalar arithmetic)."""
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def synth_key(seed, env, frame_id):
    seed = int(seed)
    frame_id = int(frame_id)
    k = lowbias32_int((seed & 0xFFFFFFFF) ^ lowbias32_int(int(env) + 0x9E3779B9))
    idh = lowbias32_int((frame_id & 0xFFFFFFFF) * 0x85EBCA6B + (frame_id >> 32) + 0x7F4A7C15)
    return lowbias32_int(k ^ ((seed >> 32) & 0xFFFFFFFF) ^ idh)


def _word_offsets(n):
    return ((np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B9) + np.uint64(0x165667B1)) & M32).astype(np.uint32)


def _lowbias32_u32(x):
    """lowbias32 in place on a uint32 array (the arithmetic wraps by itself)."""
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    return x


def synth_words(key, w):
    w = np.asarray(w, dtype=np.uint64)
    return lowbias32((np.uint64(key) + w * np.uint64(0x9E3779B9) + np.uint64(0x165667B1)) & M32).astype(np.uint32)


_PLANE_OFF = _word_offsets(84 * 21)
_RAW_OFF = _word_offsets(2 * 210 * 40)


def terminal_threshold(p):
    return int(min(max(p, 0.0), 1.0) * 4294967296.0) & 0xFFFFFFFF if p < 1.0 else 0xFFFFFFFF


def plane_a(key):
    words = _lowbias32_u32(_PLANE_OFF + np.uint32(key & 0xFFFFFFFF))          # == synth_words(key, arange(84 * 21))
    return words.astype("<u4", copy=False).view(np.uint8).reshape(84, 84)


def raw_frames_b(key):
    words = _lowbias32_u32(_RAW_OFF + np.uint32((key ^ 0x5bd1e995) & 0xFFFFFFFF))
    return words.astype("<u4", copy=False).view(np.uint8).reshape(2, 210, 160)


ROW_LUT = pil_nearest_lut(210)
COL_LUT = pil_nearest_lut(160)


class SyntheticEnvironment(BaseEnvironment):
    def __init__(self, actor_id, num_actions, seed=0, terminal_p=0.01, raw_frames=False):
        self.actor_id = int(actor_id)
        self.num_actions = int(num_actions)
        self.seed = int(seed)
        self.threshold = terminal_threshold(terminal_p)
        self.raw_frames = bool(raw_frames)
        self.frame_id = 0
        self.stack = np.zeros((84, 84, 4), dtype=np.uint8)
        self.plane = None

    def _plane(self, key):
        if self.raw_frames:
            fr = raw_frames_b(key)
            return np.maximum(fr[0], fr[1])[ROW_LUT][:, COL_LUT]
        return plane_a(key)

    def get_initial_state(self):
        if self.plane is None:
            self.plane = self._plane(synth_key(self.seed, self.actor_id, 0))
        self.stack = np.zeros((84, 84, 4), dtype=np.uint8)
        self.stack[..., 3] = self.plane
        return np.copy(self.stack)

    def next(self, action):
        a = int(np.argmax(action))
        self.frame_id += 1
        key = synth_key(self.seed, self.actor_id, self.frame_id)
        self.plane = self._plane(key)
        # shift by one channel, the new plane as channel 3: one dword = the 4 channels of a pixel (little endian)
        new32 = self.stack.view("<u4").reshape(84, 84) >> np.uint32(8)
        new32 |= self.plane.astype(np.uint32) << np.uint32(24)
        self.stack = new32.view(np.uint8).reshape(84, 84, 4)
        hr = lowbias32_int(key ^ 0xA511E9B3)
        reward = float(REWARD_TABLE[(hr % 5 + a) % 5])
        terminal = lowbias32_int(key ^ 0x3C6EF372) < self.threshold
        return np.copy(self.stack), reward, bool(terminal)

    def get_legal_actions(self):
        return np.arange(self.num_actions)

    def get_noop(self):
        return [1.0, 0.0]
