"""Oracle: frame preprocessing and 4-frame stacking (TEST INFRASTRUCTURE, see __init__).

Follows:
  atari_emulator.py:69-75  img = amax(frame_pool, axis=0); imresize(img,(84,84),'nearest'); u8
  environment.py:42-55     FramePool: 2-slot ring written round-robin, reduced by `operation`
  environment.py:58-75     ObservationPool: write slot `idx`, idx=(idx+1)%4, read-out rotated so
                           channel 0 is the oldest and channel 3 the newest frame
scipy.misc.imresize(interp='nearest') == PIL Image.resize((84,84), NEAREST) (third-party,
PIL is installed; the LUTs below are regenerated from PIL in
tests/test_cabi_and_hostlogic.py::test_oracle_luts_regenerate_from_pil and pinned to the reference's own
stacked states by tests/test_oracle_golden.py).
The reference converts RGB->gray inside ALE (atari_emulator.py:51; third-party C++, absent):
`rgb_to_gray` is the build's own spec (ITU-R 601 fixed point == PIL convert('L')); parity unpinned.
"""
import numpy as np

RAW_H, RAW_W = 210, 160
OUT_H, OUT_W = 84, 84


def _pil_nearest_lut(src, dst):
    """PIL's ImagingScaleAffine nearest path as exercised by Image.resize(NEAREST): the source
    position starts at 0.5*scale and is ACCUMULATED in double precision (xo += scale), then
    truncated -- not floor((x+0.5)*scale): for 160->84 columns 52 and 73 land on 99 / 139
    where the closed form gives 100 / 140.  Verified against PIL 12.2."""
    scale = src / float(dst)
    xo = 0.0 + scale * 0.5
    lut = np.empty(dst, dtype=np.int32)
    for x in range(dst):
        lut[x] = int(xo)
        xo += scale
    return np.clip(lut, 0, src - 1)


ROW_LUT = _pil_nearest_lut(RAW_H, OUT_H)
COL_LUT = _pil_nearest_lut(RAW_W, OUT_W)


def max_resize(frame_pool):
    """[2,210,160] u8 -> [84,84] u8.  atari_emulator.py:69-75."""
    img = np.amax(frame_pool, axis=0)
    return img[ROW_LUT][:, COL_LUT].astype(np.uint8)


def rgb_to_gray(rgb):
    """[...,3] u8 -> [...] u8, L = (19595 R + 38470 G + 7471 B + 32768) >> 16 (PIL 'L')."""
    r = rgb[..., 0].astype(np.uint32)
    g = rgb[..., 1].astype(np.uint32)
    b = rgb[..., 2].astype(np.uint32)
    return ((r * 19595 + g * 38470 + b * 7471 + 32768) >> 16).astype(np.uint8)


def push_observation(stack, plane):
    """ObservationPool.new_observation + get_pooled_observations restated on the rotated view:
    stack [84,84,4] (oldest..newest) -> drop channel 0, append `plane` as channel 3."""
    out = np.empty_like(stack)
    out[..., :3] = stack[..., 1:]
    out[..., 3] = plane
    return out


class FramePoolOracle:
    """environment.py:42-55 restated (same constructor shape as the reference class)."""
    def __init__(self, frame_pool=None, operation=None):
        self.pool = np.zeros((2, RAW_H, RAW_W), dtype=np.uint8) if frame_pool is None else frame_pool
        self.idx = 0
        self.operation = max_resize if operation is None else operation

    def new_frame(self, frame):
        self.pool[self.idx] = frame
        self.idx = (self.idx + 1) % self.pool.shape[0]

    def get_processed_frame(self):
        return self.operation(self.pool)


class ObservationPoolOracle:
    """environment.py:58-75 restated without the ring: the stack is kept already rotated
    (channel 0 oldest .. channel 3 newest) and shifted on every push."""
    def __init__(self, observation_pool=None):
        self.stack = np.zeros((OUT_H, OUT_W, 4), dtype=np.uint8) if observation_pool is None else observation_pool

    def new_observation(self, observation):
        self.stack = push_observation(self.stack, observation)

    def get_pooled_observations(self):
        return np.copy(self.stack)
