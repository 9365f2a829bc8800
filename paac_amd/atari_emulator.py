"""Atari environment adapter over an ALE-compatible backend (the role of reference atari_emulator.py:14-118).

Episode semantics follow the reference exactly, so that a policy trained / evaluated here sees what it would see there:
  * one `next()` = the action applied 4 times, reward = sum over the 4 emulator frames, and only the LAST TWO screens
    are kept for the observation (:77-86);
  * reset = new game, then -- with `random_start` -- up to 30 no-op emulator frames (:60-67), then four no-op
    `next()`-sized repeats whose pooled screens fill the 4-deep history (:88-96);
  * terminal = game over, or a lost life when `single_life_episodes` (:108-112); `get_noop()` is [1.0, 0.0] (:117-118).

Observations can be produced in two places:
  * host -- `get_initial_state()` / `next()` (the BaseEnvironment contract): max of the two kept screens, PIL-nearest
    210x160 -> 84x84 (:69-75), 4-deep history (environment.py:58-75), all numpy;
  * device -- `initial_raw()` / `next_raw()` hand out the two kept RAW screens (u8 [2,210,160]) and
    paac_preprocess_stack does max + resize + history on the GPU for all environments in one launch (PAACLearner's
    raw-frame host loop, `--device_preprocess true`).  Bit-identical observations either way
    (tests/test_atari_adapter.py).

ALE itself is third-party and not shipped: pass any object with ALE's Python interface as `ale=`, or have
`ale_python_interface` / `ale_py` importable.
"""
import random

import numpy as np

from .environment import BaseEnvironment, max_resize_84

HISTORY = 4              # frames per observation
REPEAT = 4               # emulator frames per agent step
KEPT = 2                 # screens of a step that reach the observation
MAX_START_WAIT = 30

# (setter, key, value) applied before the ROM is loaded: explicit action repeat means no sticky actions, no ALE-side
# frame skip and no colour averaging (atari_emulator.py:19-24)
_ALE_OPTIONS = (("setFloat", b"repeat_action_probability", 0.0), ("setInt", b"frame_skip", 1),
                ("setBool", b"color_averaging", False))


def _open_ale():
    for module in ("ale_python_interface", "ale_py"):
        try:
            return __import__(module, fromlist=["ALEInterface"]).ALEInterface()
        except ImportError:
            continue
    raise ImportError("no Arcade Learning Environment binding found (ale_python_interface / ale_py); pass an "
                      "ALE-compatible object as AtariEmulator(..., ale=...)")


class AtariEmulator(BaseEnvironment):
    def __init__(self, actor_id, args, ale=None):
        self.ale = _open_ale() if ale is None else ale
        self.ale.setInt(b"random_seed", args.random_seed * (actor_id + 1))        # per-actor seed, :18
        for setter, key, value in _ALE_OPTIONS:
            getattr(self.ale, setter)(key, value)
        self.ale.loadROM(("%s/%s.bin" % (args.rom_path, args.game)).encode())
        self.legal_actions = self.ale.getMinimalActionSet()
        self.screen_width, self.screen_height = self.ale.getScreenDims()
        self.lives = self.ale.lives()
        self.random_start = args.random_start
        self.single_life_episodes = args.single_life_episodes
        self.call_on_new_frame = args.visualize

        h, w = self.screen_height, self.screen_width
        self._gray = np.zeros((h, w, 1), dtype=np.uint8)
        self._rgb = np.zeros((h, w, 3), dtype=np.uint8)
        self._kept = np.zeros((KEPT, h, w), dtype=np.uint8)        # the screens of the last step, oldest first
        self._history = np.zeros((84, 84, HISTORY), dtype=np.uint8)  # channel 0 oldest .. channel 3 newest

    # -- BaseEnvironment surface ----------------------------------------------------------------------
    def get_legal_actions(self):
        return self.legal_actions

    def get_noop(self):
        return [1.0, 0.0]

    def on_new_frame(self, frame):
        pass

    def get_initial_state(self):
        for pair in self.initial_raw():
            self._push(pair)
        return self._history.copy()

    def next(self, action):
        pair, reward, terminal = self.next_raw(action)
        self._push(pair)
        return self._history.copy(), reward, terminal

    # -- raw screens out (device preprocessing) --------------------------------------------------------
    def initial_raw(self):
        """New game; the HISTORY screen pairs that make up the initial observation, u8 [4,2,H,W]."""
        self.ale.reset_game()
        self.lives = self.ale.lives()
        if self.random_start:
            for _ in range(random.randint(0, MAX_START_WAIT)):
                self.ale.act(self.legal_actions[0])
        pairs = np.empty((HISTORY,) + self._kept.shape, dtype=np.uint8)
        for k in range(HISTORY):
            self._repeat(0)
            pairs[k] = self._kept
        if self._terminal():
            raise Exception('This should never happen.')
        return pairs

    def next_raw(self, action):
        """-> (screen pair u8 [2,H,W], summed reward, terminal)."""
        reward = self._repeat(int(np.argmax(action)))
        terminal = self._terminal()
        self.lives = self.ale.lives()
        return self._kept.copy(), reward, terminal

    # -- internals ---------------------------------------------------------------------------------------
    def _screen(self):
        self.ale.getScreenGrayscale(self._gray)
        if self.call_on_new_frame:
            self.ale.getScreenRGB(self._rgb)
            self.on_new_frame(self._rgb)
        return self._gray[..., 0]

    def _repeat(self, a):
        total = 0
        for i in range(REPEAT):
            total += self.ale.act(self.legal_actions[a])
            if i >= REPEAT - KEPT:
                self._kept[i - (REPEAT - KEPT)] = self._screen()
        return total

    def _terminal(self):
        over = self.ale.game_over()
        if self.single_life_episodes:
            return over or (self.lives > self.ale.lives())
        return over

    def _push(self, pair):
        self._history[..., :-1] = self._history[..., 1:]
        self._history[..., -1] = max_resize_84(pair)
