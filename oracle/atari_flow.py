"""Oracle: the reference's Atari episode flow (TEST INFRASTRUCTURE, see __init__).

Follows, step by step, on the oracle's FramePool / ObservationPool / max_resize:
  atari_emulator.py:17-29   per-actor seed = random_seed * (actor_id + 1); minimal action set; lives
  atari_emulator.py:60-67   new game: reset, then -- with random_start -- random.randint(0, 30) no-op emulator frames
  atari_emulator.py:77-86   action repeat: 4 emulator frames, reward summed over all 4, only the last 2 screens pooled
  atari_emulator.py:88-96   initial state: new game + 4 no-op action repeats, each pushing one processed frame
  atari_emulator.py:98-106  next: repeat, push, terminal, THEN lives refreshed
  atari_emulator.py:108-112 terminal = game over, or a lost life with single_life_episodes
  emulator_runner.py:24-31  runner step: on terminal the shared observation becomes a fresh initial state while reward
                            and terminal still describe the ending transition
PINNED: tests/golden/atari_emulator_flow.npz holds the outputs of the reference's own AtariEmulator class driven by
the reference's own EmulatorRunner._run over tests/fake_ale.FakeALE (tests/golden/make_golden_atari.py);
tests/test_atari_adapter.py::test_oracle_flow_reproduces_the_reference_capture replays them through this class.
"""
import random

import numpy as np

from . import preprocess as opre


class ReferenceFlow(object):
    def __init__(self, actor_id, args, ale):
        self.ale = ale
        self.ale.setInt(b"random_seed", args.random_seed * (actor_id + 1))
        self.legal = self.ale.getMinimalActionSet()
        self.args = args
        self.lives = self.ale.lives()
        self.frames = opre.FramePoolOracle()
        self.obs = opre.ObservationPoolOracle()

    def _screen(self):
        g = np.zeros((opre.RAW_H, opre.RAW_W, 1), dtype=np.uint8)
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

    def runner_step(self, a):
        """emulator_runner.py:24-31 -> (shared observation, reward, terminal)."""
        o, r, t = self.next(a)
        return (self.initial() if t else o), r, t
