"""A deterministic stand-in with ALE's Python interface, for the adapter tests (no emulator ships with this build).
Screens, rewards, lives and game-over are pure functions of (seed, frames since power-on), so two instances with the
same seed driven by the same actions stay identical."""
import numpy as np


class FakeALE(object):
    WIDTH, HEIGHT = 160, 210

    def __init__(self, episode_frames=150, life_every=61):
        self.seed = 0
        self.t = 0                  # emulator frames since power-on
        self.start = 0              # frame index at which the current game started
        self.episode_frames = episode_frames
        self.life_every = life_every
        self.options = {}

    # -- configuration -------------------------------------------------------------------------------
    def setInt(self, key, value):
        self.options[key] = value
        if key == b"random_seed":
            self.seed = int(value)

    def setFloat(self, key, value):
        self.options[key] = value

    def setBool(self, key, value):
        self.options[key] = value

    def loadROM(self, path):
        self.rom = path

    def getMinimalActionSet(self):
        return np.array([0, 1, 3, 4], dtype=np.int32)

    def getScreenDims(self):
        return self.WIDTH, self.HEIGHT

    # -- emulation ------------------------------------------------------------------------------------
    def reset_game(self):
        self.start = self.t

    def act(self, action):
        self.t += 1
        h = (self.seed * 1000003 + self.t * 7919 + int(action) * 104729) % 11
        return [0, 0, 0, 0, 1, 0, 0, 4, 0, -3, 0][h]

    def lives(self):
        return 5 - min(4, (self.t - self.start) // self.life_every)

    def game_over(self):
        return (self.t - self.start) >= self.episode_frames

    def _screen(self):
        rs = np.random.RandomState((self.seed * 7919 + self.t) % (2 ** 31))
        return rs.randint(0, 256, (self.HEIGHT, self.WIDTH), dtype=np.uint8)

    def getScreenGrayscale(self, out):
        out[..., 0] = self._screen()

    def getScreenRGB(self, out):
        out[...] = self._screen()[..., None]
