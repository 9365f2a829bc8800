"""Environment factory (mirrors reference environment_creator.py:1-15).

The reference probes an ALE ROM for `num_actions` and builds AtariEmulator instances.  ALE is a third-party
emulator that is out of this build's scope, so the factory creates the synthetic
environments of paac_amd/synthetic.py; `num_actions` per game is ALE's minimal action-set size (pinned for
breakout/qbert/seaquest by the reference's pretrained/*/checkpoints/*.index actor_output_biases shapes).
A user environment plugs in exactly as in the reference: subclass BaseEnvironment and return it from
create_environment(i).
"""
from .synthetic import SyntheticEnvironment, terminal_threshold

# ALE minimal action set sizes.
GAME_NUM_ACTIONS = {
    "pong": 6, "breakout": 4, "qbert": 6, "seaquest": 18, "space_invaders": 6, "beam_rider": 9,
    "boxing": 18, "ms_pacman": 9, "name_this_game": 6,
}


class EnvironmentCreator(object):
    def __init__(self, args):
        game = getattr(args, "game", "pong")
        self.args = args
        if getattr(args, "emulator", "synthetic") == "ale":
            # environment_creator.py:8-14: the ROM decides the action count; one AtariEmulator per actor.  No device
            # twin: these environments are stepped on the host (screens -> GPU with --device_preprocess true).
            from . import atari_emulator
            probe = atari_emulator._open_ale()
            probe.loadROM(("%s/%s.bin" % (args.rom_path, game)).encode())
            self.num_actions = len(probe.getMinimalActionSet())
            self.create_environment = lambda i: atari_emulator.AtariEmulator(i, args)
            self._device_twin = False
            return
        self._device_twin = True
        self.num_actions = int(getattr(args, "num_actions_override", 0) or GAME_NUM_ACTIONS.get(game, 6))
        # args.random_seed is set by train.get_network_and_environment_creator AFTER this constructor runs
        # (train.py:52-56), so it is read when an environment is created, like atari_emulator.py:18 does.
        self.create_environment = lambda i: SyntheticEnvironment(
            i, self.num_actions, seed=self._seed(), terminal_p=self._terminal_p(), raw_frames=self._raw())

    def _seed(self):
        return int(getattr(self.args, "random_seed", 3))

    def _terminal_p(self):
        return float(getattr(self.args, "synthetic_terminal_p", 0.01))

    def _raw(self):
        return bool(getattr(self.args, "synthetic_raw_frames", False))

    @property
    def device_env_spec(self):
        """Device-batched twin of the same environments (PAACLearner uses it when present)."""
        if not self._device_twin:
            return None
        return dict(kind="synthetic", seed=self._seed(), terminal_threshold=terminal_threshold(self._terminal_p()),
                    raw_frames=self._raw())
