"""Command line and creator wiring of the trainer (the role of reference train.py:14-109).

The flag NAMES, short forms, destinations and defaults are the reference's (train.py:79-98) -- they are the contract
a run script depends on -- and so are `get_network_and_environment_creator(args)`, `get_arg_parser()`, `bool_arg` and
`main(args)`.  The flags are kept as a table below; `BUILD_FLAGS` are this build's additions.

  python -m paac_amd.train -g breakout --arch NATURE -df logs/
"""
import argparse
import functools
import logging
import os
import signal
import sys

from . import environment_creator, logger_utils, parallel
from .paac import PAACLearner
from .policy_v_network import NaturePolicyVNetwork, NIPSPolicyVNetwork


def bool_arg(string):
    """'true' / 'false' (any case) -> bool; anything else is an argparse error."""
    try:
        return {"true": True, "false": False}[string.lower()]
    except KeyError:
        raise argparse.ArgumentTypeError("Expected True or False, but got {}".format(string))


# (option strings, dest, default, type, help)
REFERENCE_FLAGS = (
    (("-g",), "game", "pong", None, "game to play"),
    (("-d", "--device"), "device", "/gpu:0", str, "'/gpu:N' selects the MI355X; '/cpu:0' is rejected (no CPU path)"),
    (("--rom_path",), "rom_path", "./atari_roms", None, "directory with the game ROMs (ALE environments only)"),
    (("-v", "--visualize"), "visualize", False, bool_arg, "call on_new_frame with every emulator screen"),
    (("--e",), "e", 0.1, float, "RMSProp epsilon"),
    (("--alpha",), "alpha", 0.99, float, "RMSProp decay of the squared-gradient average"),
    (("-lr", "--initial_lr"), "initial_lr", 0.0224, float, "learning rate at step 0"),
    (("-lra", "--lr_annealing_steps"), "lr_annealing_steps", 80000000, int,
     "global steps over which the learning rate falls linearly to zero"),
    (("--entropy",), "entropy_regularisation_strength", 0.02, float, "weight of the policy-entropy bonus"),
    (("--clip_norm",), "clip_norm", 3.0, float, "gradient norm the update is clipped to"),
    (("--clip_norm_type",), "clip_norm_type", "global", None, "'global' (joint norm), 'ignore' (no clipping); "
                                                             "'local' is undefined upstream and rejected"),
    (("--gamma",), "gamma", 0.99, float, "discount factor"),
    (("--max_global_steps",), "max_global_steps", 80000000, int, "environment steps to train for"),
    (("--max_local_steps",), "max_local_steps", 5, int, "t_max: steps per environment between updates"),
    (("--arch",), "arch", "NIPS", None, "'NIPS' or 'NATURE' (anything that is not NIPS selects NATURE)"),
    (("--single_life_episodes",), "single_life_episodes", False, bool_arg, "end an episode when a life is lost"),
    (("-ec", "--emulator_counts"), "emulator_counts", 32, int, "environments per learner (per GPU)"),
    (("-ew", "--emulator_workers"), "emulator_workers", 8, int, "host processes stepping host environments"),
    (("-df", "--debugging_folder"), "debugging_folder", "logs/", str, "checkpoints, args.json, metrics.jsonl"),
    (("-rs", "--random_start"), "random_start", True, bool_arg, "up to 30 no-op frames after every reset"),
)
BUILD_FLAGS = (
    (("--sampler",), "sampler", "philox", None,
     "device loop action sampler: 'numpy' = the reference's np.random.multinomial stream bit for bit, 'philox' = "
     "counter-based"),
    (("--sampler_seed",), "sampler_seed", 42, int, "seed of the counter-based sampler"),
    (("--host_environments",), "host_environments", False, bool_arg,
     "step BaseEnvironment plugins on the host even when a device twin exists"),
    (("--synthetic_terminal_p",), "synthetic_terminal_p", 0.01, float, "per-step terminal probability (synthetic)"),
    (("--synthetic_raw_frames",), "synthetic_raw_frames", False, bool_arg,
     "synthetic environments emit two raw 210x160 screens per step (GPU max + resize + history)"),
    (("--emulator",), "emulator", "synthetic", None,
     "'synthetic' (paac_amd/synthetic.py) or 'ale' (Atari through an installed Arcade Learning Environment)"),
    (("--device_preprocess",), "device_preprocess", False, bool_arg,
     "host environments hand out raw screen pairs; max + resize + frame history run on the GPU"),
    (("--user_arch",), "user_arch", "", None,
     "a user architecture instead of --arch (compiled on first use): filter counts of the 2 or 3 conv layers of the "
     "reference trunks' shapes (8x8/4, 4x4/2[, 3x3/1]) and the fc width, e.g. 32,64,64,1024 -- or filters:size:stride per "
     "layer, e.g. 32:8:4,64:5:2,64:3:1,512"),
    (("--checkpoint_format",), "checkpoint_format", "npz", None,
     "container of the checkpoints written: 'npz', or 'tf' = the reference's TensorFlow V2 tensor bundle "
     "(.index + .data-00000-of-00001); both are read"),
)


def get_arg_parser():
    parser = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for options, dest, default, kind, text in REFERENCE_FLAGS + BUILD_FLAGS:
        kwargs = dict(dest=dest, default=default, help=text)
        if kind is not None:
            kwargs["type"] = kind
        if dest in ("sampler", "emulator", "checkpoint_format"):
            kwargs["choices"] = {"sampler": ["philox", "numpy"], "emulator": ["synthetic", "ale"],
                                 "checkpoint_format": ["npz", "tf"]}[dest]
        parser.add_argument(*options, **kwargs)
    return parser


def get_network_and_environment_creator(args, random_seed=3):
    """-> (network_creator(name='local_learning'), environment creator); fills args.num_actions / args.random_seed
    the way train.py:52-56 does."""
    env_creator = environment_creator.EnvironmentCreator(args)
    args.num_actions = env_creator.num_actions
    args.random_seed = random_seed
    conf = dict(num_actions=args.num_actions, device=args.device, clip_norm=args.clip_norm,
                clip_norm_type=args.clip_norm_type,
                entropy_regularisation_strength=args.entropy_regularisation_strength)
    network_class = NIPSPolicyVNetwork if args.arch == 'NIPS' else NaturePolicyVNetwork
    if getattr(args, "user_arch", ""):
        # networks.py:117-120 / README.md:80-83: a new trunk mixed into PolicyVNetwork
        from .networks import define_architecture
        from .policy_v_network import PolicyVNetwork
        from .build import parse_user_arch
        trunk = define_architecture("USER", *parse_user_arch(args.user_arch))
        network_class = type("UserPolicyVNetwork", (PolicyVNetwork, trunk), {})

    def network_creator(name='local_learning'):
        return network_class(dict(conf, name=name))

    return network_creator, env_creator


def _stop(learner, owner_pid, signum, frame):
    """First signal: ask the training loop to stop at the next cycle boundary (it then runs cleanup() itself, with
    nothing in flight on the GPU and, data parallel, every rank leaving at the same cycle).  Second signal: clean up
    right here, as upstream does (train.py:38-49)."""
    if os.getpid() != owner_pid:          # forked emulator workers inherit the handler: only the trainer reacts
        return
    if not learner.stop_requested:
        logging.info('Signal %s detected, stopping at the next cycle boundary.', signum)
        learner.stop_requested = True
        return
    if parallel.world_size() > 1:
        # data parallel: cleanup() is collective (the checkpoint barrier, the pending optimizer step's graph) and only THIS
        # rank was signalled twice -- running it here would hang this rank in the barrier and leave its peers one
        # collective out of step.  The first signal's agreed stop is the clean way out; a second one just leaves.
        logging.info('Signal %s detected again on a data-parallel rank: exiting without the collective cleanup.', signum)
        os._exit(1)
    logging.info('Signal %s detected again, cleaning up now.', signum)
    learner.cleanup()
    logging.info('Cleanup completed, shutting down...')
    sys.exit(0)


def setup_kill_signal_handler(learner):
    handler = functools.partial(_stop, learner, os.getpid())
    for signum in (signal.SIGTERM, signal.SIGINT):
        signal.signal(signum, handler)


def main(args):
    """train.py:24-35.  Under `python -m torch.distributed.run --nproc-per-node G -m paac_amd.train ...` every process
    is one data-parallel rank: its GPU is '/gpu:<LOCAL_RANK>', `-ec` environments PER GPU (global_step advances by
    G * ec per step), gradients are summed over RCCL once per update, rank 0 writes checkpoints / args / metrics."""
    world = parallel.init_from_env(args)          # before anything touches a GPU
    logging.debug('Configuration: %s', args)
    network_creator, env_creator = get_network_and_environment_creator(args)
    learner = PAACLearner(network_creator, env_creator, args)
    setup_kill_signal_handler(learner)
    logging.info('Starting training (%d data-parallel rank%s)', world, '' if world == 1 else 's')
    try:
        learner.train()
    except parallel.ReplicaMismatch as exc:
        # every rank raises at the same cycle (the comparison is itself a collective): no checkpoint of diverged weights,
        # no collective cleanup, a non-zero exit code for the launcher
        logging.error('%s -- stopping', exc)
        sys.exit(3)
    logging.info('Finished training')
    parallel.shutdown()


save_args = logger_utils.save_args      # logger_utils.py:15-20


if __name__ == '__main__':
    logging.basicConfig(stream=sys.stdout, level=logging.DEBUG)
    cli_args = get_arg_parser().parse_args()
    if int(os.environ.get("RANK", "0")) == 0:
        save_args(cli_args, cli_args.debugging_folder)
    main(cli_args)
