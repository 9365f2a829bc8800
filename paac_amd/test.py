"""Evaluation harness (the role of reference test.py:22-88): restore the latest network checkpoint of a training
folder, play `test_count` environments with the sampled policy (PAACLearner.choose_next_actions) after up to `noops`
random no-op steps, print mean / min / max / std of the episode scores.

  python -m paac_amd.test -f logs/ -tc 30 -np 30

Flags as upstream (-f, -tc, -np, -gn, -gf, -d).  One deliberate difference: the reference's loop
(`while not all(episodes_over)`, test.py:77-83) overwrites every environment's flag each step, so with test_count > 1
it only stops when all environments happen to finish on the same step and keeps adding post-episode rewards; here an
environment's score is frozen when its first episode ends.
"""
import argparse
import os
import random
import time

import numpy as np

from . import hip_ops, logger_utils
from .paac import PAACLearner
from .session import Session
from .train import get_network_and_environment_creator

# (option strings, dest, default, type, required, help) -- names and defaults are upstream's (test.py:22-30)
FLAGS = (
    (("-f", "--folder"), "folder", None, str, True, "training folder: holds args.json and checkpoints/"),
    (("-tc", "--test_count"), "test_count", 1, int, False, "number of environments, one scored episode each"),
    (("-np", "--noops"), "noops", 30, int, False, "upper bound of the random number of no-op steps before play"),
    (("-gn", "--gif_name"), "gif_name", None, str, False, "record every screen of environment i into <name><i>.gif"),
    (("-gf", "--gif_folder"), "gif_folder", "", str, False, "directory the gifs are written to"),
    (("-d", "--device"), "device", "/gpu:0", str, False, "'/gpu:N': which MI355X evaluates the policy"),
)


def get_arg_parser():
    parser = argparse.ArgumentParser(description="Score the latest checkpoint of a training folder.")
    for options, dest, default, kind, required, text in FLAGS:
        parser.add_argument(*options, dest=dest, default=default, type=kind, required=required, help=text)
    return parser


class GifRecorder(object):
    """on_new_frame hook (environment.py:34-39): appends every screen an environment shows to one gif (30 fps)."""

    def __init__(self, path):
        try:
            import imageio
        except ImportError:
            raise ImportError("--gif_name needs the imageio package")
        self.writer = imageio.get_writer(path + '.gif', fps=30)

    def __call__(self, frame):
        self.writer.append_data(frame)


def get_save_frame(name):
    """Upstream's name for the hook factory (test.py:12-20)."""
    return GifRecorder(name)


def evaluate(network, env_creator, session, test_count, noops=30, max_steps=None, on_new_frame=None):
    """-> float32 [test_count]: score of the FIRST episode of each environment under the sampled policy."""
    environments = [env_creator.create_environment(i) for i in range(test_count)]
    if on_new_frame is not None:
        for i, environment in enumerate(environments):
            environment.on_new_frame = on_new_frame(i)
    states = np.stack([environment.get_initial_state() for environment in environments])
    for i, environment in enumerate(environments):          # random start: 0..noops no-op steps (test.py:69-73)
        for _ in range(random.randint(0, noops) if noops else 0):
            states[i] = environment.next(environment.get_noop())[0]
    scores = np.zeros(test_count, dtype=np.float32)
    playing = list(range(test_count))
    steps = 0
    while playing and (max_steps is None or steps < max_steps):
        actions, _, _ = PAACLearner.choose_next_actions(network, env_creator.num_actions, states, session)
        still = []
        for j in playing:
            states[j], reward, over = environments[j].next(actions[j])
            scores[j] += reward
            if not over:
                still.append(j)
        playing = still
        steps += 1
    return scores


def restore_settings(cli):
    """The training run's args.json with the evaluation overrides of test.py:32-48 applied on top."""
    settings = argparse.Namespace(**vars(cli))
    for k, v in logger_utils.load_args(os.path.join(cli.folder, 'args.json')).items():
        setattr(settings, k, v)
    overrides = dict(device=cli.device, debugging_folder='/tmp/logs', max_global_steps=0, random_start=False,
                     single_life_episodes=False, actor_id=0)
    if cli.gif_name:
        overrides["visualize"] = 1
    for k, v in overrides.items():
        setattr(settings, k, v)
    return settings


def main(argv=None):
    cli = get_arg_parser().parse_args(argv)
    args = restore_settings(cli)
    seed = int(np.random.RandomState(int(time.time())).randint(1000))
    network_creator, env_creator = get_network_and_environment_creator(args, random_seed=seed)
    network = network_creator()
    ctx = hip_ops.Context(network.arch_id, env_creator.num_actions, max_batch=max(1, args.test_count),
                          device_index=network.torch_device.index or 0)
    session = Session(network, ctx)
    network.init(os.path.join(cli.folder, 'checkpoints'), network.make_saver(), session)
    hook = None
    if args.gif_name:
        hook = lambda i: get_save_frame(os.path.join(args.gif_folder, args.gif_name + str(i)))
    try:
        rewards = evaluate(network, env_creator, session, args.test_count, noops=args.noops, on_new_frame=hook)
    finally:
        session.close()
        ctx.close()
    print('Performed {} tests for {}.'.format(args.test_count, args.game))
    for label, stat in (('Mean', np.mean), ('Min', np.min), ('Max', np.max), ('Std', np.std)):
        print('{0}: {1:.2f}'.format(label, stat(rewards)))
    return rewards


if __name__ == '__main__':
    main()
