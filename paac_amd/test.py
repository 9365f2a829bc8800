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
from .session import Saver, Session
from .train import get_network_and_environment_creator


def get_save_frame(name):
    try:
        import imageio
    except ImportError:
        raise ImportError("--gif_name needs the imageio package")
    writer = imageio.get_writer(name + '.gif', fps=30)

    def get_frame(frame):
        writer.append_data(frame)

    return get_frame


def get_arg_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('-f', '--folder', type=str, help="Folder where to save the debugging information.", dest="folder", required=True)
    parser.add_argument('-tc', '--test_count', default='1', type=int, help="The amount of tests to run on the given network", dest="test_count")
    parser.add_argument('-np', '--noops', default=30, type=int, help="Maximum amount of no-ops to use", dest="noops")
    parser.add_argument('-gn', '--gif_name', default=None, type=str, help="If provided, a gif will be produced and stored with this name", dest="gif_name")
    parser.add_argument('-gf', '--gif_folder', default='', type=str, help="The folder where to save gifs.", dest="gif_folder")
    parser.add_argument('-d', '--device', default='/gpu:0', type=str, help="Device to be used ('/gpu:0', '/gpu:1',...)", dest="device")
    return parser


def evaluate(network, env_creator, session, test_count, noops=30, max_steps=None, on_new_frame=None):
    """-> float32 [test_count] scores of the first episode of each environment."""
    environments = [env_creator.create_environment(i) for i in range(test_count)]
    if on_new_frame is not None:
        for i, environment in enumerate(environments):
            environment.on_new_frame = on_new_frame(i)
    states = np.asarray([environment.get_initial_state() for environment in environments])
    if noops != 0:
        for i, environment in enumerate(environments):
            for _ in range(random.randint(0, noops)):
                state, _, _ = environment.next(environment.get_noop())
                states[i] = state
    episodes_over = np.zeros(test_count, dtype=bool)
    rewards = np.zeros(test_count, dtype=np.float32)
    steps = 0
    while not episodes_over.all() and (max_steps is None or steps < max_steps):
        actions, _, _ = PAACLearner.choose_next_actions(network, env_creator.num_actions, states, session)
        for j, environment in enumerate(environments):
            if episodes_over[j]:
                continue
            state, r, episode_over = environment.next(actions[j])
            states[j] = state
            rewards[j] += r
            episodes_over[j] = episode_over
        steps += 1
    return rewards


def main(argv=None):
    args = get_arg_parser().parse_args(argv)
    arg_file = os.path.join(args.folder, 'args.json')
    device = args.device
    for k, v in logger_utils.load_args(arg_file).items():
        setattr(args, k, v)
    args.max_global_steps = 0
    df = args.folder
    args.debugging_folder = '/tmp/logs'
    args.device = device
    args.random_start = False
    args.single_life_episodes = False
    if args.gif_name:
        args.visualize = 1
    args.actor_id = 0
    rng = np.random.RandomState(int(time.time()))
    seed = int(rng.randint(1000))

    network_creator, env_creator = get_network_and_environment_creator(args, random_seed=seed)
    network = network_creator()
    scope = network.name
    saver = Saver(lambda: {"%s/%s" % (scope, k): v for k, v in network.get_parameters().items()},
                  lambda d: network.set_parameters({k.split("/", 1)[1]: v for k, v in d.items()}))
    ctx = hip_ops.Context(network.arch_id, env_creator.num_actions, max_batch=max(1, args.test_count),
                          device_index=network.torch_device.index or 0)
    session = Session(network, ctx)
    network.init(os.path.join(df, 'checkpoints'), saver, session)
    hook = None
    if args.gif_name:
        hook = lambda i: get_save_frame(os.path.join(args.gif_folder, args.gif_name + str(i)))
    rewards = evaluate(network, env_creator, session, args.test_count, noops=args.noops, on_new_frame=hook)
    session.close()
    ctx.close()
    print('Performed {} tests for {}.'.format(args.test_count, args.game))
    print('Mean: {0:.2f}'.format(np.mean(rewards)))
    print('Min: {0:.2f}'.format(np.min(rewards)))
    print('Max: {0:.2f}'.format(np.max(rewards)))
    print('Std: {0:.2f}'.format(np.std(rewards)))
    return rewards


if __name__ == '__main__':
    main()
