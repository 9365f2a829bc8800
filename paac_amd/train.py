"""Command line + creator wiring (mirrors reference train.py:14-109): identical flag names and defaults,
`get_network_and_environment_creator(args)`, SIGINT/SIGTERM -> learner.cleanup().

Extra flags (not in the reference) are namespaced `--synthetic_*`, `--sampler`, `--host_environments`.
"""
import argparse
import copy
import logging
import os
import signal
import sys

from . import environment_creator, logger_utils
from .paac import PAACLearner
from .policy_v_network import NaturePolicyVNetwork, NIPSPolicyVNetwork


def bool_arg(string):
    value = string.lower()
    if value == 'true':
        return True
    elif value == 'false':
        return False
    else:
        raise argparse.ArgumentTypeError("Expected True or False, but got {}".format(string))


def main(args):
    logging.debug('Configuration: {}'.format(args))
    network_creator, env_creator = get_network_and_environment_creator(args)
    learner = PAACLearner(network_creator, env_creator, args)
    setup_kill_signal_handler(learner)
    logging.info('Starting training')
    learner.train()
    logging.info('Finished training')


def setup_kill_signal_handler(learner):
    main_process_pid = os.getpid()

    def signal_handler(signal, frame):
        if os.getpid() == main_process_pid:
            logging.info('Signal ' + str(signal) + ' detected, cleaning up.')
            learner.cleanup()
            logging.info('Cleanup completed, shutting down...')
            sys.exit(0)

    signal.signal(signal.SIGTERM, signal_handler)
    signal.signal(signal.SIGINT, signal_handler)


def get_network_and_environment_creator(args, random_seed=3):
    env_creator = environment_creator.EnvironmentCreator(args)
    num_actions = env_creator.num_actions
    args.num_actions = num_actions
    args.random_seed = random_seed

    network_conf = {'num_actions': num_actions,
                    'entropy_regularisation_strength': args.entropy_regularisation_strength,
                    'device': args.device,
                    'clip_norm': args.clip_norm,
                    'clip_norm_type': args.clip_norm_type}
    if args.arch == 'NIPS':
        network = NIPSPolicyVNetwork
    else:
        network = NaturePolicyVNetwork

    def network_creator(name='local_learning'):
        copied_network_conf = copy.copy(network_conf)
        copied_network_conf['name'] = name
        return network(copied_network_conf)

    return network_creator, env_creator


def get_arg_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('-g', default='pong', help='Name of game', dest='game')
    parser.add_argument('-d', '--device', default='/gpu:0', type=str, help="Device to be used ('/gpu:0', '/gpu:1',...); '/cpu:0' is rejected: this build is MI355X-only", dest="device")
    parser.add_argument('--rom_path', default='./atari_roms', help='Directory where the game roms are located (needed for ALE environment)', dest="rom_path")
    parser.add_argument('-v', '--visualize', default=False, type=bool_arg, help="0: no visualization of emulator; 1: all emulators, for all actors, are visualized; 2: only 1 emulator (for one of the actors) is visualized", dest="visualize")
    parser.add_argument('--e', default=0.1, type=float, help="Epsilon for the Rmsprop and Adam optimizers", dest="e")
    parser.add_argument('--alpha', default=0.99, type=float, help="Discount factor for the history/coming gradient, for the Rmsprop optimizer", dest="alpha")
    parser.add_argument('-lr', '--initial_lr', default=0.0224, type=float, help="Initial value for the learning rate. Default = 0.0224", dest="initial_lr")
    parser.add_argument('-lra', '--lr_annealing_steps', default=80000000, type=int, help="Nr. of global steps during which the learning rate will be linearly annealed towards zero", dest="lr_annealing_steps")
    parser.add_argument('--entropy', default=0.02, type=float, help="Strength of the entropy regularization term (needed for actor-critic)", dest="entropy_regularisation_strength")
    parser.add_argument('--clip_norm', default=3.0, type=float, help="If clip_norm_type is local/global, grads will be clipped at the specified maximum (avaerage) L2-norm", dest="clip_norm")
    parser.add_argument('--clip_norm_type', default="global", help="Whether to clip grads by their norm or not. Values: ignore (no clipping), local (layer-wise norm), global (global norm)", dest="clip_norm_type")
    parser.add_argument('--gamma', default=0.99, type=float, help="Discount factor", dest="gamma")
    parser.add_argument('--max_global_steps', default=80000000, type=int, help="Max. number of training steps", dest="max_global_steps")
    parser.add_argument('--max_local_steps', default=5, type=int, help="Number of steps to gain experience from before every update.", dest="max_local_steps")
    parser.add_argument('--arch', default='NIPS', help="Which network architecture to use: from the NIPS or NATURE paper", dest="arch")
    parser.add_argument('--single_life_episodes', default=False, type=bool_arg, help="If True, training episodes will be terminated when a life is lost (for games)", dest="single_life_episodes")
    parser.add_argument('-ec', '--emulator_counts', default=32, type=int, help="The amount of emulators per agent. Default is 32.", dest="emulator_counts")
    parser.add_argument('-ew', '--emulator_workers', default=8, type=int, help="The amount of emulator workers per agent. Default is 8.", dest="emulator_workers")
    parser.add_argument('-df', '--debugging_folder', default='logs/', type=str, help="Folder where to save the debugging information.", dest="debugging_folder")
    parser.add_argument('-rs', '--random_start', default=True, type=bool_arg, help="Whether or not to start with 30 noops for each env. Default True", dest="random_start")
    # -- additions of this build --
    parser.add_argument('--sampler', default='philox', choices=['philox', 'numpy'], help="Action sampler of the device-resident loop: 'numpy' = the reference's np.random.multinomial stream bit for bit, 'philox' = counter-based", dest="sampler")
    parser.add_argument('--sampler_seed', default=42, type=int, dest="sampler_seed")
    parser.add_argument('--host_environments', default=False, type=bool_arg, help="Step BaseEnvironment plugins on the host even when a device twin exists", dest="host_environments")
    parser.add_argument('--synthetic_terminal_p', default=0.01, type=float, dest="synthetic_terminal_p")
    parser.add_argument('--synthetic_raw_frames', default=False, type=bool_arg, help="Synthetic envs emit two raw 210x160 frames per step (max + nearest resize + stack on the GPU)", dest="synthetic_raw_frames")
    parser.add_argument('--emulator', default='synthetic', choices=['synthetic', 'ale'], help="Environments: the synthetic family of paac_amd/synthetic.py, or Atari through an installed Arcade Learning Environment (paac_amd/atari_emulator.py)", dest="emulator")
    parser.add_argument('--device_preprocess', default=False, type=bool_arg, help="Host environments hand out raw 210x160 screen pairs; max + resize + frame history run on the GPU", dest="device_preprocess")
    return parser


save_args = logger_utils.save_args      # logger_utils.py:15-20


if __name__ == '__main__':
    logging.basicConfig(stream=sys.stdout, level=logging.DEBUG)
    args = get_arg_parser().parse_args()
    save_args(args, args.debugging_folder)
    logging.debug(args)
    main(args)
