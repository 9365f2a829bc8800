"""Base learner (mirrors reference actor_learner.py:11-127).

Holds what the reference's ActorLearner holds -- hyper-parameters, the environments, the network, the
RMSProp optimizer state, savers, lr schedule, reward clipping, checkpoint cadence -- with the TensorFlow
optimizer graph (compute_gradients / clip_by_global_norm / apply_gradients, :31-70) replaced by
paac_loss_backward + paac_clip_rmsprop on a flat parameter buffer.
"""
import logging
import os

import numpy as np
import torch

from . import _lib, hip_ops, parallel
from .networks import Placeholder
from .session import Saver, Session

CHECKPOINT_INTERVAL = 1000000      # actor_learner.py:8


class ActorLearner(object):

    def __init__(self, network_creator, environment_creator, args):
        self.global_step = 0
        self.max_local_steps = args.max_local_steps
        self.num_actions = args.num_actions
        self.initial_lr = args.initial_lr
        self.lr_annealing_steps = args.lr_annealing_steps
        self.emulator_counts = args.emulator_counts
        self.device = args.device
        self.debugging_folder = args.debugging_folder
        self.network_checkpoint_folder = os.path.join(self.debugging_folder, 'checkpoints/')
        self.optimizer_checkpoint_folder = os.path.join(self.debugging_folder, 'optimizer_checkpoints/')
        self.last_saving_step = 0

        # RMSPropOptimizer(lr, decay=alpha, epsilon=e): momentum 0.0, rms slot init 1.0 (actor_learner.py:31-34)
        self.learning_rate = Placeholder('learning_rate')
        self.alpha = args.alpha
        self.e = args.e
        self.momentum = 0.0
        self.clip_norm = args.clip_norm
        self.clip_norm_type = args.clip_norm_type
        if self.clip_norm_type == 'ignore':
            self.clip_mode = _lib.CLIP_IGNORE
        elif self.clip_norm_type == 'global':
            self.clip_mode = _lib.CLIP_GLOBAL
        elif self.clip_norm_type == 'local':
            # actor_learner.py:62-63 iterates over (grad, var) tuples and hands the tuple to tf.clip_by_norm:
            # the branch cannot run upstream, so there is no behaviour to reproduce.
            raise Exception("clip_norm_type 'local' is undefined in the reference (actor_learner.py:62-63)")
        else:
            raise Exception('Norm type not recognized')

        self.environment_creator = environment_creator
        self.emulators = np.asarray([environment_creator.create_environment(i)
                                     for i in range(self.emulator_counts)])
        self.max_global_steps = args.max_global_steps
        self.gamma = args.gamma
        self.game = args.game
        self.network = network_creator()
        self.entropy_beta = float(self.network.entropy_regularisation_strength)

        dev = self.network.torch_device
        torch.cuda.set_device(dev)
        self.torch_device = dev
        n = self.network.layout["total"]
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.rms = torch.ones(n, dtype=torch.float32, device=dev)         # .meta: OptimizerVariables init 1.0
        self.mom = torch.zeros(n, dtype=torch.float32, device=dev)        # .meta: OptimizerVariables_1 zeros
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self.gnorm_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self.loss_dev = torch.zeros(4, dtype=torch.float32, device=dev)
        self.train_step = Placeholder('train_step')

        self.ctx = hip_ops.Context(self.network.arch_id, self.num_actions,
                                   max_batch=self.emulator_counts * (self.max_local_steps + 1),   # + bootstrap rows
                                   device_index=dev.index or 0)
        self.session = Session(self.network, self.ctx, learner=self)

        scope = self.network.name
        self.network_saver = Saver(lambda: {"%s/%s" % (scope, k): v for k, v in self.network.get_parameters().items()},
                                   lambda d: self.network.set_parameters({k.split("/", 1)[1]: v for k, v in d.items()}))
        self.optimizer_saver = Saver(self._get_optimizer_arrays, self._set_optimizer_arrays, max_to_keep=1)

    # -- optimizer slots in the reference's naming ---------------------------------------------------
    def _get_optimizer_arrays(self):
        scope = self.network.name
        out = {}
        for k, v in self.network.get_parameters(self.rms).items():
            out["%s/%s/OptimizerVariables" % (scope, k)] = v
        for k, v in self.network.get_parameters(self.mom).items():
            out["%s/%s/OptimizerVariables_1" % (scope, k)] = v
        return out

    def _set_optimizer_arrays(self, d):
        lay = self.network.layout
        rms = np.ones(lay["total"], dtype=np.float32)
        mom = np.zeros(lay["total"], dtype=np.float32)
        for t in lay["tensors"]:
            for suffix, dst in (("OptimizerVariables", rms), ("OptimizerVariables_1", mom)):
                key = "%s/%s/%s" % (self.network.name, t["name"], suffix)
                dst[t["offset"]:t["offset"] + t["size"]] = np.asarray(d[key], dtype=np.float32).reshape(-1)
        self.rms.copy_(torch.from_numpy(rms))
        self.mom.copy_(torch.from_numpy(mom))

    # -- one optimizer step from a reference-style feed dict (Session.run([train_step, ...], feed)) ----
    def _train_step_from_feed(self, feed_dict):
        net = self.network
        dev = self.torch_device
        states = torch.from_numpy(np.ascontiguousarray(np.asarray(feed_dict[net.input_ph]).astype(np.uint8))).to(dev)
        onehot = np.asarray(feed_dict[net.selected_action_ph])
        actions = torch.from_numpy(np.argmax(onehot, axis=1).astype(np.int32)).to(dev)
        y = torch.from_numpy(np.asarray(feed_dict[net.critic_target_ph]).astype(np.float32)).to(dev)
        adv = torch.from_numpy(np.asarray(feed_dict[net.adv_actor_ph]).astype(np.float32)).to(dev)
        self.lr_dev.fill_(float(np.float32(feed_dict[self.learning_rate])))
        self.ctx.loss_backward(net.params, states, actions, y, adv, self.entropy_beta, self.grad, self.loss_dev)
        self._allreduce_grad()
        self.ctx.clip_rmsprop(net.params, self.grad, self.rms, self.mom, self.lr_dev, self.alpha, self.momentum,
                              self.e, self.clip_norm, self.clip_mode, self._grad_scale(), self.gnorm_dev)

    # -- data parallel: one sum all-reduce of the flat gradient per update (paac_amd/parallel.py) ------
    @staticmethod
    def _world():
        return parallel.world_size()

    def _grad_scale(self):
        return parallel.grad_scale()

    def _allreduce_grad(self):
        parallel.allreduce_sum_(self.grad)

    # -- reference methods ---------------------------------------------------------------------------
    def save_vars(self, force=False):
        if force or self.global_step - self.last_saving_step >= CHECKPOINT_INTERVAL:
            self.last_saving_step = self.global_step
            self.network_saver.save(self.session, self.network_checkpoint_folder, global_step=self.last_saving_step)
            self.optimizer_saver.save(self.session, self.optimizer_checkpoint_folder, global_step=self.last_saving_step)

    def rescale_reward(self, reward):
        """ Clip immediate reward """
        if reward > 1.0:
            reward = 1.0
        elif reward < -1.0:
            reward = -1.0
        return reward

    def init_network(self):
        if not os.path.exists(self.network_checkpoint_folder):
            os.makedirs(self.network_checkpoint_folder)
        if not os.path.exists(self.optimizer_checkpoint_folder):
            os.makedirs(self.optimizer_checkpoint_folder)
        last_saving_step = self.network.init(self.network_checkpoint_folder, self.network_saver, self.session)
        path = Saver.latest_checkpoint(self.optimizer_checkpoint_folder)
        if path is not None:
            logging.info('Restoring optimizer variables from previous run')
            self.optimizer_saver.restore(self.session, path)
        return last_saving_step

    def get_lr(self):
        if self.global_step <= self.lr_annealing_steps:
            return self.initial_lr - (self.global_step * self.initial_lr / self.lr_annealing_steps)
        else:
            return 0.0

    def cleanup(self):
        self.save_vars(True)
        self.session.close()
