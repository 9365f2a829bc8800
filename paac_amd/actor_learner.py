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
from .session import Saver, Session, checkpoint_key, tensor_of_key

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
        # the learner owns every write to the parameters: the optimizer step re-packs the conv weights for the fused conv
        # launch itself; host-side writes (set_parameters, restore, broadcast) go through weights_changed
        self.ctx.set_managed_weights(True)
        self.network.weights_changed = lambda: self.ctx.pack_weights(self.network.params)
        self.ctx.pack_weights(self.network.params)

        self.network_saver = self.network.make_saver()
        self.optimizer_saver = Saver(self._get_optimizer_arrays, self._set_optimizer_arrays, max_to_keep=1)
        fmt = getattr(args, "checkpoint_format", None)
        if fmt:
            self.network_saver.fmt = self.optimizer_saver.fmt = fmt
        if self.network_saver.fmt == "tf":
            # the reference's network saver is tf.train.Saver() over ALL variables, optimizer slots included
            # (actor_learner.py:79): a bundle its restore accepts holds them too
            variables = self.network_saver.get_arrays
            self.network_saver.get_arrays = lambda: dict(variables(), **self._get_optimizer_arrays())

    # -- optimizer slots under the reference's names ('<var>/OptimizerVariables', '<var>/OptimizerVariables_1') ----
    def _get_optimizer_arrays(self):
        scope = self.network.name
        out = {}
        for slot, flat in (("OptimizerVariables", self.rms), ("OptimizerVariables_1", self.mom)):
            for k, v in self.network.get_parameters(flat).items():
                out[checkpoint_key(scope, k, slot)] = v
        return out

    def _set_optimizer_arrays(self, d):
        lay = self.network.layout
        host = {"OptimizerVariables": np.ones(lay["total"], dtype=np.float32),
                "OptimizerVariables_1": np.zeros(lay["total"], dtype=np.float32)}
        where = {t["name"]: t for t in lay["tensors"]}
        seen = set()
        for key, value in d.items():
            name, slot = tensor_of_key(key)
            if slot is None:         # a bundle of all variables (the reference's network saver): the slots are taken from it
                continue
            t = where[name]
            host[slot][t["offset"]:t["offset"] + t["size"]] = np.asarray(value, dtype=np.float32).reshape(-1)
            seen.add((name, slot))
        if len(seen) != 2 * len(where):
            raise KeyError("optimizer checkpoint holds %d of %d slot tensors" % (len(seen), 2 * len(where)))
        self.rms.copy_(torch.from_numpy(host["OptimizerVariables"]))
        self.mom.copy_(torch.from_numpy(host["OptimizerVariables_1"]))

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

    # -- reference methods (actor_learner.py:89-127): same names and behaviour ------------------------------
    def save_vars(self, force=False):
        """Network + optimizer checkpoints once CHECKPOINT_INTERVAL global steps have passed since the last one (or
        when forced).  Data parallel: the replicas are identical, rank 0 writes, every rank keeps the same cadence."""
        due = force or (self.global_step - self.last_saving_step) >= CHECKPOINT_INTERVAL
        if not due:
            return
        self.last_saving_step = self.global_step
        self._sync_device()                    # nothing in flight: weights, rms and mom belong to the same update
        if parallel.rank() == 0:
            for saver, folder in ((self.network_saver, self.network_checkpoint_folder),
                                  (self.optimizer_saver, self.optimizer_checkpoint_folder)):
                saver.save(self.session, folder, global_step=self.last_saving_step)
        parallel.barrier()

    def _sync_device(self):
        torch.cuda.synchronize(self.torch_device)

    def rescale_reward(self, reward):
        """Immediate reward clipped to [-1, 1] (actor_learner.py:95-101)."""
        return min(1.0, max(-1.0, reward))

    def init_network(self):
        """Restore the latest network / optimizer checkpoints if present, else initialise (actor_learner.py:103-117,
        networks.py:122-135); returns the global step to resume from.  Data parallel: every rank restores (or
        initialises), then rank 0's weights and optimizer slots are broadcast so the replicas start identical."""
        for folder in (self.network_checkpoint_folder, self.optimizer_checkpoint_folder):
            os.makedirs(folder, exist_ok=True)
        resumed_step = self.network.init(self.network_checkpoint_folder, self.network_saver, self.session)
        optimizer_checkpoint = Saver.latest_checkpoint(self.optimizer_checkpoint_folder)
        if optimizer_checkpoint is not None:
            logging.info('Restoring optimizer variables from previous run')
            self.optimizer_saver.restore(self.session, optimizer_checkpoint)
            # save_vars writes the network first and the optimizer second, and a torn newest file is skipped on resume:
            # the two can come from different updates (upstream has the same window, silently)
            optimizer_step = Saver.step_of(optimizer_checkpoint)
            if optimizer_step is not None and optimizer_step != int(resumed_step):
                logging.warning('Optimizer checkpoint is from step %d, network checkpoint from step %d: resuming with '
                                'weights and RMSProp statistics of different updates', optimizer_step, int(resumed_step))
        if parallel.world_size() > 1:
            step = torch.tensor([int(resumed_step)], dtype=torch.int64, device=self.torch_device)
            for t in (self.network.params, self.rms, self.mom, step):
                parallel.broadcast_(t, src=0)
            self.network.weights_changed()
            resumed_step = int(step.item())
        return resumed_step          # like upstream, last_saving_step stays 0: a resumed run checkpoints on its first cycle

    def get_lr(self):
        """Linear anneal to zero over lr_annealing_steps (actor_learner.py:119-123)."""
        if self.global_step > self.lr_annealing_steps:
            return 0.0
        return self.initial_lr - (self.global_step * self.initial_lr / self.lr_annealing_steps)

    def cleanup(self):
        self.save_vars(True)
        self.session.close()
