"""`Session`: the object passed where the reference passes a tf.Session (paac.py:18-29, test.py:77).

It understands exactly the fetch/feed patterns of the hot path and dispatches them to the HIP library:
  run([net.output_layer_v, net.output_layer_pi], {net.input_ph: states})          paac.py:20-23
  run(net.output_layer_v, {net.input_ph: states})                                  paac.py:140-142
  run([learner.train_step, ...], {input_ph, critic_target_ph, selected_action_ph,  paac.py:157-165
                                  adv_actor_ph, learner.learning_rate})
Host numpy in, host numpy out (the reference's contract); PAACLearner.train() itself uses the fused
device-resident path and never goes through here per step.
"""
import os

import numpy as np
import torch

from . import hip_ops, tf_bundle


def variable_scope(scope, tensor_name):
    """TensorFlow scope of a variable as the reference's graph names it: the nested tf.name_scope(self.name) blocks
    put the trunk under '<name>_1' (networks.py:111,144,160) and the heads under '<name>_2' (policy_v_network.py:16) --
    the keys of pretrained/*/checkpoints/*.index ('local_learning_1/conv1_weights', 'local_learning_2/actor_output_biases')."""
    return "%s_%d" % (scope, 2 if tensor_name.startswith(("actor_output", "critic_output")) else 1)


def checkpoint_key(scope, tensor_name, slot=None):
    """slot: None (the variable), 'OptimizerVariables' (RMSProp ms) or 'OptimizerVariables_1' (momentum)
    (actor_learner.py:31-34: RMSPropOptimizer(..., name='OptimizerVariables') names its slots after itself)."""
    key = "%s/%s" % (variable_scope(scope, tensor_name), tensor_name)
    return key if slot is None else "%s/%s" % (key, slot)


def tensor_of_key(key):
    """-> (tensor name, slot or None); also accepts the un-numbered '<scope>/<tensor>' keys of round-1 checkpoints."""
    parts = key.split("/")
    slot = parts[2] if len(parts) > 2 else None
    return parts[1], slot


def _step_of(path):
    return int(path[path.rindex('-') + 1:].split('.')[0])


class Saver(object):
    """Stand-in for tf.train.Saver (actor_learner.py:26-27,79-82) in two containers, both keyed by the reference's
    variable names (checkpoint_key) and named after the global step like the reference's '-<global step>' bundles
    (networks.py:134 parses the step from the name):
      * 'npz' (default): one '-<step>.npz' file;
      * 'tf': the reference's own container, a TensorFlow V2 tensor bundle '-<step>.index' + '-<step>.data-00000-of-00001'
        plus the `checkpoint` state file (paac_amd/tf_bundle.py) -- what the reference's Saver reads and writes.
    restore() and latest_checkpoint() take either, whatever the saver writes.  A save is atomic (temporary files + rename,
    the bundle's index last) and older checkpoints are pruned only afterwards; a truncated or unreadable checkpoint left by
    a killed run is skipped on resume.  PAAC_CHECKPOINT_FORMAT / --checkpoint_format choose the written container."""

    def __init__(self, get_arrays, set_arrays, max_to_keep=5, fmt=None):
        self.get_arrays, self.set_arrays, self.max_to_keep = get_arrays, set_arrays, max_to_keep
        self.fmt = fmt or os.environ.get("PAAC_CHECKPOINT_FORMAT", "npz")
        if self.fmt not in ("npz", "tf"):
            raise ValueError("checkpoint format %r: expected 'npz' or 'tf'" % (self.fmt,))

    @staticmethod
    def _readable(path):
        if path.endswith(".index"):
            return tf_bundle.readable(path[:-len(".index")])
        try:
            with np.load(path, allow_pickle=False) as z:
                return len(z.files) > 0
        except Exception:
            return False

    @staticmethod
    def checkpoints(folder):
        import glob
        found = glob.glob(os.path.join(folder, "-*.npz")) + glob.glob(os.path.join(folder, "-*.index"))
        # oldest first; two containers saved at the same step (the format was switched between runs): the one written last
        # is the newer checkpoint
        return sorted((p for p in found if Saver.step_of(p) is not None), key=lambda p: (_step_of(p), os.path.getmtime(p), p))

    @staticmethod
    def step_of(path):
        """Global step a checkpoint file was saved at (the suffix after the last '-', networks.py:134), None if unparsable."""
        try:
            return _step_of(path)
        except ValueError:
            return None

    @staticmethod
    def latest_checkpoint(folder):
        for path in reversed(Saver.checkpoints(folder)):
            if Saver._readable(path):
                return path
        return None

    @staticmethod
    def _remove(path):
        os.remove(path)
        if path.endswith(".index") and os.path.exists(path[:-len(".index")] + tf_bundle.DATA_SUFFIX):
            os.remove(path[:-len(".index")] + tf_bundle.DATA_SUFFIX)

    def save(self, session, folder, global_step):
        if self.fmt == "tf":
            name = "-%d" % int(global_step)
            path = tf_bundle.write(os.path.join(folder, name), self.get_arrays())
        else:
            path = os.path.join(folder, "-%d.npz" % int(global_step))
            tmp = os.path.join(folder, ".tmp-%d-%d.npz" % (int(global_step), os.getpid()))
            with open(tmp, "wb") as f:
                np.savez(f, **self.get_arrays())
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp, path)
        for p in self.checkpoints(folder)[:-self.max_to_keep]:
            if p != path:
                self._remove(p)
        if self.fmt == "tf":
            # every bundle still on disk, oldest first, the new one last: a tf.train.Saver resuming from this folder keeps
            # tracking (and pruning) the older ones
            kept = [os.path.basename(p)[:-len(".index")] for p in self.checkpoints(folder) if p.endswith(".index") and p != path]
            tf_bundle.write_state_file(folder, name, kept + [name])
        return path

    def restore(self, session, path):
        if path.endswith(".index"):
            self.set_arrays(tf_bundle.read(path[:-len(".index")]))
            return
        with np.load(path, allow_pickle=False) as z:
            self.set_arrays({k: z[k] for k in z.files})


class Session(object):
    def __init__(self, network, ctx, learner=None):
        self.network = network
        self.ctx = ctx
        self.learner = learner
        self.device = network.torch_device
        self.closed = False

    def _states(self, feed_dict):
        s = np.ascontiguousarray(np.asarray(feed_dict[self.network.input_ph]).astype(np.uint8))
        return torch.from_numpy(s).to(self.device)

    def run(self, fetches, feed_dict=None):
        if self.closed:
            raise RuntimeError("Session is closed")
        net = self.network
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        if self.learner is not None and self.learner.train_step in flist:
            self.learner._train_step_from_feed(feed_dict)
            out = [None for _ in flist]
            return out[0] if single else out
        states = self._states(feed_dict)
        B = states.shape[0]
        A = net.num_actions
        probs = torch.empty((B, A), dtype=torch.float32, device=self.device)
        values = torch.empty((B,), dtype=torch.float32, device=self.device)
        for i in range(0, B, self.ctx.max_batch):            # evaluation batches may exceed the training batch
            j = min(B, i + self.ctx.max_batch)
            self.ctx.forward(net.params, states[i:j], probs=probs[i:j], values=values[i:j])
        res = []
        for f in flist:
            if f is net.output_layer_v:
                res.append(values.cpu().numpy())
            elif f is net.output_layer_pi:
                res.append(probs.cpu().numpy())
            else:
                raise ValueError("Session.run: unsupported fetch %r" % (f,))
        return res[0] if single else res

    def close(self):
        self.closed = True
