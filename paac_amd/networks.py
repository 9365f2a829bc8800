"""Network plugin surface (mirrors reference networks.py:100-169).

In the reference a Network is a TensorFlow graph fragment; here it is a DESCRIPTION (architecture id, flat
parameter layout) plus the flat fp32 parameter buffer in HBM.  The arithmetic -- u8->f32 * (1/255)
(networks.py:115), VALID NHWC x HWIO convs + bias + ReLU (:12-21), HWC flatten (:6-9), fc + ReLU (:49-60),
softmax head (:84-89) -- is executed by libpaac_hip.so (csrc/net_fwd.hip, csrc/net_bwd.hip).  The attribute names the learner and
test.py touch (`input_ph`, `output`, `init(checkpoint_folder, saver, session)`) are kept.

New architectures: the reference lets users subclass Network and set `self.output` (networks.py:117-120,
README.md:80-83).  Here an architecture is a compiled kernel chain -- its geometry is a template argument of every kernel
-- so `define_architecture(name, convs, fc)` compiles a library for it on first use (paac_amd/build.py:
build_user_arch, about a minute of hipcc, cached in-tree) and returns the trunk class to mix into PolicyVNetwork, exactly
like NIPSNetwork / NatureNetwork.  Supported: two or three VALID conv layers of any kernel size and stride (the first
layer's size 4, 8, 12 or 16) with filter counts that are multiples of 16, and an fc width that is a multiple of 256.  The
reference trunks' layer shapes (conv 8x8 / 4, conv 4x4 / 2[, conv 3x3 / 1]) run on the MFMA data-gradient forms; other
shapes take the generic contraction for forward / weight gradient and a direct kernel for the data gradient.
"""
import glob
import logging
import os

import numpy as np
import torch

from . import _lib


class Placeholder(object):
    """Stand-in for a tf.placeholder / tf.Tensor handle: identity-compared key of feed_dict / fetches."""

    def __init__(self, name, dtype=None, shape=None):
        self.name, self.dtype, self.shape = name, dtype, shape

    def __repr__(self):
        return "<paac_amd.%s>" % self.name


ARCH_IDS = {"NIPS": _lib.ARCH_NIPS, "NATURE": _lib.ARCH_NATURE}


def resolve_device(device):
    """'/gpu:K' (reference flag syntax, train.py:80) or 'cuda:K' -> torch.device.  There is no CPU path."""
    d = str(device)
    if "cpu" in d:
        raise RuntimeError("paac_amd is MI355X-only: device %r is not supported (the CPU restatement of the "
                           "reference path lives in oracle/ and is test infrastructure)" % device)
    idx = int(d.rsplit(":", 1)[1]) if ":" in d else 0
    return torch.device("cuda", idx)


def initial_values(layout, rng):
    """{tensor name: float32 array}: every tensor ~ U(-d, d) with d = 1/sqrt(fan-in of ITS LAYER) -- conv: kh*kw*cin
    (networks.py:24-33, the conv bias uses the same d, :16), fc: number of inputs (networks.py:63-70, bias :53).
    Layout order is weights then biases per layer, so a bias takes the fan-in of the weight tensor before it."""
    values, fan_in = {}, None
    for t in layout["tensors"]:
        if t["name"].endswith("weights"):
            fan_in = int(np.prod(t["shape"][:-1]))
        d = 1.0 / np.sqrt(fan_in)
        values[t["name"]] = rng.uniform(-d, d, size=t["shape"]).astype(np.float32)
    return values


class Network(object):
    ARCH = None

    def __init__(self, conf):
        self.name = conf['name']
        self.num_actions = conf['num_actions']
        self.clip_norm = conf['clip_norm']
        self.clip_norm_type = conf['clip_norm_type']
        self.device = conf['device']
        self.loss_scaling = 5.0                                   # networks.py:112
        self.input_ph = Placeholder('input', np.uint8, [None, 84, 84, 4])
        self.selected_action_ph = Placeholder('selected_action', np.float32, [None, self.num_actions])
        self.input = Placeholder('input_scaled')                  # cast(u8) * (1/255), fused into conv1
        self.output = None
        self.layout = None
        self.params = None
        self.weights_changed = None      # callback(): the flat parameter buffer was rewritten from the host side

    # -- parameters --------------------------------------------------------------------------------
    def _allocate(self):
        if self.ARCH is None:
            raise NotImplementedError("Network must be subclassed (networks.py:117)")
        self.arch_id = ARCH_IDS[self.ARCH]
        self.layout = _lib.param_layout(self.arch_id, self.num_actions)
        self.torch_device = resolve_device(self.device)
        self.params = torch.zeros(self.layout["total"], dtype=torch.float32, device=self.torch_device)

    def tensor_names(self):
        return [t["name"] for t in self.layout["tensors"]]

    def set_parameters(self, named):
        """named: {tensor name: array of the reference shape} (conv HWIO, fc [in,out])."""
        host = np.zeros(self.layout["total"], dtype=np.float32)
        for t in self.layout["tensors"]:
            a = np.asarray(named[t["name"]], dtype=np.float32)
            if a.shape != t["shape"]:
                raise ValueError("%s: shape %s, expected %s" % (t["name"], a.shape, t["shape"]))
            host[t["offset"]:t["offset"] + t["size"]] = a.reshape(-1)
        self.params.copy_(torch.from_numpy(host))
        if self.weights_changed is not None:
            self.weights_changed()

    def get_parameters(self, flat=None):
        host = (self.params if flat is None else flat).detach().cpu().numpy()
        return {t["name"]: host[t["offset"]:t["offset"] + t["size"]].reshape(t["shape"]).copy()
                for t in self.layout["tensors"]}

    def make_saver(self, max_to_keep=5):
        """Saver over this network's variables under the reference's checkpoint names
        ('local_learning_1/conv1_weights' ... 'local_learning_2/critic_output_biases', session.checkpoint_key)."""
        from .session import Saver, checkpoint_key, tensor_of_key
        # (a bundle written by the reference's tf.train.Saver() also holds the optimizer slots: only the variables are taken)
        return Saver(lambda: {checkpoint_key(self.name, k): v for k, v in self.get_parameters().items()},
                     lambda d: self.set_parameters({tensor_of_key(k)[0]: v for k, v in d.items()
                                                    if tensor_of_key(k)[1] is None}),
                     max_to_keep=max_to_keep)

    def initialize(self, rng=None):
        """'torch' initialisation of the reference (the default of networks.py:12-13,49-50)."""
        self.set_parameters(initial_values(self.layout, np.random.RandomState() if rng is None else rng))

    def init(self, checkpoint_folder, saver, session):
        """networks.py:122-135: restore the latest checkpoint if there is one (step parsed from the file
        name after the last '-'), else initialise all variables.  Returns last_saving_step."""
        last_saving_step = 0
        path = saver.latest_checkpoint(checkpoint_folder) if saver is not None else None
        if path is None:
            logging.info('Initializing all variables')
            self.initialize()
        else:
            logging.info('Restoring network variables from previous run')
            saver.restore(session, path)
            last_saving_step = int(path[path.rindex('-') + 1:].split('.')[0])
        return last_saving_step


def define_architecture(name, convs, fc, build=True):
    """A user architecture: `convs` = [(filters, size, stride), ...] (2 or 3 layers), `fc` = width of the hidden fc layer.
    Builds (or finds) the library compiled for that geometry, makes it THE library of this process and returns a Network
    subclass -- use it like the reference's trunks:  class MyNet(PolicyVNetwork, define_architecture(...)): pass."""
    from . import build as builder
    convs = [tuple(int(v) for v in c) for c in convs]
    lib = builder.build_user_arch(convs, int(fc)) if build else builder.user_arch_library(convs, int(fc))[0]
    _lib.use_library(lib)
    have = _lib.user_arch()
    if have != (convs, int(fc)):
        raise RuntimeError("the loaded library holds the user architecture %r, not %r" % (have, (convs, int(fc))))
    ARCH_IDS[name] = _lib.ARCH_USER
    layers = len(convs) + 1

    class UserNetwork(Network):
        ARCH = name
        CONVS, FC = convs, int(fc)

        def __init__(self, conf):
            super(UserNetwork, self).__init__(conf)
            self.output = Placeholder('fc%d' % layers)

    UserNetwork.__name__ = "%sNetwork" % name
    return UserNetwork


class NIPSNetwork(Network):
    ARCH = "NIPS"       # conv(16,8,4) -> conv(32,4,2) -> fc 256   (networks.py:138-151)

    def __init__(self, conf):
        super(NIPSNetwork, self).__init__(conf)
        self.output = Placeholder('fc3')


class NatureNetwork(Network):
    ARCH = "NATURE"     # conv(32,8,4) -> conv(64,4,2) -> conv(64,3,1) -> fc 512   (networks.py:154-169)

    def __init__(self, conf):
        super(NatureNetwork, self).__init__(conf)
        self.output = Placeholder('fc4')
