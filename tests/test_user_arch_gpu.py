"""-m gpu: a user-defined architecture through the plugin surface (reference networks.py:117-120, README.md:80-83: "subclass
the trunk and mix it into PolicyVNetwork").  Here an architecture is a compiled geometry: networks.define_architecture
builds a library for it (paac_amd/build.py: build_user_arch) and a process holds one such library, so the checks run in a
child process.  Parity: a third architecture -- three conv layers of 16 / 32 / 32 filters, fc 256 -- and a fourth with a layer
shape the reference trunks do not have (conv 5x5 / 2) against the oracle with their ARCHS entries; "parity unpinned" at the TensorFlow boundary like the stock trunks (oracle/network.py header)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, %(root)r)
import torch
from oracle import network as onet
from paac_amd import _lib, hip_ops, networks
from paac_amd.policy_v_network import PolicyVNetwork

CONVS, FC, A, B = %(convs)r, %(fc)d, 6, 40
onet.ARCHS["TINY3"] = (CONVS, FC)
Trunk = networks.define_architecture("TINY3", CONVS, FC)          # builds / finds the library, makes it this process's
assert _lib.user_arch() == (CONVS, FC)

class TinyPolicyVNetwork(PolicyVNetwork, Trunk):                    # exactly like NaturePolicyVNetwork
    pass

# -- forward / loss / gradients against the oracle (same bars as the stock trunks) --------------------------------------
rs = np.random.RandomState(0)
params = onet.init_params("TINY3", A, rs, dtype=np.float32)
states = rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)
idx = rs.randint(0, A, B).astype(np.int32)
y, adv = rs.randn(B).astype(np.float32), rs.randn(B).astype(np.float32)
ctx = hip_ops.Context(_lib.ARCH_USER, A, max_batch=B)
names = [t["name"] for t in ctx.layout["tensors"]]
assert names[:8] == ["conv1_weights", "conv1_biases", "conv2_weights", "conv2_biases", "conv3_weights", "conv3_biases",
                     "fc4_weights", "fc4_biases"]
assert [t["shape"] for t in ctx.layout["tensors"]][6] == (onet.layer_dims("TINY3")[1], FC)
assert [t["shape"] for t in ctx.layout["tensors"]][2] == (CONVS[1][1], CONVS[1][1], CONVS[0][0], CONVS[1][0])
flat = np.zeros(ctx.layout["total"], dtype=np.float32)
for t in ctx.layout["tensors"]:
    flat[t["offset"]:t["offset"] + t["size"]] = params[t["name"]].reshape(-1)
p = torch.from_numpy(flat).cuda()
s = torch.from_numpy(states).cuda()
logits, probs, values = torch.zeros((B, A), device="cuda"), torch.zeros((B, A), device="cuda"), torch.zeros(B, device="cuda")
grad = torch.zeros(ctx.layout["total"], device="cuda")
ctx.forward(p, s, logits, probs, values)
ctx.loss_backward(p, s, torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(adv).cuda(), 0.02, grad)
torch.cuda.synchronize()
masks = {"a%%d" %% i: ctx.debug_activation(i, B).cpu().numpy() > 0 for i in (1, 2, 3)}
masks["h"] = ctx.debug_activation(4, B).cpu().numpy() > 0
L, g = onet.loss_and_grads(params, states, np.eye(A)[idx], y, adv, 0.02, "TINY3", dtype=np.float64, relu_masks=masks)
assert np.abs(logits.cpu().numpy() - L["logits"]).max() < 1e-4
assert np.abs(values.cpu().numpy() - L["v"]).max() < 1e-4
assert np.abs(probs.cpu().numpy() - L["pi"]).max() < 1e-5
gh, gn = grad.cpu().numpy(), onet.global_norm(g)
for t in ctx.layout["tensors"]:
    err = np.abs(gh[t["offset"]:t["offset"] + t["size"]] - g[t["name"]].reshape(-1)).max()
    assert err < 1e-4 * max(np.abs(g[t["name"]]).max(), 1e-3 * gn), (t["name"], err)
ctx.close()

# -- the stock Nature geometry is still served by this library; the NIPS one is not (it gave its place to the user's) -----
nat = hip_ops.Context(_lib.ARCH_NATURE, 4, max_batch=8)
nat.close()
try:
    hip_ops.Context(_lib.ARCH_NIPS, 4, max_batch=8)
    raise SystemExit("NIPS accepted by a user-architecture library")
except _lib.PaacHipError as exc:
    assert "not compiled into this library" in str(exc)

# -- the whole learner on it: device loop, checkpoint under the reference's naming, resume ----------------------------
from paac_amd import train
from paac_amd.paac import PAACLearner
args = train.get_arg_parser().parse_args(["-g", "qbert", "--user_arch", %(flag)r, "-ec", "8", "-ew", "0",
                                          "--max_global_steps", str(8 * 5 * 4), "-df", tempfile.mkdtemp(prefix="paac_user_")])
nc, ec = train.get_network_and_environment_creator(args)
learner = PAACLearner(nc, ec, args)
assert type(learner.network).__name__ == "UserPolicyVNetwork" and learner.network.ARCH == "USER"
learner.train()
w = learner.network.get_parameters()
assert all(np.isfinite(v).all() for v in w.values()) and set(w) >= {"conv3_weights", "fc4_weights"}
nc2, ec2 = train.get_network_and_environment_creator(args)
l2 = PAACLearner(nc2, ec2, args)
assert l2.init_network() == 160
assert all(np.array_equal(v, w[k]) for k, v in l2.network.get_parameters().items())
print("USER_ARCH_OK")
"""


@pytest.mark.parametrize("convs,fc,flag", [
    ([(16, 8, 4), (32, 4, 2), (32, 3, 1)], 256, "16,32,32,256"),                # the reference trunks' layer shapes, other widths
    # a layer the reference trunks do not have -- 5x5 / 2 -- and with it other spatial sizes all the way down (20 -> 8 -> 6):
    # forward and weight gradients on the generic MFMA contraction, data gradients on the direct kernel
    ([(32, 8, 4), (64, 5, 2), (64, 3, 1)], 512, "32:8:4,64:5:2,64:3:1,512"),
])
def test_user_architecture_runs_and_matches_the_oracle(convs, fc, flag):
    res = subprocess.run([sys.executable, "-c", _SCRIPT % dict(root=ROOT, convs=convs, fc=fc, flag=flag)], cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "USER_ARCH_OK" in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])
