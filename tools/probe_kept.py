"""GPU probe: the update from kept acting rows (paac_keep_next_forward) against the float64 oracle, per tensor: activation
errors of the kept rows and gradient errors (ReLU masks taken from the device).  PAAC_HIP_LIB selects the library."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import network as onet
from paac_amd import hip_ops
from test_hip_network import make_case, upload_params, unflatten, ARCH_ID

arch, A, T, N = "NATURE", 4, 5, int(os.environ.get("PROBE_N", "32"))
B = T * N
for scale in (1.0, 3.5):
    params, states, idx, _, _ = make_case(arch, A, B + N, seed=41, weight_scale=scale)
    ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B + N)
    p = upload_params(ctx, params)
    ctx.set_managed_weights(True); ctx.pack_weights(p)
    rs = np.random.RandomState(6)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    s, acts = dev(states), dev(idx[:B])
    rewards = dev(rs.choice([-1.0, 0.0, 1.0], size=(T, N)).astype(np.float32))
    masks = dev((rs.rand(T, N) > 0.2).astype(np.float32))
    n = ctx.layout["total"]
    values = torch.zeros((T, N), device="cuda"); probs = torch.zeros((N, A), device="cuda")
    for t in range(T):
        ctx.keep_next_forward(t * N)
        ctx.forward(p, s[t * N:(t + 1) * N], probs=probs, values=values[t])
    ctx.bootstrap_forward_trunk(p, s[B:], B)
    y, adv = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
    grad, loss = torch.zeros(n, device="cuda"), torch.zeros(4, device="cuda")
    ctx.loss_backward_returns(p, s[:B], acts, None, rewards, masks, values, 0.99, y, adv, 0.02, grad, loss, phase=0, forward_done=True)
    torch.cuda.synchronize()
    ref = onet.forward(params, states, arch, dtype=np.float64, keep=True)["cache"]
    m = {}
    for i, k in ((1, "a1"), (2, "a2"), (3, "a3"), (4, "h")):
        got = ctx.debug_activation(i, B + N).cpu().numpy().reshape(B + N, -1)[:B]
        want = ref[k].reshape(B + N, -1)[:B]
        e = np.abs(got - want)
        r, c = np.unravel_index(e.argmax(), e.shape)
        print("scale %.1f kept %s: max err %.2e (rel %.2e) at row %d col %d | mean err %.2e | flips %d" % (
            scale, k, e.max(), e.max() / np.abs(want).max(), r, c, e.mean(), int(((got > 0) != (want > 0)).sum())))
        m[k] = (got > 0).reshape(ref[k][:B].shape)
    L, g_ref = onet.loss_and_grads(params, states[:B], np.eye(A)[idx[:B]], y.cpu().numpy().astype(np.float64),
                                   adv.cpu().numpy().astype(np.float64), 0.02, arch, dtype=np.float64, relu_masks=m)
    got = unflatten(ctx, grad)
    gn = onet.global_norm(g_ref)
    for name, want in g_ref.items():
        err = np.abs(got[name] - want).max()
        print("   grad %-14s max err %.2e  / max %.2e = %.2e" % (name, err, np.abs(want).max(), err / max(np.abs(want).max(), 1e-3 * gn)))
    ctx.close()
