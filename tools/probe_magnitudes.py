"""GPU probe: forward / backward errors against the float64 oracle at trained-network magnitudes (weights scaled so that
|logits| and |v| reach 1-20): what the parity bars of tests/test_hip_network.py see, printed per scale and shape."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import network as onet
from paac_amd import hip_ops
from test_hip_network import make_case, upload_params, unflatten, ARCH_ID

for arch, A, B, managed in (("NATURE", 4, 32, True), ("NATURE", 4, 192, False), ("NATURE", 18, 128, True), ("NATURE", 4, 1536, False),
                            ("NIPS", 6, 40, False), ("NATURE", 18, 160, False)):
    for s in (1.0, 3.0, 3.5, 4.0):
        params, states, idx, y, adv = make_case(arch, A, B, seed=1, weight_scale=s)
        ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B)
        p = upload_params(ctx, params)
        if managed:
            ctx.set_managed_weights(True)
            ctx.pack_weights(p)
        st = torch.from_numpy(states).cuda()
        logits = torch.zeros((B, A), device="cuda"); probs = torch.zeros((B, A), device="cuda"); values = torch.zeros(B, device="cuda")
        ctx.forward(p, st, logits, probs, values)
        torch.cuda.synchronize()
        ref = onet.forward(params, states, arch, dtype=np.float64, keep=True)
        msg = "%s A=%d B=%d managed=%d scale=%.1f |logit|max %.3g |v|max %.3g pmin %.2g : dlogit %.2e dv %.2e dp %.2e" % (
            arch, A, B, managed, s, np.abs(ref["logits"]).max(), np.abs(ref["v"]).max(), ref["pi"].min(),
            np.abs(logits.cpu().numpy() - ref["logits"]).max(), np.abs(values.cpu().numpy() - ref["v"]).max(),
            np.abs(probs.cpu().numpy() - ref["pi"]).max())
        if not managed:
            nconv = 3 if arch == "NATURE" else 2
            for i in list(range(1, nconv + 1)) + [4]:
                got = ctx.debug_activation(i, B).cpu().numpy()
                want = ref["cache"]["a%d" % i if i < 4 else "h"].reshape(-1)
                msg += " a%d %.1e/%.1e" % (i, np.abs(got - want).max(), np.abs(got - want).max() / np.abs(want).max())
        if B <= 192 and not managed:
            grad = torch.zeros(ctx.layout["total"], device="cuda")
            ctx.loss_backward(p, st, torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(adv).cuda(), 0.02, grad)
            torch.cuda.synchronize()
            nconv = 3 if arch == "NATURE" else 2
            masks = {"a%d" % (i + 1): ctx.debug_activation(i + 1, B).cpu().numpy() > 0 for i in range(nconv)}
            masks["h"] = ctx.debug_activation(4, B).cpu().numpy() > 0
            L, g_ref = onet.loss_and_grads(params, states, np.eye(A)[idx], y, adv, 0.02, arch, dtype=np.float64, relu_masks=masks)
            got = unflatten(ctx, grad)
            gn = onet.global_norm(g_ref)
            worst = max(np.abs(got[k] - v).max() / max(np.abs(v).max(), 1e-3 * gn) for k, v in g_ref.items())
            msg += " | grad rel %.2e gn %.3g" % (worst, gn)
        print(msg, flush=True)
        ctx.close()
