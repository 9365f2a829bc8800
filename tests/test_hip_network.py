"""-m gpu: HIP forward / backward / optimizer vs the float64 oracle, through the C-ABI."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import network as onet

ARCH_ID = {"NIPS": 0, "NATURE": 1}


def make_case(arch, A, B, seed=0, weight_scale=1.0):
    rs = np.random.RandomState(seed)
    params = onet.init_params(arch, A, rs, dtype=np.float32)
    if weight_scale != 1.0:
        for k in params:
            params[k] = (params[k] * weight_scale).astype(np.float32)
    states = rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)
    idx = rs.randint(0, A, B).astype(np.int32)
    y = rs.randn(B).astype(np.float32)
    adv = rs.randn(B).astype(np.float32)
    return params, states, idx, y, adv


# Weights scaled so that the network behaves like a TRAINED one: |logits| and |v| of order 5-20 (initialisation-scale
# weights give +-0.03, where an absolute 1e-4 is a 0.3 % relative bar), some policies saturated (p < 6e-8: the
# `p - epsneg < 0` case of paac.py:42).  Measured on MI355X (tools/probe_magnitudes.py): logit / value errors 4e-6 .. 1.6e-5,
# activation errors <= 1.3e-6 of the layer's maximum, gradients 1e-6 relative.
TRAINED = 3.5        # |logits| ~ 5-10, |v| ~ 6-9
SATURATED = 4.0      # |logits| ~ 11-19, p_min down to 4e-14


def check_activations(ctx, ref_cache, arch, B, weight_scale):
    """conv / fc activations against the float64 oracle: the absolute 2e-5 (conv) / 5e-5 (fc) bars at initialisation scale, a
    RELATIVE bar -- 1e-5 of the layer's largest activation -- whatever the weights."""
    nconv = 3 if arch == "NATURE" else 2
    for i in list(range(1, nconv + 1)) + [4]:
        got = ctx.debug_activation(i, B).cpu().numpy()
        want = ref_cache["a%d" % i if i < 4 else "h"].reshape(-1)
        assert got.shape == want.shape
        err = np.abs(got - want).max()
        assert err <= 1e-5 * np.abs(want).max(), "layer %d: max abs err %g of max %g" % (i, err, np.abs(want).max())
        if weight_scale == 1.0:
            assert err < (2e-5 if i < 4 else 5e-5), "layer %d: max abs err %g" % (i, err)


def upload_params(ctx, params):
    from paac_amd import _lib
    lay = ctx.layout
    host = np.zeros(lay["total"], dtype=np.float32)
    for t in lay["tensors"]:
        host[t["offset"]:t["offset"] + t["size"]] = params[t["name"]].reshape(-1)
    return torch.from_numpy(host).cuda()


def unflatten(ctx, flat):
    host = flat.detach().cpu().numpy()
    return {t["name"]: host[t["offset"]:t["offset"] + t["size"]].reshape(t["shape"]) for t in ctx.layout["tensors"]}


@pytest.mark.parametrize("arch,A,B", [("NATURE", 4, 32), ("NATURE", 4, 160), ("NATURE", 18, 7), ("NATURE", 6, 1),
                                      ("NIPS", 6, 40), ("NIPS", 4, 33), ("NIPS", 18, 160),
                                      ("NATURE", 4, 192),      # the device loop's training forward: N*(T+1) rows
                                      ("NATURE", 4, 1536),     # 256 envs x (5+1): size heuristics instead of the tuned table
                                      ("NIPS", 6, 1280),
                                      # BASELINE configs[3] per-GPU shard (Qbert A=6, 32 envs, t_max 5): acting / training rows
                                      ("NATURE", 6, 32), ("NATURE", 6, 192),
                                      # BASELINE configs[4] per-GPU shard (Seaquest A=18, 128 envs, t_max 20)
                                      ("NATURE", 18, 128), ("NATURE", 18, 2688)])
def test_forward_parity(arch, A, B):
    _forward_parity(arch, A, B, 1.0)


@pytest.mark.parametrize("arch,A,B,scale", [
    ("NATURE", 4, 32, TRAINED), ("NATURE", 4, 192, TRAINED), ("NATURE", 4, 192, SATURATED),     # configs[1]: acting / training rows
    ("NATURE", 4, 256, TRAINED), ("NATURE", 4, 1536, SATURATED),                                # configs[2]
    ("NATURE", 6, 192, TRAINED),                                                                # configs[3] shard
    ("NATURE", 18, 128, SATURATED), ("NATURE", 18, 2688, TRAINED),                              # configs[4] shard
    ("NIPS", 6, 40, SATURATED), ("NIPS", 6, 1280, TRAINED)])                                    # configs[0]'s network
def test_forward_parity_at_trained_magnitudes(arch, A, B, scale):
    """north_star's "logits within 1e-4" in the regime it is about: logits and values of order 5-20 (1e-4 absolute = 1e-5
    relative), saturated policies included (policy_v_network.py:24-57, networks.py:84-89)."""
    ref = _forward_parity(arch, A, B, scale)
    assert np.abs(ref["logits"]).max() > 4.0 and np.abs(ref["v"]).max() > 2.0
    if scale == SATURATED:
        assert ref["pi"].min() < 5.9604645e-08          # rows the sampler sees as p - epsneg < 0


def _forward_parity(arch, A, B, weight_scale):
    from paac_amd import hip_ops
    params, states, idx, y, adv = make_case(arch, A, B, seed=1, weight_scale=weight_scale)
    ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=max(B, 8))
    p = upload_params(ctx, params)
    s = torch.from_numpy(states).cuda()
    logits = torch.zeros((B, A), device="cuda")
    probs = torch.zeros((B, A), device="cuda")
    values = torch.zeros((B,), device="cuda")
    ctx.forward(p, s, logits, probs, values)
    torch.cuda.synchronize()
    ref = onet.forward(params, states, arch, dtype=np.float64, keep=True)
    check_activations(ctx, ref["cache"], arch, B, weight_scale)
    # north_star tolerance: logits / values within 1e-4 of the reference-equivalent CPU path
    assert np.abs(logits.cpu().numpy() - ref["logits"]).max() < 1e-4
    assert np.abs(values.cpu().numpy() - ref["v"]).max() < 1e-4
    assert np.abs(probs.cpu().numpy() - ref["pi"]).max() < 1e-5
    ctx.close()
    return ref


@pytest.mark.parametrize("A,B,scale", [(18, 128, 1.0), (4, 256, 1.0), (6, 65, 1.0), (18, 200, 1.0), (4, 64, 1.0), (32, 100, 1.0),
                                       # trained-network magnitudes (see TRAINED / SATURATED): configs[1], [2], [4] acting rows
                                       (4, 32, SATURATED), (4, 256, TRAINED), (18, 128, SATURATED), (6, 32, TRAINED)])
def test_managed_acting_forward_parity(A, B, scale):
    """The acting forward of the learner (managed weights: conv tower -> fc with the head contractions in its epilogue ->
    heads finish by one workgroup up to 64 rows, by a few -- heads_finish_rows_kernel -- up to 256) against the oracle: the
    128- and 256-environment shards' policy step."""
    from paac_amd import hip_ops
    params, states, idx, y, adv = make_case("NATURE", A, B, seed=5, weight_scale=scale)
    ctx = hip_ops.Context(ARCH_ID["NATURE"], A, max_batch=B)
    p = upload_params(ctx, params)
    ctx.set_managed_weights(True)
    ctx.pack_weights(p)
    s = torch.from_numpy(states).cuda()
    logits = torch.zeros((B, A), device="cuda")
    probs = torch.zeros((B, A), device="cuda")
    values = torch.zeros((B,), device="cuda")
    ctx.forward(p, s, logits, probs, values)
    torch.cuda.synchronize()
    ref = onet.forward(params, states, "NATURE", dtype=np.float64)
    if scale != 1.0:
        assert np.abs(ref["logits"]).max() > 4.0 and np.abs(ref["v"]).max() > 2.0
    assert np.abs(logits.cpu().numpy() - ref["logits"]).max() < 1e-4
    assert np.abs(values.cpu().numpy() - ref["v"]).max() < 1e-4
    assert np.abs(probs.cpu().numpy() - ref["pi"]).max() < 1e-5
    # ... and the same bits as the unmanaged route's heads at up to 64 rows (one finishing workgroup either way)
    ctx.close()


@pytest.mark.parametrize("managed", [True, False])
@pytest.mark.parametrize("arch,A,B,scale", [("NATURE", 4, 32, 1.0), ("NATURE", 4, 32, SATURATED), ("NATURE", 6, 1, 1.0),
                                            ("NATURE", 18, 7, 1.0), ("NATURE", 4, 9, TRAINED), ("NATURE", 6, 24, 1.0),
                                            ("NIPS", 6, 32, 1.0), ("NIPS", 4, 57, TRAINED), ("NIPS", 6, 64, 1.0)])
def test_fc_quarter_tiles(arch, A, B, scale, managed, monkeypatch):
    """csrc/fc_heads.h: fc_heads_q_kernel (8 rows x 8 fc columns per workgroup, the even and the odd K groups of a pair riding
    in the two diagonal blocks of one MFMA; up to 32 rows of the stock widths, 64 for NIPS) against the float64 oracle and
    against the whole-tile kernel (PAAC_FC_QUARTER=0): the same sums in another order, ragged row counts included, from the
    tower's fragment-order hand-off (managed) and from plain rows."""
    from paac_amd import hip_ops
    params, states, idx, y, adv = make_case(arch, A, B, seed=9, weight_scale=scale)
    ref = onet.forward(params, states, arch, dtype=np.float64)
    out = []
    for quarter in ("1", "0"):
        monkeypatch.setenv("PAAC_FC_QUARTER", quarter)          # read when the context is created
        ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B)
        p = upload_params(ctx, params)
        if managed:
            ctx.set_managed_weights(True)
            ctx.pack_weights(p)
        s = torch.from_numpy(states).cuda()
        logits, probs, values = torch.zeros((B, A), device="cuda"), torch.zeros((B, A), device="cuda"), torch.zeros(B, device="cuda")
        ctx.prof_enable(True)
        ctx.forward(p, s, logits, probs, values)
        torch.cuda.synchronize()
        assert ("fc_fwd", B) in {(n, b) for n, b, _ in ctx.prof_read()}
        ctx.prof_enable(False)
        out.append((logits.cpu().numpy(), values.cpu().numpy(), probs.cpu().numpy()))
        ctx.close()
    for lg, v, pr in out:
        assert np.abs(lg - ref["logits"]).max() < 1e-4 and np.abs(v - ref["v"]).max() < 1e-4
        assert np.abs(pr - ref["pi"]).max() < 1e-5
    tol = 2e-5 * max(1.0, np.abs(ref["logits"]).max())
    assert np.abs(out[0][0] - out[1][0]).max() < tol and np.abs(out[0][1] - out[1][1]).max() < tol
    if scale == 1.0:
        assert not np.array_equal(out[0][0], out[1][0])      # two different kernels did run (another summation order)


@pytest.mark.parametrize("arch,A,B", [("NATURE", 4, 160), ("NATURE", 6, 40), ("NATURE", 18, 9), ("NIPS", 6, 40),
                                      ("NIPS", 4, 160), ("NATURE", 4, 320),
                                      ("NATURE", 4, 1280),     # 256 envs x t_max 5 (BASELINE configs[2]): heuristics
                                      ("NIPS", 6, 640),
                                      ("NATURE", 6, 160),      # configs[3] shard: 32 envs x t_max 5, Qbert action set
                                      ("NATURE", 18, 2560)])   # configs[4] shard: 128 envs x t_max 20, Seaquest action set
def test_backward_parity(arch, A, B):
    _backward_parity(arch, A, B, 1.0)


@pytest.mark.parametrize("arch,A,B,scale", [("NATURE", 4, 160, TRAINED), ("NATURE", 4, 160, SATURATED),     # configs[1]
                                            ("NATURE", 4, 1280, TRAINED),                                   # configs[2]
                                            ("NATURE", 18, 2560, SATURATED),                                # configs[4] shard
                                            ("NIPS", 6, 40, TRAINED)])
def test_backward_parity_at_trained_magnitudes(arch, A, B, scale):
    """Loss terms and every gradient at trained-network magnitudes (saturated policies: log(pi + 1e-30) far from 0,
    policy_v_network.py:29-35), same bars: 1e-4 relative."""
    _backward_parity(arch, A, B, scale)


def _backward_parity(arch, A, B, weight_scale):
    from paac_amd import hip_ops
    params, states, idx, y, adv = make_case(arch, A, B, seed=2, weight_scale=weight_scale)
    ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B)
    p = upload_params(ctx, params)
    s = torch.from_numpy(states).cuda()
    grad = torch.zeros(ctx.layout["total"], device="cuda")
    loss = torch.zeros(4, device="cuda")
    ctx.loss_backward(p, s, torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(adv).cuda(),
                      0.02, grad, loss)
    torch.cuda.synchronize()
    # ReLU masks are taken from the device activations: a pre-activation within rounding of 0 may legitimately
    # land on either side under a different summation order; the forward test bounds the activations themselves.
    nconv = 3 if arch == "NATURE" else 2
    masks = {"a%d" % (i + 1): ctx.debug_activation(i + 1, B).cpu().numpy() > 0 for i in range(nconv)}
    masks["h"] = ctx.debug_activation(4, B).cpu().numpy() > 0
    ref_fw = onet.forward(params, states, arch, dtype=np.float64, keep=True)["cache"]
    flips = sum(int((masks[k].reshape(-1) != (ref_fw[k].reshape(-1) > 0)).sum()) for k in masks)
    total = sum(m.size for m in masks.values())
    assert flips <= max(4, total * 2e-6), "%d of %d ReLU masks differ from the float64 oracle" % (flips, total)
    L, g_ref = onet.loss_and_grads(params, states, np.eye(A)[idx], y, adv, 0.02, arch, dtype=np.float64,
                                   relu_masks=masks)
    lo = loss.cpu().numpy()
    assert abs(lo[0] - L["loss"]) < 1e-4 * max(1.0, abs(L["loss"]))
    assert abs(lo[1] - L["actor"]) < 1e-4 * max(1.0, abs(L["actor"])) and abs(lo[2] - L["critic"]) < 1e-4 * max(1.0, abs(L["critic"]))
    assert abs(lo[3] - L["entropy"].mean()) < 1e-4
    got = unflatten(ctx, grad)
    gn_ref = onet.global_norm(g_ref)
    for name, want in g_ref.items():
        err = np.abs(got[name] - want).max()
        scale = max(np.abs(want).max(), 1e-3 * gn_ref)
        assert err / scale < 1e-4, "%s: max abs err %g (scale %g)" % (name, err, scale)
    # pads stay zero
    flat = grad.cpu().numpy()
    used = np.zeros(flat.shape, dtype=bool)
    for t in ctx.layout["tensors"]:
        used[t["offset"]:t["offset"] + t["size"]] = True
    assert np.all(flat[~used] == 0.0)
    ctx.close()


@pytest.mark.parametrize("arch,A,B", [("NATURE", 4, 160), ("NIPS", 6, 24)])
def test_backward_phases_compose(arch, A, B):
    """phase 1 (heads + fc -> gradient tail) then phase 2 (conv -> gradient head) == phase 0, bit for bit; phase 1
    alone leaves the conv head untouched (the data-parallel loop all-reduces the tail while phase 2 runs)."""
    from paac_amd import hip_ops
    params, states, idx, y, adv = make_case(arch, A, B, seed=5)
    ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B)
    p = upload_params(ctx, params)
    dev = [torch.from_numpy(a).cuda() for a in (states, idx, y, adv)]
    whole = torch.zeros(ctx.layout["total"], device="cuda")
    ctx.loss_backward(p, *dev, 0.02, whole)
    tail = [t["offset"] for t in ctx.layout["tensors"] if t["name"].startswith("fc")][0]
    split = torch.full((ctx.layout["total"],), 7.0, device="cuda")
    ctx.loss_backward(p, *dev, 0.02, split, phase=1)
    torch.cuda.synchronize()
    assert torch.all(split[:tail] == 7.0)
    assert torch.equal(split[tail:][whole[tail:] != 0], whole[tail:][whole[tail:] != 0])
    ctx.loss_backward(p, *dev, 0.02, split, forward_done=True, phase=2)
    torch.cuda.synchronize()
    used = torch.zeros(ctx.layout["total"], dtype=torch.bool, device="cuda")
    for t in ctx.layout["tensors"]:
        used[t["offset"]:t["offset"] + t["size"]] = True
    assert torch.equal(split[used], whole[used])
    with pytest.raises(Exception):
        ctx.loss_backward(p, *dev, 0.02, split, phase=4)
    ctx.close()


@pytest.mark.parametrize("arch,A,N,T", [("NATURE", 4, 32, 5), ("NIPS", 6, 8, 3)])
def test_train_forward_with_bootstrap_rows(arch, A, N, T):
    """The device loop runs ONE training forward over N*T rollout rows + N bootstrap rows, reads the bootstrap
    values from it and runs the backward over the first N*T rows: values must match paac_forward on the bootstrap
    observations (paac.py:140-142) and the gradient must match the plain N*T-row call."""
    from paac_amd import hip_ops
    B = N * T
    params, states, idx, y, adv = make_case(arch, A, B + N, seed=6)
    ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B + N)
    p = upload_params(ctx, params)
    s = torch.from_numpy(states).cuda()
    dev = [torch.from_numpy(a[:B]).cuda() for a in (idx, y, adv)]
    want = torch.zeros(ctx.layout["total"], device="cuda")
    ctx.loss_backward(p, s[:B], *dev, 0.02, want)
    v_boot = torch.zeros(N, device="cuda")
    ctx.forward(p, s[B:], values=v_boot)
    got = torch.zeros(ctx.layout["total"], device="cuda")
    values = torch.zeros(B + N, device="cuda")
    ctx.train_forward(p, s, values=values)
    ctx.loss_backward(p, s[:B], *dev, 0.02, got, forward_done=True)
    torch.cuda.synchronize()
    assert torch.allclose(values[B:], v_boot, rtol=0, atol=1e-5)
    ref = onet.forward(params, states[B:], arch, dtype=np.float64)
    assert np.abs(values[B:].cpu().numpy() - ref["v"]).max() < 1e-4
    scale = float(want.norm())
    assert float((got - want).abs().max()) < 1e-6 * scale
    ctx.close()


@pytest.mark.parametrize("mode,gscale,momentum", [("global", 1.0, 0.0), ("ignore", 1.0, 0.0), ("global", 0.5, 0.0),
                                                  ("global", 1.0, 0.9)])
def test_clip_rmsprop_parity(mode, gscale, momentum):
    from paac_amd import hip_ops, _lib
    arch, A = "NATURE", 6
    ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=8)
    n = ctx.layout["total"]
    rs = np.random.RandomState(3)
    var = rs.randn(n).astype(np.float32) * 0.1
    g = rs.randn(n).astype(np.float32) * (0.01 if mode == "global" else 0.001)
    ms = (1.0 + rs.rand(n)).astype(np.float32)
    mom = (rs.randn(n) * 1e-3).astype(np.float32)       # momentum 0 (the reference's setting): slot written, not read
    lr = np.float32(0.0224)
    dv, dg, dms, dmom = [torch.from_numpy(a.copy()).cuda() for a in (var, g, ms, mom)]
    lr_dev = torch.tensor([lr], device="cuda")
    gn_dev = torch.zeros(1, device="cuda")
    ctx.clip_rmsprop(dv, dg, dms, dmom, lr_dev, 0.99, momentum, 0.1, 3.0,
                     _lib.CLIP_GLOBAL if mode == "global" else _lib.CLIP_IGNORE, gscale, gn_dev)
    torch.cuda.synchronize()
    gs = g.astype(np.float64) * gscale
    gn = np.sqrt((gs ** 2).sum())
    f = 3.0 * min(1.0 / gn, 1.0 / 3.0) if mode == "global" else 1.0
    gc = gs * f
    ms_e = ms + (gc * gc - ms) * 0.01
    mom_e = momentum * mom.astype(np.float64) + lr * gc / np.sqrt(ms_e + 0.1)
    var_e = var - mom_e
    assert abs(gn_dev.item() - gn) / gn < 1e-5
    assert np.abs(dms.cpu().numpy() - ms_e).max() < 1e-6
    assert np.abs(dmom.cpu().numpy() - mom_e).max() < 1e-7
    assert np.abs(dv.cpu().numpy() - var_e).max() < 1e-6
    ctx.close()


@pytest.mark.parametrize("case", ["mixed", "all_negative", "negative_with_real_zeros"])
def test_grad_stats_are_the_reference_summaries(case):
    """actor_learner.py:85-87 -> logger_utils.py:23-33: mean / stddev / max / min of the flat raw and clipped gradients
    (the reference's flat gradient has no alignment pads: the pad zeros must not leak into max / min)."""
    from paac_amd import hip_ops, _lib
    ctx = hip_ops.Context(ARCH_ID["NATURE"], 6, max_batch=8)       # A=6: bias tensors end in pads
    lay = ctx.layout
    assert lay["total"] > lay["total_unpadded"]
    n = lay["total"]
    rs = np.random.RandomState(8)
    flat = np.zeros(n, dtype=np.float32)
    real = []
    for t in lay["tensors"]:
        v = (rs.randn(t["size"]) * 0.01).astype(np.float32)
        if case != "mixed":
            v = -np.abs(v) - np.float32(1e-6)
        if case == "negative_with_real_zeros" and t["size"] > 100:
            v[::17] = 0.0
        flat[t["offset"]:t["offset"] + t["size"]] = v
        real.append(v)
    real = np.concatenate(real).astype(np.float64) * 0.5           # grad_scale 0.5 (two ranks)
    z = lambda: torch.zeros(n, device="cuda")
    ctx.clip_rmsprop(z(), torch.from_numpy(flat).cuda(), torch.ones(n, device="cuda"), z(),
                     torch.tensor([0.01], device="cuda"), 0.99, 0.0, 0.1, 3.0, _lib.CLIP_GLOBAL, 0.5)
    got = ctx.grad_stats(3.0, _lib.CLIP_GLOBAL)
    gn = np.sqrt((real ** 2).sum())
    f = 3.0 * min(1.0 / gn, 1.0 / 3.0)
    assert abs(got["global_norm"] - gn) / gn < 1e-5
    for name, x in (("raw_gradients", real), ("clipped_gradients", real * f)):
        want = dict(mean=x.mean(), stddev=np.sqrt(((x - x.mean()) ** 2).mean()), max=x.max(), min=x.min())
        for k, w in want.items():
            assert abs(got[name][k] - w) <= 2e-5 * max(abs(w), np.abs(x).max() * 1e-2), (case, name, k, got[name][k], w)
    if case == "all_negative":
        assert got["raw_gradients"]["max"] < 0.0                   # no pad zero leaked in
    if case == "negative_with_real_zeros":
        assert got["raw_gradients"]["max"] == 0.0
    ctx.close()


def test_errors_are_loud():
    from paac_amd import hip_ops, _lib
    ctx = hip_ops.Context(1, 4, max_batch=8)
    p = torch.zeros(ctx.layout["total"], device="cuda")
    with pytest.raises(ValueError):
        ctx.forward(p, torch.zeros((9, 84, 84, 4), dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError):
        ctx.forward(p, torch.zeros((4, 84, 84, 3), dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError):
        ctx.forward(p[:100], torch.zeros((4, 84, 84, 4), dtype=torch.uint8, device="cuda"))
    with pytest.raises(_lib.PaacHipError):
        hip_ops.Context(1, 99, max_batch=8)
    ctx.close()


@pytest.mark.parametrize("B,fwd_cfg,wgrad_cfg", [(32, 107, None), (160, 104, 102), (192, 100, 100), (37, 112, 108)])
def test_conv1_exact_bf16_path(B, fwd_cfg, wgrad_cfg):
    """conv1 on v_mfma_f32_16x16x32_bf16 with the u8 pixels as exact bf16 and the fp32 weights / output gradients split
    exactly into three bf16 terms (dmm.h: XB): same parity bars as the fp32-MFMA path, and the two paths agree to
    fp32 rounding."""
    from paac_amd import hip_ops, _lib
    arch, A = "NATURE", 4
    params, states, idx, y, adv = make_case(arch, A, B, seed=8)
    ref = onet.forward(params, states, arch, dtype=np.float64, keep=True)
    outs = {}
    for label, fc, wc in (("fp32", None, None), ("bf16x3", fwd_cfg, wgrad_cfg)):
        ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B)
        if fc is not None:
            for cls in (0, 1):
                _lib.check(ctx.lib.paac_debug_set_tuning(ctx.handle, 0, cls, fc, 0, -1), "set_tuning")
            if wc is not None:
                _lib.check(ctx.lib.paac_debug_set_tuning(ctx.handle, 10, 1 if B > 64 else 0, wc, 32, 2), "set_tuning")
        p = upload_params(ctx, params)
        s = torch.from_numpy(states).cuda()
        logits = torch.zeros((B, A), device="cuda")
        values = torch.zeros((B,), device="cuda")
        ctx.forward(p, s, logits, None, values)
        a1 = ctx.debug_activation(1, B).cpu().numpy()
        assert np.abs(a1 - ref["cache"]["a1"].reshape(-1)).max() < 2e-5
        assert np.abs(logits.cpu().numpy() - ref["logits"]).max() < 1e-4
        assert np.abs(values.cpu().numpy() - ref["v"]).max() < 1e-4
        grad = torch.zeros(ctx.layout["total"], device="cuda")
        ctx.loss_backward(p, s, torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(adv).cuda(),
                          0.02, grad)
        torch.cuda.synchronize()
        outs[label] = (a1, unflatten(ctx, grad))
        ctx.close()
    a_ref, g_ref = outs["fp32"]
    a_new, g_new = outs["bf16x3"]
    assert np.abs(a_new - a_ref).max() < 2e-6
    for name in ("conv1_weights", "conv1_biases"):
        scale = max(np.abs(g_ref[name]).max(), 1e-6)
        assert np.abs(g_new[name] - g_ref[name]).max() / scale < 2e-5, name


@pytest.mark.parametrize("arch,A,B,cfg_fwd,cfg_dgrad,cfg_wgrad", [("NATURE", 4, 160, 204, 205, 200), ("NATURE", 6, 32, 201, 201, 201),
                                                                  ("NIPS", 6, 72, 207, 210, 203), ("NATURE", 4, 45, 211, 209, 203),
                                                                  ("NATURE", 4, 192, 212, 211, 201), ("NIPS", 18, 33, 210, 209, 200)])
@pytest.mark.parametrize("scale", [1.0, SATURATED])
def test_split_bf16_path(arch, A, B, cfg_fwd, cfg_dgrad, cfg_wgrad, scale):
    """Every contraction on the six-product split-bf16 path (dmm.h: XB = 2; the ids are entries of
    PAAC_*_SPLIT_CFGS, i.e. really instantiated there): the same parity bars as the fp32 MFMA path -- logits / values
    within 1e-4, gradients within 1e-4 of the float64 oracle."""
    from paac_amd import hip_ops, _lib
    params, states, idx, y, adv = make_case(arch, A, B, seed=9, weight_scale=scale)
    ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B)
    cls = 1 if B > 64 else 0
    for op in (1, 2, 3):
        _lib.check(ctx.lib.paac_debug_set_tuning(ctx.handle, op, cls, cfg_fwd, 8 if op == 3 else 0, -1), "set_tuning")
    for op in (5, 7, 9):
        _lib.check(ctx.lib.paac_debug_set_tuning(ctx.handle, op, cls, cfg_dgrad, 0, -1), "set_tuning")
    for op in (4, 6, 8):
        _lib.check(ctx.lib.paac_debug_set_tuning(ctx.handle, op, cls, cfg_wgrad, 1 if op == 4 else 16, -1), "set_tuning")
    p = upload_params(ctx, params)
    s = torch.from_numpy(states).cuda()
    logits = torch.zeros((B, A), device="cuda")
    values = torch.zeros((B,), device="cuda")
    ctx.forward(p, s, logits, None, values)
    ref = onet.forward(params, states, arch, dtype=np.float64, keep=True)
    assert np.abs(logits.cpu().numpy() - ref["logits"]).max() < 1e-4
    assert np.abs(values.cpu().numpy() - ref["v"]).max() < 1e-4
    nconv = 3 if arch == "NATURE" else 2
    check_activations(ctx, ref["cache"], arch, B, scale)
    grad = torch.zeros(ctx.layout["total"], device="cuda")
    ctx.loss_backward(p, s, torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(adv).cuda(), 0.02,
                      grad)
    torch.cuda.synchronize()
    masks = {"a%d" % (i + 1): ctx.debug_activation(i + 1, B).cpu().numpy() > 0 for i in range(nconv)}
    masks["h"] = ctx.debug_activation(4, B).cpu().numpy() > 0
    L, g_ref = onet.loss_and_grads(params, states, np.eye(A)[idx], y, adv, 0.02, arch, dtype=np.float64, relu_masks=masks)
    got = unflatten(ctx, grad)
    gn_ref = onet.global_norm(g_ref)
    for name, want in g_ref.items():
        err = np.abs(got[name] - want).max()
        scale = max(np.abs(want).max(), 1e-3 * gn_ref)
        assert err / scale < 1e-4, "%s: max abs err %g (scale %g)" % (name, err, scale)
    ctx.close()


@pytest.mark.parametrize("regions", [8, 4, 2, 1])
@pytest.mark.parametrize("B,A,scale", [(5, 4, 1.0), (33, 6, 1.0), (32, 4, SATURATED)])
def test_conv_tower_variants(regions, B, A, scale):
    """csrc/tower.h: the fused conv1->conv2->conv3 launch in each of its region layouts (4 overlapping 4x4 regions, 2
    halves, one workgroup per sample) against the float64 oracle -- conv1 / conv2 / conv3 activations, logits, values --
    and against the unfused per-layer kernels (PAAC_TOWER=0 is the same arithmetic in a different summation order)."""
    from paac_amd import _lib, hip_ops
    params, states, idx, y, adv = make_case("NATURE", A, B, seed=3, weight_scale=scale)
    ctx = hip_ops.Context(ARCH_ID["NATURE"], A, max_batch=B)
    for cls in (0, 1, 2):
        _lib.check(ctx.lib.paac_debug_set_tuning(ctx.handle, 11, cls, regions, 0, -1), "set_tuning")
    p = upload_params(ctx, params)
    s = torch.from_numpy(states).cuda()
    logits = torch.zeros((B, A), device="cuda")
    values = torch.zeros((B,), device="cuda")
    ctx.forward(p, s, logits=logits, values=values)
    torch.cuda.synchronize()
    ref = onet.forward(params, states, "NATURE", dtype=np.float64, keep=True)
    check_activations(ctx, ref["cache"], "NATURE", B, scale)
    assert np.abs(logits.cpu().numpy() - ref["logits"]).max() < 1e-4
    assert np.abs(values.cpu().numpy() - ref["v"]).max() < 1e-4
    # managed mode: the acting forward keeps only conv3's output; weights come from an explicit pack
    ctx.set_managed_weights(True)
    ctx.pack_weights(p)
    logits2 = torch.zeros((B, A), device="cuda")
    ctx.forward(p, s, logits=logits2)
    assert torch.equal(logits, logits2)
    # a stale pack is really used in managed mode (the contract: the owner of the writes re-packs) ...
    p2 = p * 1.5
    ctx.forward(p2, s, logits=logits2)
    ctx.pack_weights(p2)
    logits3 = torch.zeros((B, A), device="cuda")
    ctx.forward(p2, s, logits=logits3)
    assert not torch.equal(logits2, logits3)
    ctx.close()


@pytest.mark.parametrize("arch,A", [("NATURE", 4), ("NIPS", 6)])
def test_optimizer_step_keeps_packed_weights_current(arch, A):
    """paac_clip_rmsprop also rewrites the pre-split conv planes (tower.h) and the fragment-ordered fc weights (fc_heads.h)
    from the values it has just computed: a managed-mode forward right after an update equals the forward after an explicit
    paac_pack_weights of the same parameters, bit for bit -- and differs from the forward before the update."""
    from paac_amd import hip_ops, _lib
    B = 24
    params, states, idx, y, adv = make_case(arch, A, B, seed=9)
    ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B)
    p = upload_params(ctx, params)
    s = torch.from_numpy(states).cuda()
    ctx.set_managed_weights(True)
    ctx.pack_weights(p)
    n = ctx.layout["total"]
    before = torch.zeros((B, A), device="cuda")
    ctx.forward(p, s, logits=before)
    grad = torch.zeros(n, device="cuda")
    ctx.loss_backward(p, s, torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(adv).cuda(), 0.02, grad)
    ctx.clip_rmsprop(p, grad, torch.ones(n, device="cuda"), torch.zeros(n, device="cuda"), torch.tensor([0.05], device="cuda"),
                     0.99, 0.0, 0.1, 3.0, _lib.CLIP_GLOBAL)
    after = torch.zeros((B, A), device="cuda")
    ctx.forward(p, s, logits=after)              # packed copies as left by the optimizer step
    ctx.pack_weights(p)
    repacked = torch.zeros((B, A), device="cuda")
    ctx.forward(p, s, logits=repacked)
    torch.cuda.synchronize()
    assert torch.equal(after, repacked)
    assert not torch.equal(after, before)
    ref = onet.forward(unflatten(ctx, p), states, arch, dtype=np.float64)
    assert np.abs(after.cpu().numpy() - ref["logits"]).max() < 1e-4
    # the backward pass reads packed data-gradient weights too (dgrad_tower.h): same check on the gradient
    dev = [torch.from_numpy(a).cuda() for a in (idx, y, adv)]
    ctx.clip_rmsprop(p, grad, torch.ones(n, device="cuda"), torch.zeros(n, device="cuda"), torch.tensor([0.05], device="cuda"),
                     0.99, 0.0, 0.1, 3.0, _lib.CLIP_GLOBAL)
    g_after = torch.zeros(n, device="cuda")
    ctx.loss_backward(p, s, *dev, 0.02, g_after)
    ctx.pack_weights(p)
    g_repacked = torch.zeros(n, device="cuda")
    ctx.loss_backward(p, s, *dev, 0.02, g_repacked)
    torch.cuda.synchronize()
    assert torch.equal(g_after, g_repacked)
    ctx.close()


@pytest.mark.parametrize("arch,A,B", [("NATURE", 4, 160), ("NIPS", 6, 24), ("NATURE", 18, 40)])
def test_deferred_slab_reduction_is_bit_identical(arch, A, B):
    """paac_loss_backward(phase=3) leaves the split-K slab sums of the conv weight gradients to the norm pass of the next
    paac_clip_rmsprop: norm, updated weights, optimizer slots AND the completed gradient buffer equal the phase-0 route
    bit for bit; a pending reduction for another buffer is refused."""
    from paac_amd import hip_ops, _lib
    params, states, idx, y, adv = make_case(arch, A, B, seed=21)
    ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B)
    n = ctx.layout["total"]
    s = torch.from_numpy(states).cuda()
    dev = [torch.from_numpy(a).cuda() for a in (idx, y, adv)]
    out = []
    for phase in (0, 3):
        p = upload_params(ctx, params)
        grad = torch.zeros(n, device="cuda")
        ms, mom, gn = torch.ones(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(1, device="cuda")
        ctx.loss_backward(p, s, *dev, 0.02, grad, phase=phase)
        if phase == 3:
            with pytest.raises(RuntimeError):
                ctx.clip_rmsprop(p, torch.zeros(n, device="cuda"), ms, mom, torch.tensor([0.05], device="cuda"), 0.99, 0.0,
                                 0.1, 3.0, _lib.CLIP_GLOBAL)
        ctx.clip_rmsprop(p, grad, ms, mom, torch.tensor([0.05], device="cuda"), 0.99, 0.0, 0.1, 3.0, _lib.CLIP_GLOBAL,
                         gnorm_out=gn)
        stats = ctx.grad_stats(3.0, _lib.CLIP_GLOBAL)
        torch.cuda.synchronize()
        out.append((p.cpu().numpy(), grad.cpu().numpy(), ms.cpu().numpy(), float(gn.item()), stats))
    for a, b in zip(out[0][:3], out[1][:3]):
        assert np.array_equal(a, b)
    assert out[0][3] == out[1][3] and out[0][4] == out[1][4]
    ctx.close()


@pytest.mark.parametrize("arch,A,T,N,phase", [("NATURE", 4, 5, 32, 0), ("NATURE", 6, 5, 32, 3), ("NATURE", 18, 20, 4, 0),
                                              ("NATURE", 4, 5, 8, 0), ("NIPS", 6, 5, 16, 0), ("NATURE", 4, 5, 32, 1)])
def test_trunk_forward_with_heads_in_the_backward_is_bit_identical(arch, A, T, N, phase):
    """paac_train_forward_trunk + paac_loss_backward_returns(v_boot=NULL): the heads forward of the rollout rows and the
    value head of the bootstrap rows ride in the backward's first launch (three-conv network, whole backward), or run as
    the launch that was left out (other networks / phases / small batches).  Gradient, returns, loss terms, schedule and
    the update equal paac_train_forward + paac_loss_backward_returns(v_boot=values[B:]) bit for bit."""
    from paac_amd import hip_ops, _lib
    B = T * N
    params, states, idx, _, _ = make_case(arch, A, B + N, seed=31)
    ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B + N)
    rs = np.random.RandomState(5)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    s, acts = dev(states), dev(idx[:B])
    rewards = dev(rs.choice([-1.0, 0.0, 1.0], size=(T, N)).astype(np.float32))
    masks = dev((rs.rand(T, N) > 0.2).astype(np.float32))
    values = dev(rs.randn(T, N).astype(np.float32))
    n = ctx.layout["total"]
    out = []
    for fused in (False, True):
        p = upload_params(ctx, params)
        y, adv = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
        gstep = torch.tensor([1000], dtype=torch.int64, device="cuda")
        tick = torch.tensor([7], dtype=torch.int64, device="cuda")
        lr = torch.zeros(1, device="cuda")
        grad, loss = torch.zeros(n, device="cuda"), torch.zeros(4, device="cuda")
        kw = dict(global_step_dev=gstep, increment=B, initial_lr=0.0224, lr_annealing_steps=80000000, lr_out_dev=lr,
                  tick_dev=tick, tick_inc=T, forward_done=True)
        if fused:
            ctx.train_forward_trunk(p, s)
            v_boot = None
        else:
            vt = torch.zeros(B + N, device="cuda")
            ctx.train_forward(p, s, values=vt)
            v_boot = vt[B:]
        if phase == 1:
            ctx.loss_backward_returns(p, s[:B], acts, v_boot, rewards, masks, values, 0.99, y, adv, 0.02, grad, loss, phase=1, **kw)
            ctx.loss_backward(p, s[:B], acts, y, adv, 0.02, grad, loss, forward_done=True, phase=2)
        else:
            ctx.loss_backward_returns(p, s[:B], acts, v_boot, rewards, masks, values, 0.99, y, adv, 0.02, grad, loss, phase=phase, **kw)
        ms, mom, gn = torch.ones(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(1, device="cuda")
        ctx.clip_rmsprop(p, grad, ms, mom, lr, 0.99, 0.0, 0.1, 3.0, _lib.CLIP_GLOBAL, gnorm_out=gn)
        torch.cuda.synchronize()
        out.append([t.cpu().numpy() for t in (grad, y, adv, loss, gstep, tick, lr, p, ms, gn)])
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    assert np.isfinite(out[0][0]).all() and np.abs(out[0][0]).max() > 0
    ctx.close()


@pytest.mark.parametrize("arch,A,T,N", [("NATURE", 4, 5, 8), ("NIPS", 6, 20, 3)])
def test_backward_with_fused_returns_equals_separate_calls(arch, A, T, N):
    """paac_loss_backward_returns == paac_nstep_returns_tick + paac_loss_backward: y, adv, lr, counters and the whole
    gradient bit for bit (the returns arithmetic inside the heads-gradient launch is the returns kernel's)."""
    from paac_amd import hip_ops
    B = T * N
    params, states, idx, _, _ = make_case(arch, A, B, seed=12)
    ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B)
    p = upload_params(ctx, params)
    rs = np.random.RandomState(3)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    s, acts = dev(states), dev(idx)
    v_boot = dev(rs.randn(N).astype(np.float32))
    rewards = dev(rs.choice([-1.0, 0.0, 1.0], size=(T, N)).astype(np.float32))
    masks = dev((rs.rand(T, N) > 0.2).astype(np.float32))
    values = dev(rs.randn(T, N).astype(np.float32))
    n = ctx.layout["total"]
    out = []
    for fused in (False, True):
        y, adv = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
        gstep = torch.tensor([1000], dtype=torch.int64, device="cuda")
        tick = torch.tensor([7], dtype=torch.int64, device="cuda")
        lr = torch.zeros(1, device="cuda")
        grad, loss = torch.zeros(n, device="cuda"), torch.zeros(4, device="cuda")
        if fused:
            ctx.loss_backward_returns(p, s, acts, v_boot, rewards, masks, values, 0.99, y, adv, 0.02, grad, loss,
                                      global_step_dev=gstep, increment=B, initial_lr=0.0224, lr_annealing_steps=80000000,
                                      lr_out_dev=lr, tick_dev=tick, tick_inc=T)
        else:
            hip_ops.nstep_returns_tick(v_boot, rewards, masks, values, 0.99, y, adv, gstep, B, 0.0224, 80000000, lr, tick, T)
            ctx.loss_backward(p, s, acts, y, adv, 0.02, grad, loss)
        torch.cuda.synchronize()
        out.append((y, adv, gstep, tick, lr, grad, loss))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert int(out[1][2].item()) == 1000 + B and int(out[1][3].item()) == 7 + T
    ctx.close()


def test_profiling_hooks_name_the_launches_and_their_instruction_mix():
    """paac_prof_read / paac_prof_read_mix: every launch of one update carries its family -- paired launches named for
    what the launcher paired -- and the MFMA products per fp32 multiply of each contraction body (1 fp32 MFMA, 3 exact
    bf16, 6 split bf16): what bench.py's roofline fractions are priced with."""
    from paac_amd import hip_ops
    B = 160
    params, states, idx, y, adv = make_case("NATURE", 4, B + 32, seed=3)
    ctx = hip_ops.Context(ARCH_ID["NATURE"], 4, max_batch=B + 32)
    p = upload_params(ctx, params)
    ctx.set_managed_weights(True)
    ctx.pack_weights(p)
    s = torch.from_numpy(states).cuda()
    grad = torch.zeros(ctx.layout["total"], device="cuda")
    probs = torch.zeros((32, 4), device="cuda")
    ctx.prof_enable(True)
    ctx.forward(p, s[:32], probs=probs)
    ctx.train_forward_trunk(p, s)
    ctx.loss_backward(p, s[:B], torch.from_numpy(idx[:B]).cuda(), torch.from_numpy(y[:B]).cuda(),
                      torch.from_numpy(adv[:B]).cuda(), 0.02, grad, forward_done=True)
    torch.cuda.synchronize()
    recs = ctx.prof_read(with_mix=True)
    ctx.prof_enable(False)
    got = {(name, batch): mix for name, batch, ms, mix in recs}
    assert all(ms > 0 for _, _, ms, _ in recs)
    assert got[("conv_tower", 32)] == (3, 6) and got[("conv_tower", 192)] == (3, 6)
    assert got[("fc_fwd", 32)] == (1,)                       # fc_heads_kernel: fp32 MFMA
    assert got[("fc_fwd", 192)] == (6,) and got[("fc_dgrad", B)] == (6,)
    assert got[("fc_conv3_wgrad", B)] == (1, 1) and got[("conv2_conv1_wgrad", B)] == (1, 3)
    assert got[("dgrad_tower", B)] == (6,)
    assert got[("heads_fwd", 32)] == () and ("conv3_wgrad", B) not in got and ("conv2_wgrad", B) not in got
    assert ctx.prof_read() == []                              # the table was cleared
    ctx.close()


@pytest.mark.parametrize("arch,A,T,N,scale", [("NATURE", 4, 5, 32, 1.0), ("NATURE", 4, 5, 32, TRAINED), ("NATURE", 18, 3, 128, 1.0),
                                              ("NATURE", 6, 2, 7, 1.0), ("NATURE", 4, 2, 256, TRAINED),
                                              ("NIPS", 6, 5, 32, 1.0), ("NIPS", 6, 5, 8, TRAINED), ("NIPS", 4, 2, 100, 1.0)])
def test_update_from_kept_acting_rows(arch, A, T, N, scale):
    """paac_keep_next_forward + paac_bootstrap_forward_trunk: the T acting forwards keep their rows in the training activation
    set, the bootstrap observations run as one acting-shaped forward, and the backward starts from there -- no training
    forward (weights are frozen inside a cycle: paac.py:105 and :163-165 evaluate the same network on the same
    observations).  Against the recomputed route (paac_train_forward_trunk): same returns, gradient equal to fp32 summation
    order; against the float64 oracle: activations of the kept rows, gradient within the usual bars."""
    from paac_amd import hip_ops, _lib
    B = T * N
    params, states, idx, _, _ = make_case(arch, A, B + N, seed=41, weight_scale=scale)
    ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B + N)
    p = upload_params(ctx, params)
    ctx.set_managed_weights(True)
    ctx.pack_weights(p)
    rs = np.random.RandomState(6)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    s, acts = dev(states), dev(idx[:B])
    rewards = dev(rs.choice([-1.0, 0.0, 1.0], size=(T, N)).astype(np.float32))
    masks = dev((rs.rand(T, N) > 0.2).astype(np.float32))
    n = ctx.layout["total"]
    out = []
    for kept in (False, True):
        values = torch.zeros((T, N), device="cuda")
        probs = torch.zeros((N, A), device="cuda")
        for t in range(T):                          # the acting forwards of the rollout (they produce values[t] either way)
            if kept:
                ctx.keep_next_forward(t * N)
            ctx.forward(p, s[t * N:(t + 1) * N], probs=probs, values=values[t])
        if kept:
            ctx.bootstrap_forward_trunk(p, s[B:], B)
        else:
            ctx.train_forward_trunk(p, s)
        y, adv = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
        grad, loss = torch.zeros(n, device="cuda"), torch.zeros(4, device="cuda")
        ctx.loss_backward_returns(p, s[:B], acts, None, rewards, masks, values, 0.99, y, adv, 0.02, grad, loss, phase=3,
                                  forward_done=True)
        ms, mom, gn = torch.ones(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(1, device="cuda")
        q = p.clone()
        ctx.clip_rmsprop(q, grad, ms, mom, torch.tensor([0.01], device="cuda"), 0.99, 0.0, 0.1, 3.0, _lib.CLIP_GLOBAL, gnorm_out=gn)
        ctx.pack_weights(p)                          # the optimizer step re-packed q's weights: back to p's for the next route
        torch.cuda.synchronize()
        layers = (1, 2, 3, 4) if arch == "NATURE" else (1, 2, 4)
        acts_kept = [ctx.debug_activation(i, B + N).cpu().numpy() for i in layers] if kept else None
        out.append(dict(grad=grad.cpu().numpy(), y=y.cpu().numpy(), adv=adv.cpu().numpy(), loss=loss.cpu().numpy(),
                        gn=float(gn.item()), values=values.cpu().numpy(), acts=acts_kept))
    a, b = out
    assert np.array_equal(a["values"], b["values"])                         # the acting forward itself does not change
    assert np.abs(a["y"] - b["y"]).max() <= 2e-6 * max(1.0, np.abs(a["y"]).max())      # bootstrap values: another summation order
    scale_g = np.abs(a["grad"]).max()
    assert np.abs(a["grad"] - b["grad"]).max() <= 2e-5 * scale_g
    assert abs(a["gn"] - b["gn"]) <= 1e-5 * a["gn"]
    assert np.abs(a["loss"] - b["loss"]).max() <= 1e-5 * max(1.0, np.abs(a["loss"]).max())
    # the kept rows against the oracle
    ref = onet.forward(params, states, arch, dtype=np.float64, keep=True)["cache"]
    for i, got in zip(layers, b["acts"]):
        want = ref["a%d" % i if i < 4 else "h"]
        want = want.reshape(B + N, -1)[:B].reshape(-1)          # the update's B rows: of the bootstrap rows only the fc
        got = got.reshape(B + N, -1)[:B].reshape(-1)            # activations are kept (in the slab, read for v only)
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), "kept layer %d" % i
    # the counter-based sampler's forward (split-K fc + per-row heads launch with the sampler inside) keeps its rows too:
    # same kept activations as the fc + head partials route, to fp32 summation order
    if N <= 64:
        acts_i = torch.zeros(N, dtype=torch.int32, device="cuda")
        tick = torch.zeros(1, dtype=torch.int64, device="cuda")
        ctx.keep_next_forward(0)
        ctx.forward_sample(p, s[:N], 7, tick, 0, 0, acts_i, probs=torch.zeros((N, A), device="cuda"), values=torch.zeros(N, device="cuda"))
        torch.cuda.synchronize()
        nl = 3 if arch == "NATURE" else 2
        for i in list(range(1, nl + 1)):
            got = ctx.debug_activation(20 + i, B + N).cpu().numpy().reshape(B + N, -1)[:N].reshape(-1)      # the training set
            want = ref["a%d" % i].reshape(B + N, -1)[:N].reshape(-1)
            assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), "philox-kept layer %d" % i
    # a forward that cannot keep its rows says so
    ctx.keep_next_forward(B)                        # rows [B, B + N + 1) do not fit the training set
    with pytest.raises(Exception):
        ctx.forward(p, s[:N + 1], values=torch.zeros(N + 1, device="cuda"))
    ctx.close()


@pytest.mark.parametrize("regions", [9, 4, 1])
@pytest.mark.parametrize("B,A,scale", [(5, 6, 1.0), (33, 4, 1.0), (32, 6, SATURATED), (160, 6, TRAINED)])
def test_conv_tower2_variants(regions, B, A, scale, monkeypatch):
    """csrc/tower2.h: the reference's DEFAULT architecture (networks.py:138-151, NIPS) with conv1 -> conv2 in one launch, in
    each of its region layouts (nine 3x3 regions, four 5x5, one workgroup per sample), against the float64 oracle --
    activations, logits, values -- and against the per-layer kernels (PAAC_TOWER=0: the same arithmetic in another order);
    then the managed (acting) mode: fragment-order hand-off to the fc kernel, same bits."""
    from paac_amd import _lib, hip_ops
    params, states, idx, y, adv = make_case("NIPS", A, B, seed=13, weight_scale=scale)
    ctx = hip_ops.Context(ARCH_ID["NIPS"], A, max_batch=B)
    for cls in (0, 1, 2):
        _lib.check(ctx.lib.paac_debug_set_tuning(ctx.handle, 11, cls, regions, 0, -1), "set_tuning")
    p = upload_params(ctx, params)
    s = torch.from_numpy(states).cuda()
    logits = torch.zeros((B, A), device="cuda")
    values = torch.zeros((B,), device="cuda")
    ctx.prof_enable(True)
    ctx.forward(p, s, logits=logits, values=values)
    torch.cuda.synchronize()
    fams = [name for name, batch, ms, mix in ctx.prof_read(with_mix=True)]
    ctx.prof_enable(False)
    assert "conv_tower" in fams and "conv1_fwd" not in fams and "conv2_fwd" not in fams
    ref = onet.forward(params, states, "NIPS", dtype=np.float64, keep=True)
    check_activations(ctx, ref["cache"], "NIPS", B, scale)
    assert np.abs(logits.cpu().numpy() - ref["logits"]).max() < 1e-4
    assert np.abs(values.cpu().numpy() - ref["v"]).max() < 1e-4
    # the per-layer route (what NIPS ran on before): same values to fp32 summation order
    monkeypatch.setenv("PAAC_TOWER", "0")
    plain = hip_ops.Context(ARCH_ID["NIPS"], A, max_batch=B)
    monkeypatch.delenv("PAAC_TOWER")
    logits0 = torch.zeros((B, A), device="cuda")
    plain.prof_enable(True)
    plain.forward(p, s, logits=logits0)
    torch.cuda.synchronize()
    assert "conv1_fwd" in [name for name, batch, ms, mix in plain.prof_read(with_mix=True)]
    plain.prof_enable(False)
    assert float((logits - logits0).abs().max()) < 2e-5 * max(1.0, float(logits.abs().max()))
    plain.close()
    # managed mode: the acting forward keeps only conv2's output, in the fc kernel's fragment order
    ctx.set_managed_weights(True)
    ctx.pack_weights(p)
    logits2 = torch.zeros((B, A), device="cuda")
    ctx.forward(p, s, logits=logits2)
    torch.cuda.synchronize()
    if B <= 64:
        assert torch.equal(logits, logits2)
    else:       # up to 256 rows the managed forward takes the acting fc kernel, the unmanaged one the split-K GEMM
        assert float((logits - logits2).abs().max()) < 2e-5 * max(1.0, float(logits.abs().max()))
    ctx.close()
