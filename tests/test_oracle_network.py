"""Oracle network restatement vs torch float64 autograd (the TF boundary is 'parity unpinned':
this is the cross-check DESIGN.md names) + constants frozen in the reference's .meta graphs."""
import numpy as np
import pytest
import torch

from oracle import network as onet


def torch_loss(params, states, onehot, y, adv, beta, arch):
    convs, flat, fc = onet.layer_dims(arch)
    P = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    x = torch.tensor(states.astype(np.float32) * onet.INPUT_SCALE, dtype=torch.float64).permute(0, 3, 1, 2)
    for i, L in enumerate(convs):
        w = P["conv%d_weights" % (i + 1)].permute(3, 2, 0, 1)          # HWIO -> OIHW
        x = torch.relu(torch.nn.functional.conv2d(x, w, P["conv%d_biases" % (i + 1)], stride=L["stride"]))
    n = len(convs) + 1
    xf = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                 # flatten in HWC order
    h = torch.relu(xf @ P["fc%d_weights" % n] + P["fc%d_biases" % n])
    logits = h @ P["actor_output_weights"] + P["actor_output_biases"]
    pi = torch.softmax(logits, dim=1)
    v = (h @ P["critic_output_weights"] + P["critic_output_biases"]).reshape(-1)
    lp = torch.log(pi + 1e-30)
    ent = -(pi * lp).sum(1)
    oh = torch.tensor(onehot, dtype=torch.float64)
    logp = (lp * oh).sum(1)
    actor = (-(logp * torch.tensor(adv) + beta * ent)).mean()
    critic = (0.25 * (torch.tensor(y) - v) ** 2).mean()
    loss = 5.0 * (actor + critic)
    loss.backward()
    return loss.item(), {k: p.grad.numpy() for k, p in P.items()}, pi.detach().numpy(), v.detach().numpy()


@pytest.mark.parametrize("arch,A", [("NIPS", 6), ("NATURE", 4), ("NATURE", 18)])
def test_grads_match_autograd(arch, A):
    rs = np.random.RandomState(0)
    B = 6
    params = onet.init_params(arch, A, rs, dtype=np.float64)
    states = rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)
    idx = rs.randint(0, A, B)
    onehot = np.eye(A)[idx]
    y = rs.randn(B)
    adv = rs.randn(B)
    L, grads = onet.loss_and_grads(params, states, onehot, y, adv, 0.02, arch, dtype=np.float64)
    tl, tg, tpi, tv = torch_loss(params, states, onehot, y, adv, 0.02, arch)
    assert abs(L["loss"] - tl) < 1e-12
    assert np.abs(L["pi"] - tpi).max() < 1e-14
    assert np.abs(L["v"] - tv).max() < 1e-13
    for k in grads:
        scale = max(np.abs(tg[k]).max(), 1e-12)
        assert np.abs(grads[k] - tg[k]).max() / scale < 1e-10, k


def test_param_counts():
    # Nature(A=4) 1,686,693 ; NIPS(A=6) 677,943
    assert onet.num_params("NATURE", 4) == 1686693
    assert onet.num_params("NIPS", 6) == 677943
    names = [n for n, _ in onet.param_shapes("NIPS", 4)]
    assert names == ["conv1_weights", "conv1_biases", "conv2_weights", "conv2_biases", "fc3_weights", "fc3_biases",
                     "actor_output_weights", "actor_output_biases", "critic_output_weights", "critic_output_biases"]


def test_clip_and_rmsprop_semantics():
    g = {"a": np.array([3.0, 4.0]), "b": np.array([12.0])}       # norm 13
    c, gn = onet.clip_by_global_norm(g, 3.0)
    assert abs(gn - 13.0) < 1e-12
    assert np.allclose(c["a"], np.array([3.0, 4.0]) * 3.0 / 13.0)
    small = {"a": np.array([0.3, 0.4])}
    c2, gn2 = onet.clip_by_global_norm(small, 3.0)                 # below the clip: factor = 3 * (1/3) = 1
    assert np.allclose(c2["a"], small["a"])
    c3, _ = onet.clip_by_global_norm(g, 3.0, "ignore")
    assert np.array_equal(c3["a"], g["a"])
    with pytest.raises(ValueError):
        onet.clip_by_global_norm(g, 3.0, "local")
    p = {"w": np.array([1.0], dtype=np.float32)}
    ms, mom = onet.rmsprop_init(p)
    assert ms["w"][0] == 1.0 and mom["w"][0] == 0.0                # slot inits pinned by .meta
    p, ms, mom = onet.rmsprop_step(p, {"w": np.array([2.0], dtype=np.float32)}, ms, mom, lr=0.1)
    ms_e = 1.0 + (4.0 - 1.0) * 0.01
    assert abs(ms["w"][0] - ms_e) < 1e-6
    assert abs(p["w"][0] - (1.0 - 0.1 * 2.0 / np.sqrt(ms_e + 0.1))) < 1e-6
