"""Structural pin of the oracle's TensorFlow half against the one artefact the reference holds for it: the
serialized NIPS training graph pretrained/breakout/checkpoints/-80000000.meta (TF 1.0.1) and the checkpoint .index
next to it.  The protobuf / table wire formats are walked by tests/tfproto.py (nothing is executed, TensorFlow is
not needed) and every assertion is made BY NODE: op type, input wiring, attribute or constant value -- so each
constant and formula oracle/network.py hard-codes is tied to the graph node that carries it.

This is the most /root/reference can give the forward / loss / gradient / clip / RMSProp arithmetic; the VALUES those
ops produce remain "parity unpinned" (no TensorFlow here, no golden vectors upstream).  Skipped where /root/reference
is not mounted (the GPU box)."""
import os

import numpy as np
import pytest

import tfproto
from oracle import network as onet

CKPT = "/root/reference/pretrained/breakout/checkpoints/-80000000"
pytestmark = pytest.mark.skipif(not os.path.exists(CKPT + ".meta"), reason="reference not mounted")


@pytest.fixture(scope="module")
def nodes():
    return tfproto.graph_nodes(CKPT + ".meta")


def const(nodes, name):
    n = tfproto.producer(nodes, name)
    assert n.op == "Const", (name, n.op)
    vals, dims = n.const_floats()
    return vals, dims


def scalar(nodes, name):
    vals, dims = const(nodes, name)
    assert dims == [] and len(vals) == 1
    return np.float32(vals[0])


def test_input_scaling_node(nodes):
    """networks.py:115: input = scalar_mul(1/255, cast(uint8 -> float32)); the multiplier is a float32 Const."""
    mul = nodes["local_learning/mul"]
    assert mul.op == "Mul"
    ops = sorted(tfproto.producer(nodes, i).op for i in mul.inputs)
    assert ops == ["Cast", "Const"]
    assert tfproto.producer(nodes, "local_learning/Cast").inputs == ["local_learning/input"]
    assert nodes["local_learning/input"].op == "Placeholder"
    k = scalar(nodes, "local_learning/scalar")
    assert k == onet.INPUT_SCALE and k.tobytes() == np.float32(1.0 / 255.0).tobytes()
    # the scaled input is what conv1 consumes
    assert nodes["local_learning_1/conv1_convs"].inputs[0] == "local_learning/mul"


def test_conv_nodes(nodes):
    """networks.py:12-21,145-147: Conv2D VALID NHWC, strides [1,4,4,1] / [1,2,2,1], then Add(bias) then Relu."""
    convs, _, _ = onet.layer_dims("NIPS")
    chain_in = "local_learning/mul"
    for i, L in enumerate(convs):
        conv = nodes["local_learning_1/conv%d_convs" % (i + 1)]
        assert conv.op == "Conv2D"
        assert conv.inputs[0] == chain_in
        w = tfproto.through_identity(nodes, conv.inputs[1])
        assert w.op == "VariableV2" and w.name == "local_learning_1/conv%d_weights" % (i + 1)
        assert conv.attr_s("padding") == "VALID" and conv.attr_s("data_format") == "NHWC"
        assert conv.attr_ints("strides") == [1, L["stride"], L["stride"], 1]
        add = nodes["local_learning_1/Add" + ("" if i == 0 else "_%d" % i)]
        assert add.op == "Add" and add.inputs[0] == conv.name
        assert tfproto.through_identity(nodes, add.inputs[1]).name == "local_learning_1/conv%d_biases" % (i + 1)
        relu = nodes["local_learning_1/conv%d_activations" % (i + 1)]
        assert relu.op == "Relu" and relu.inputs == [add.name]
        chain_in = relu.name
    # flatten is a plain Reshape of the NHWC activations (networks.py:6-9) feeding MatMul without transposes
    flat = nodes["local_learning_1/_flattened"]
    assert flat.op == "Reshape" and flat.inputs[0] == chain_in
    mm = nodes["local_learning_1/MatMul"]
    assert mm.op == "MatMul" and mm.inputs[0] == flat.name
    for key in ("transpose_a", "transpose_b"):
        assert tfproto.first(tfproto.fields(mm.attr[key]), 5, 0) == 0
    assert nodes["local_learning_1/fc3_out"].op == "Add" and nodes["local_learning_1/fc3_out"].inputs[0] == mm.name
    assert nodes["local_learning_1/fc3_relu"].inputs == ["local_learning_1/fc3_out"]


def test_heads_and_loss_nodes(nodes):
    """policy_v_network.py:24-57 as wired in the graph: softmax head, linear critic, log(pi + 1e-30), entropy,
    actor / critic terms, loss scale."""
    h = "local_learning_1/fc3_relu"
    assert nodes["local_learning_2/MatMul"].inputs[0] == h and nodes["local_learning_2/MatMul_1"].inputs[0] == h
    pi = nodes["local_learning_2/actor_output_policy"]
    assert pi.op == "Softmax" and pi.inputs == ["local_learning_2/Add"]
    assert nodes["local_learning_2/Add"].inputs[0] == "local_learning_2/MatMul"
    # log policy = Log(Add(pi, 1e-30))
    lp = nodes["local_learning_2/actor_output_log_policy"]
    assert lp.op == "Log"
    add = tfproto.producer(nodes, lp.inputs[0])
    assert add.op == "Add" and add.inputs[0] == pi.name
    eps = scalar(nodes, add.inputs[1])
    assert eps == onet.LOG_EPS and eps.tobytes() == np.float32(1e-30).tobytes()
    # entropy = Sum(-1 * (pi * log_pi))
    ent = nodes["local_learning_2/Sum"]
    m1 = tfproto.producer(nodes, ent.inputs[0])
    assert ent.op == "Sum" and m1.op == "Mul" and scalar(nodes, m1.inputs[0]) == np.float32(-1.0)
    assert sorted(tfproto.producer(nodes, m1.inputs[1]).inputs) == sorted([pi.name, lp.name])
    # actor = Mean(-1 * (Sum(log_pi * selected_action) * advantage + beta * entropy)), beta = 0.02
    mean = nodes["local_learning_2/Mean"]
    neg = tfproto.producer(nodes, mean.inputs[0])
    assert mean.op == "Mean" and neg.op == "Mul" and scalar(nodes, neg.inputs[0]) == np.float32(-1.0)
    inner = tfproto.producer(nodes, neg.inputs[1])
    assert inner.op == "Add"
    logp_adv, beta_ent = (tfproto.producer(nodes, i) for i in inner.inputs)
    assert logp_adv.op == "Mul" and logp_adv.inputs[1] == "local_learning_2/advantage"
    logp = tfproto.producer(nodes, logp_adv.inputs[0])
    assert logp.op == "Sum" and sorted(tfproto.producer(nodes, logp.inputs[0]).inputs) == sorted(
        [lp.name, "local_learning/selected_action"])
    assert beta_ent.op == "Mul" and beta_ent.inputs[1] == ent.name
    assert scalar(nodes, beta_ent.inputs[0]) == np.float32(0.02)             # train.py:87 default --entropy
    # critic = 0.25 * Mean((target - v)^2); v = Reshape(critic_output_out)
    sub = nodes["local_learning_2/Sub"]
    assert sub.op == "Sub" and sub.inputs == ["local_learning_2/target", "local_learning_2/Reshape"]
    assert nodes["local_learning_2/Reshape"].inputs[0] == "local_learning_2/critic_output_out"
    powr = nodes["local_learning_2/Pow"]
    assert powr.inputs[0] == sub.name and scalar(nodes, powr.inputs[1]) == np.float32(2.0)
    crit = nodes["local_learning_2/mul"]
    assert crit.op == "Mul" and scalar(nodes, crit.inputs[0]) == np.float32(onet.CRITIC_COEF)
    assert tfproto.producer(nodes, crit.inputs[1]).op == "Mean"
    assert tfproto.producer(nodes, crit.inputs[1]).inputs[0] == powr.name
    # loss = 5 * (actor + critic)
    total = nodes["local_learning_2/mul_1"]
    assert scalar(nodes, total.inputs[0]) == np.float32(onet.LOSS_SCALING)
    assert sorted(tfproto.producer(nodes, total.inputs[1]).inputs) == sorted([mean.name, crit.name])
    # and that node is what the gradient graph differentiates
    assert any(n.name.startswith("gradients/local_learning_2/mul_1_grad") for n in nodes.values())


def test_global_norm_and_clip_nodes(nodes):
    """actor_learner.py:56-59 -> global_norm = sqrt(2 * sum_i L2Loss(g_i)) over the 10 gradients in variable order;
    factor = clip * min(1/global_norm, 1/clip), clip = 3.0; every gradient is multiplied by that one factor."""
    stack = nodes["global_norm/stack"]
    l2 = [tfproto.producer(nodes, i) for i in stack.inputs]
    assert len(l2) == 10 and all(n.op == "L2Loss" for n in l2)
    s = nodes["global_norm/Sum"]
    assert s.inputs[0] == stack.name
    mul = nodes["global_norm/mul"]
    assert mul.inputs[0] == s.name and scalar(nodes, mul.inputs[1]) == np.float32(2.0)
    gn = nodes["global_norm/global_norm"]
    assert gn.op == "Sqrt" and gn.inputs == [mul.name]
    a, b = nodes["clip_by_global_norm/truediv"], nodes["clip_by_global_norm/truediv_1"]
    assert a.op == b.op == "RealDiv"
    assert scalar(nodes, a.inputs[0]) == np.float32(1.0) and a.inputs[1] == gn.name            # 1 / global_norm
    assert scalar(nodes, b.inputs[0]) == np.float32(1.0) and scalar(nodes, b.inputs[1]) == np.float32(3.0)   # 1 / clip
    mn = nodes["clip_by_global_norm/Minimum"]
    assert mn.op == "Minimum" and sorted(mn.inputs) == sorted([a.name, b.name])
    fac = nodes["clip_by_global_norm/mul"]
    assert scalar(nodes, fac.inputs[0]) == np.float32(3.0) and fac.inputs[1] == mn.name
    for i in range(10):
        m = nodes["clip_by_global_norm/mul_%d" % (i + 1)]
        assert m.op == "Mul" and m.inputs[1] == fac.name and m.inputs[0] == l2[i].inputs[0]   # same gradient, same order
    # the oracle's formula on the same numbers
    g = {"a": np.array([3.0, 4.0]), "b": np.array([12.0])}
    clipped, norm = onet.clip_by_global_norm(g, 3.0)
    assert norm == 13.0 and np.allclose(clipped["b"], 12.0 * 3.0 * min(1.0 / 13.0, 1.0 / 3.0))


def test_rmsprop_nodes_and_slot_initialisers(nodes):
    """actor_learner.py:31-34,70: one ApplyRMSProp per variable, inputs (var, ms, mom, lr, decay, momentum, epsilon,
    clipped gradient); decay 0.99, momentum 0.0, epsilon 0.1; ms slot initialised to ones, mom slot to zeros; the
    variable order is the checkpoint order the flat parameter buffer uses."""
    applies = [n for n in nodes.values() if n.op == "ApplyRMSProp"]
    names = [n.inputs[0] for n in applies]
    want = ["local_learning_%d/%s" % (1 if not k.startswith(("actor", "critic")) else 2, k)
            for k, _ in onet.param_shapes("NIPS", 4)]
    assert names == want
    for i, n in enumerate(applies):
        var, ms, mom, lr, decay, momentum, eps, grad = n.inputs
        assert ms == var + "/OptimizerVariables" and mom == var + "/OptimizerVariables_1"
        assert nodes[lr].op == "Placeholder"
        assert scalar(nodes, decay) == np.float32(0.99)
        assert scalar(nodes, momentum) == np.float32(0.0)
        assert scalar(nodes, eps) == np.float32(0.1)
        assert grad == "clip_by_global_norm/clip_by_global_norm/_%d" % i
        assert tfproto.through_identity(nodes, grad).name == "clip_by_global_norm/mul_%d" % (i + 1)
        ms_init, _ = const(nodes, nodes[ms + "/Assign"].inputs[1])
        mom_init, _ = const(nodes, nodes[mom + "/Assign"].inputs[1])
        assert np.all(ms_init == 1.0) and np.all(mom_init == 0.0)
    p = {"w": np.zeros(3, dtype=np.float32)}
    ms0, mom0 = onet.rmsprop_init(p)
    assert np.all(ms0["w"] == 1.0) and np.all(mom0["w"] == 0.0)


def test_initialiser_ranges(nodes):
    """networks.py:24-46,63-81 ('torch' init): every tensor ~ U(-d, d) with d = 1/sqrt(fan_in of ITS LAYER) -- a bias
    uses its layer's weight fan-in.  The graph holds min/max constants per variable."""
    shapes = dict(onet.param_shapes("NIPS", 4))
    fan_in = None
    for name, shape in onet.param_shapes("NIPS", 4):
        if name.endswith("weights"):
            fan_in = int(np.prod(shape[:-1]))
        scope = 2 if name.startswith(("actor", "critic")) else 1
        init = tfproto.producer(nodes, nodes["local_learning_%d/%s/Assign" % (scope, name)].inputs[1])
        assert init.op == "Add"                                     # random_uniform = U[0,1) * (max - min) + min
        lo = scalar(nodes, init.inputs[1])
        sub = tfproto.producer(nodes, tfproto.producer(nodes, init.inputs[0]).inputs[1])
        hi = scalar(nodes, sub.inputs[0])
        d = np.float32(1.0 / np.sqrt(fan_in))
        assert abs(hi - d) <= 1e-7 * d and abs(lo + d) <= 1e-7 * d, (name, lo, hi, d)
        # the oracle's initialiser stays inside the same range
        drawn = onet.init_params("NIPS", 4, np.random.RandomState(1))[name]
        assert drawn.shape == shapes[name] and np.abs(drawn).max() <= d and np.abs(drawn).max() > 0.9 * d or drawn.size < 8


def test_checkpoint_index_names_and_shapes():
    """The checkpoint .index: 10 variables + two optimizer slots each, the scopes of networks.py:111,144 /
    policy_v_network.py:16 ('local_learning_1' trunk, 'local_learning_2' heads), shapes = oracle.param_shapes."""
    entries = tfproto.bundle_entries(CKPT + ".index")
    want = {}
    for name, shape in onet.param_shapes("NIPS", 4):
        key = "local_learning_%d/%s" % (2 if name.startswith(("actor", "critic")) else 1, name)
        for suffix in ("", "/OptimizerVariables", "/OptimizerVariables_1"):
            want[key + suffix] = tuple(shape)
    assert entries == want
