"""Oracle: actor-critic network forward, loss, analytic gradients, clip and
TF-semantics RMSProp -- numpy restatement (TEST INFRASTRUCTURE, see __init__).

Follows (reference file:line, relative to /root/reference):
  networks.py:6-9     flatten keeps NHWC order (feature = (h*W + w)*C + c)
  networks.py:12-21   conv2d: VALID, NHWC x HWIO, + bias, ReLU
  networks.py:49-60   fc: x @ W[in,out] + b (+ ReLU)
  networks.py:84-89   softmax head
  networks.py:100-120 input = cast(u8 -> f32) * (1/255)
  networks.py:138-169 NIPS (16,8,4)->(32,4,2)->fc256 ; Nature (32,8,4)->(64,4,2)->(64,3,1)->fc512
  policy_v_network.py:6-57 heads, log(pi+1e-30), entropy, losses, loss = 5*(actor+critic)
  actor_learner.py:31-34,44-70 RMSProp(decay=alpha, epsilon=e), clip_by_global_norm
Constants TF leaves implicit are taken from pretrained/*/checkpoints/*.meta
(SURVEY.md section 8c): rms slot init 1.0, momentum 0.0, epsilon inside sqrt,
global_norm = sqrt(sum g^2), clip factor = c * min(1/gn, 1/c).

PARITY UNPINNED at the TensorFlow boundary: the reference holds no golden
vectors for these ops.  What it does hold -- the serialized NIPS training graph
pretrained/breakout/checkpoints/-80000000.meta -- pins the STRUCTURE node by node
(tests/test_meta_graph_pin.py: op types, wiring, Conv2D strides/padding/layout,
every constant, the clip formula, ApplyRMSProp inputs, slot and weight
initialisers, checkpoint names and shapes); the restatement's values are
cross-checked against torch float64 autograd in tests/test_oracle_network.py.
"""
import numpy as np

ARCHS = {
    # name: (conv layers (filters, size, stride), fc width)   networks.py:145-149 / :161-167
    "NIPS": ([(16, 8, 4), (32, 4, 2)], 256),
    "NATURE": ([(32, 8, 4), (64, 4, 2), (64, 3, 1)], 512),
}
INPUT_SCALE = np.float32(1.0 / 255.0)   # networks.py:115 (float32 const 0.003921568859...)
LOG_EPS = np.float32(1e-30)             # policy_v_network.py:29
LOSS_SCALING = 5.0                      # networks.py:112
CRITIC_COEF = 0.25                      # policy_v_network.py:53


def arch_key(arch):
    """train.py:63-66: 'NIPS' selects NIPS, anything else selects Nature -- except a name a test registered in ARCHS (a
    user architecture, networks.py:117-120: same layer helpers, other widths)."""
    if arch in ARCHS:
        return arch
    return "NIPS" if arch == "NIPS" else "NATURE"


def layer_dims(arch):
    convs, fc = ARCHS[arch_key(arch)]
    h = w = 84
    c = 4
    out = []
    for (f, k, s) in convs:
        oh = (h - k) // s + 1
        ow = (w - k) // s + 1
        out.append(dict(kh=k, kw=k, cin=c, cout=f, stride=s, ih=h, iw=w, oh=oh, ow=ow))
        h, w, c = oh, ow, f
    return out, h * w * c, fc


def param_shapes(arch, num_actions):
    """TF variable creation order == checkpoint/.meta order."""
    convs, flat, fc = layer_dims(arch)
    shapes = []
    for i, L in enumerate(convs):
        shapes.append(("conv%d_weights" % (i + 1), (L["kh"], L["kw"], L["cin"], L["cout"])))
        shapes.append(("conv%d_biases" % (i + 1), (L["cout"],)))
    n = len(convs) + 1
    shapes.append(("fc%d_weights" % n, (flat, fc)))
    shapes.append(("fc%d_biases" % n, (fc,)))
    shapes.append(("actor_output_weights", (fc, num_actions)))
    shapes.append(("actor_output_biases", (num_actions,)))
    shapes.append(("critic_output_weights", (fc, 1)))
    shapes.append(("critic_output_biases", (1,)))
    return shapes


def num_params(arch, num_actions):
    return int(sum(int(np.prod(s)) for _, s in param_shapes(arch, num_actions)))


def init_params(arch, num_actions, rng, dtype=np.float32):
    """'torch' init, networks.py:24-46,63-81: U(-d, d) with d = 1/sqrt(fan_in) for W and b."""
    params = {}
    for name, shape in param_shapes(arch, num_actions):
        if name.endswith("weights"):
            fan_in = int(np.prod(shape[:-1]))
            last_fan_in = fan_in
        else:
            fan_in = last_fan_in
        d = 1.0 / np.sqrt(fan_in)
        params[name] = rng.uniform(-d, d, size=shape).astype(dtype)
    return params


def _im2col(x, kh, kw, s):
    """x [B,H,W,C] -> cols [B*OH*OW, kh*kw*C] with K order (kh, kw, c) = HWIO flattening."""
    B, H, W, C = x.shape
    win = np.lib.stride_tricks.sliding_window_view(x, (kh, kw), axis=(1, 2))  # [B,H-kh+1,W-kw+1,C,kh,kw]
    win = win[:, ::s, ::s]
    OH, OW = win.shape[1], win.shape[2]
    cols = np.ascontiguousarray(win.transpose(0, 1, 2, 4, 5, 3)).reshape(B * OH * OW, kh * kw * C)
    return cols, OH, OW


def _col2im(dcols, B, H, W, C, kh, kw, s, OH, OW):
    dx = np.zeros((B, H, W, C), dtype=dcols.dtype)
    d6 = dcols.reshape(B, OH, OW, kh, kw, C)
    for i in range(kh):
        for j in range(kw):
            dx[:, i:i + s * OH:s, j:j + s * OW:s, :] += d6[:, :, :, i, j, :]
    return dx


def forward(params, states_u8, arch, dtype=np.float64, keep=False):
    """Returns dict(logits, pi, v[, cache]).  Arithmetic in `dtype` except the
    input scaling, which is done in float32 exactly like the graph
    (cast u8->f32, multiply by float32(1/255)) and then promoted."""
    convs, flat, fc = layer_dims(arch)
    x = (states_u8.astype(np.float32) * INPUT_SCALE).astype(dtype)
    cache = {"x0": x}
    for i, L in enumerate(convs):
        w = params["conv%d_weights" % (i + 1)].astype(dtype)
        b = params["conv%d_biases" % (i + 1)].astype(dtype)
        cols, OH, OW = _im2col(x, L["kh"], L["kw"], L["stride"])
        z = cols @ w.reshape(-1, L["cout"]) + b
        a = np.maximum(z, 0)
        if keep:
            cache["cols%d" % (i + 1)] = cols
        x = a.reshape(x.shape[0], OH, OW, L["cout"])
        cache["a%d" % (i + 1)] = x
    n = len(convs) + 1
    xf = x.reshape(x.shape[0], -1)                       # networks.py:6-9 (HWC order)
    w = params["fc%d_weights" % n].astype(dtype)
    b = params["fc%d_biases" % n].astype(dtype)
    h = np.maximum(xf @ w + b, 0)
    cache["xf"] = xf
    cache["h"] = h
    logits = h @ params["actor_output_weights"].astype(dtype) + params["actor_output_biases"].astype(dtype)
    m = logits.max(axis=1, keepdims=True)                # tf.nn.softmax is max-subtracted
    e = np.exp(logits - m)
    pi = e / e.sum(axis=1, keepdims=True)
    v = (h @ params["critic_output_weights"].astype(dtype) + params["critic_output_biases"].astype(dtype)).reshape(-1)
    out = {"logits": logits, "pi": pi, "v": v}
    if keep:
        out["cache"] = cache
    else:
        out["h"] = h
    return out


def loss_terms(pi, v, onehot, y, adv, beta, dtype=np.float64):
    """policy_v_network.py:29-57."""
    eps = dtype(LOG_EPS)
    lp = np.log(pi + eps)
    ent = -(pi * lp).sum(axis=1)
    logp = (lp * onehot).sum(axis=1)
    actor = np.mean(-(logp * adv + beta * ent))
    critic = np.mean(CRITIC_COEF * (y - v) ** 2)
    loss = LOSS_SCALING * (actor + critic)
    return dict(loss=loss, actor=actor, critic=critic, entropy=ent, logp=logp)


def head_grads(pi, v, onehot, y, adv, beta, dtype=np.float64):
    """d loss / d logits, d loss / d v  (derived, verified vs autograd)."""
    B = pi.shape[0]
    eps = dtype(LOG_EPS)
    s = LOSS_SCALING / B
    lp = np.log(pi + eps)
    dv = s * 2.0 * CRITIC_COEF * (v - y)
    g = -(adv[:, None] * onehot / (pi + eps) - beta * (lp + pi / (pi + eps)))
    dlogits = s * pi * (g - (g * pi).sum(axis=1, keepdims=True))
    return dlogits, dv


def loss_and_grads(params, states_u8, onehot, y, adv, beta, arch, dtype=np.float64, relu_masks=None):
    """Full forward + loss + backward.  Returns (loss dict, grads dict in param order).
    relu_masks (optional): {"a1".."a3", "h"} boolean arrays overriding relu'(.) in the backward pass -- lets a
    test take the masks from the implementation under test so that pre-activations within rounding of 0
    (whose sign legitimately differs between summation orders) do not mask real backward errors."""
    convs, flat, fc = layer_dims(arch)
    fw = forward(params, states_u8, arch, dtype=dtype, keep=True)
    cache = fw["cache"]
    onehot = onehot.astype(dtype)
    y = y.astype(dtype)
    adv = adv.astype(dtype)
    L = loss_terms(fw["pi"], fw["v"], onehot, y, adv, dtype(beta), dtype)
    dlogits, dv = head_grads(fw["pi"], fw["v"], onehot, y, adv, dtype(beta), dtype)
    grads = {}
    h = cache["h"]
    grads["actor_output_weights"] = h.T @ dlogits
    grads["actor_output_biases"] = dlogits.sum(axis=0)
    grads["critic_output_weights"] = h.T @ dv[:, None]
    grads["critic_output_biases"] = dv.sum(keepdims=True)
    dh = dlogits @ params["actor_output_weights"].astype(dtype).T + dv[:, None] @ params["critic_output_weights"].astype(dtype).T
    dh = dh * ((h > 0) if relu_masks is None else relu_masks["h"].reshape(h.shape))
    n = len(convs) + 1
    grads["fc%d_weights" % n] = cache["xf"].T @ dh
    grads["fc%d_biases" % n] = dh.sum(axis=0)
    dx = dh @ params["fc%d_weights" % n].astype(dtype).T
    for i in reversed(range(len(convs))):
        Lc = convs[i]
        a = cache["a%d" % (i + 1)]
        B = a.shape[0]
        m = (a > 0) if relu_masks is None else relu_masks["a%d" % (i + 1)].reshape(a.shape)
        dz = (dx.reshape(a.shape) * m).reshape(-1, Lc["cout"])
        cols = cache["cols%d" % (i + 1)]
        grads["conv%d_weights" % (i + 1)] = (cols.T @ dz).reshape(Lc["kh"], Lc["kw"], Lc["cin"], Lc["cout"])
        grads["conv%d_biases" % (i + 1)] = dz.sum(axis=0)
        if i > 0:
            w = params["conv%d_weights" % (i + 1)].astype(dtype).reshape(-1, Lc["cout"])
            dcols = dz @ w.T
            dx = _col2im(dcols, B, Lc["ih"], Lc["iw"], Lc["cin"], Lc["kh"], Lc["kw"], Lc["stride"], Lc["oh"], Lc["ow"])
    ordered = {name: grads[name].reshape(shape) for name, shape in param_shapes(arch, onehot.shape[1])}
    L["pi"] = fw["pi"]
    L["v"] = fw["v"]
    L["logits"] = fw["logits"]
    return L, ordered


def global_norm(grads):
    """tf.global_norm: sqrt(sum_i 2*L2Loss(g_i)) = sqrt(sum g^2)."""
    return np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))


def clip_by_global_norm(grads, clip_norm, mode="global"):
    """actor_learner.py:51-59.  'ignore' leaves grads unscaled; 'global' scales by
    clip_norm * min(1/gn, 1/clip_norm)  (.meta: clip_by_global_norm/mul)."""
    gn = global_norm(grads)
    if mode == "ignore":
        return dict(grads), gn
    if mode != "global":
        raise ValueError("clip_norm_type %r: the reference's 'local' branch is broken "
                         "(actor_learner.py:62-63); only 'global' and 'ignore' are defined" % mode)
    scale = clip_norm * min(1.0 / gn, 1.0 / clip_norm) if gn > 0 else 1.0
    return {k: g * scale for k, g in grads.items()}, gn


def rmsprop_init(params):
    """Slot init pinned by .meta: rms = 1.0, momentum = 0."""
    ms = {k: np.ones_like(v) for k, v in params.items()}
    mom = {k: np.zeros_like(v) for k, v in params.items()}
    return ms, mom


def rmsprop_step(params, grads, ms, mom, lr, decay=0.99, momentum=0.0, eps=0.1):
    """TF-1.0 ApplyRMSProp: ms += (g^2 - ms)(1-decay); mom = momentum*mom + lr*g/sqrt(ms+eps); var -= mom."""
    for k in params:
        g = grads[k].astype(params[k].dtype)
        dt = params[k].dtype.type
        ms[k] = ms[k] + (g * g - ms[k]) * dt(1.0 - decay)
        mom[k] = dt(momentum) * mom[k] + dt(lr) * g / np.sqrt(ms[k] + dt(eps))
        params[k] = params[k] - mom[k]
    return params, ms, mom


def flatten_params(d, arch, num_actions):
    return np.concatenate([np.asarray(d[name]).reshape(-1) for name, _ in param_shapes(arch, num_actions)])
