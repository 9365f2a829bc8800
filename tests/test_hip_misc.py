"""-m gpu: returns / lr / samplers / preprocessing / synthetic envs vs the oracle and the golden vectors
(bit-exact for bytes and action indices), through the C-ABI."""
import glob
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import preprocess as opre
from oracle import rollout as oroll
from oracle import sampler as osamp

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "rollout_*.npz")))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("T,N", [(5, 32), (20, 128), (5, 1), (1, 7)])
def test_nstep_returns_bit_exact(T, N):
    from paac_amd import hip_ops
    rs = np.random.RandomState(T * 100 + N)
    v_boot = rs.randn(N).astype(np.float32)
    rewards = rs.choice([-1.0, 0.0, 1.0], size=(T, N)).astype(np.float32)
    masks = (rs.rand(T, N) > 0.2).astype(np.float32)
    values = rs.randn(T, N).astype(np.float32)
    y = torch.zeros(T * N, device="cuda")
    adv = torch.zeros(T * N, device="cuda")
    hip_ops.nstep_returns(dev(v_boot), dev(rewards), dev(masks), dev(values), 0.99, y, adv)
    ye, ae = oroll.nstep_returns(v_boot, rewards.astype(np.float64), masks.astype(np.float64),
                                 values.astype(np.float64), 0.99)
    assert np.array_equal(y.cpu().numpy(), ye.reshape(-1).astype(np.float32))
    assert np.array_equal(adv.cpu().numpy(), ae.reshape(-1).astype(np.float32))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_returns_and_sampler(path):
    """Reference-captured vectors: pi -> actions (MT19937 stream incl. final position), rewards/masks/values -> y, adv."""
    from paac_amd import hip_ops
    g = np.load(path)
    N, T, A = int(g["N"]), int(g["T"]), int(g["A"])
    rs = np.random.RandomState(int(g["seed"]))
    state = hip_ops.mt_state_from_numpy(rs.get_state(), "cuda")
    scratch = hip_ops.sample_mt_scratch(N, A, "cuda")
    for c in range(int(g["cycles"])):
        acts = torch.zeros((T, N), dtype=torch.int32, device="cuda")
        for t in range(T):
            hip_ops.sample_mt(dev(g["pi"][c, t]), state, scratch, acts[t])
        want = np.argmax(g["actions"][c], axis=1).reshape(T, N)
        assert np.array_equal(acts.cpu().numpy(), want), "cycle %d sampled actions differ from the reference" % c
    st = hip_ops.mt_state_to_numpy(state)
    import hashlib
    assert st[2] == int(g["mt_pos"])
    assert hashlib.sha256(np.asarray(st[1], dtype=np.uint32).tobytes()).hexdigest() == str(g["mt_key_sha256"])
    # returns: reconstruct rewards/masks from the fixture's y/adv is circular, so recompute them from the oracle replay
    from test_oracle_golden import replay_oracle
    ro, cycles, _ = replay_oracle(g)
    for c, cyc in enumerate(cycles):
        y = torch.zeros(T * N, device="cuda")
        adv = torch.zeros(T * N, device="cuda")
        hip_ops.nstep_returns(dev(cyc["v_boot"].astype(np.float32)), dev(cyc["rewards"].astype(np.float32)),
                              dev(cyc["masks"].astype(np.float32)), dev(cyc["values"].astype(np.float32)),
                              float(g["gamma"]), y, adv)
        assert np.array_equal(y.cpu().numpy(), g["y"][c].astype(np.float32))
        assert np.array_equal(adv.cpu().numpy(), g["adv"][c].astype(np.float32))


@pytest.mark.parametrize("N,A", [(32, 4), (8, 6), (16, 18), (256, 4), (1024, 18), (3, 2), (5, 32),
                                 (64, 4), (65, 4), (100, 6), (128, 18), (255, 3), (33, 9)])
def test_sampler_mt_matches_numpy(N, A):
    from paac_amd import hip_ops
    gen = np.random.RandomState(N * 31 + A)
    rs = np.random.RandomState(1234)
    rs.random_sample(N % 7)            # start from an arbitrary stream position
    state = hip_ops.mt_state_from_numpy(rs.get_state(), "cuda")
    scratch = hip_ops.sample_mt_scratch(N, A, "cuda")
    acts = torch.zeros(N, dtype=torch.int32, device="cuda")
    for it in range(12):
        logits = gen.randn(N, A) * 2.0
        p = np.exp(logits - logits.max(1, keepdims=True))
        p = (p / p.sum(1, keepdims=True)).astype(np.float32)
        p = np.maximum(p, 1e-6).astype(np.float32)
        p = (p / p.sum(1, keepdims=True)).astype(np.float32)
        hip_ops.sample_mt(dev(p), state, scratch, acts)
        want = osamp.sample_numpy_reference(p, rs)
        assert np.array_equal(acts.cpu().numpy(), np.asarray(want, dtype=np.int32)), "iteration %d" % it
        st = hip_ops.mt_state_to_numpy(state)
        ref = rs.get_state()
        assert st[2] == ref[2] and np.array_equal(st[1], ref[1]), "MT19937 stream diverged at iteration %d" % it


def test_sampler_mt_edge_probabilities():
    from paac_amd import hip_ops
    # one-hot-ish rows and an exact-zero category (numpy draws nothing for p == 0)
    eps = np.float32(np.finfo(np.float32).epsneg)
    p = np.array([[eps, 1.0 - 1e-6, 1e-6], [0.5, 0.5, 0.0], [1e-6, 1e-6, 1.0 - 2e-6], [1 / 3, 1 / 3, 1 / 3]], dtype=np.float32)
    p[0, 0] = eps            # p - epsneg == 0 exactly
    rs_a = np.random.RandomState(5)
    rs_b = np.random.RandomState(5)
    state = hip_ops.mt_state_from_numpy(rs_a.get_state(), "cuda")
    scratch = hip_ops.sample_mt_scratch(4, 3, "cuda")
    acts = torch.zeros(4, dtype=torch.int32, device="cuda")
    for _ in range(20):
        hip_ops.sample_mt(dev(p), state, scratch, acts)
        want, _ = osamp.sample_mt_restated(p, rs_b)
        assert np.array_equal(acts.cpu().numpy(), np.asarray(want, dtype=np.int32))
    assert hip_ops.mt_state_to_numpy(state)[2] == rs_b.get_state()[2]


@pytest.mark.parametrize("N,A", [(32, 4), (100, 18)])
def test_sampler_philox_matches_spec(N, A):
    from paac_amd import hip_ops
    gen = np.random.RandomState(7)
    logits = gen.randn(N, A)
    p = np.exp(logits)
    p = (p / p.sum(1, keepdims=True)).astype(np.float32)
    acts = torch.zeros(N, dtype=torch.int32, device="cuda")
    base = torch.tensor([123456789012], dtype=torch.int64, device="cuda")
    for off in (0, 3):
        hip_ops.sample_philox(dev(p), 42, base, off, 5, acts)
        want = osamp.sample_philox(p, 42, 123456789012 + off, env_offset=5)
        assert np.array_equal(acts.cpu().numpy(), want)
    hip_ops.counter_add(base, 7)
    assert int(base.item()) == 123456789012 + 7


def test_lr_step():
    from paac_amd import hip_ops
    gs = torch.tensor([0], dtype=torch.int64, device="cuda")
    lr = torch.zeros(1, device="cuda")
    step = 0
    for inc in (160, 160, 79999680, 5):
        hip_ops.lr_step(gs, inc, 0.0224, 80000000, lr)
        step += inc
        assert int(gs.item()) == step
        assert lr.item() == np.float32(oroll.get_lr(step, 0.0224, 80000000))


@pytest.mark.parametrize("N", [5, 256])          # 256: BASELINE configs[2] ("large-batch preprocess/HBM path")
@pytest.mark.parametrize("rgb", [False, True])
def test_preprocess_stack_bit_exact(rgb, N):
    """atari_emulator.py:69-75 (max of the two screens, nearest resize) + environment.py:58-75 (history push) for N
    environments in one launch, push / reset masks on: every byte against the oracle's PIL-pinned restatement."""
    from paac_amd import hip_ops
    rs = np.random.RandomState(11 + N)
    shape = (N, 2, 210, 160, 3) if rgb else (N, 2, 210, 160)
    raw = rs.randint(0, 256, shape).astype(np.uint8)
    stack = rs.randint(0, 256, (N, 84, 84, 4)).astype(np.uint8)
    push = (rs.rand(N) < 0.8).astype(np.uint8)
    reset = (rs.rand(N) < 0.3).astype(np.uint8)
    push[:5] = [1, 1, 0, 1, 1]
    reset[:5] = [0, 1, 0, 0, 0]
    out = torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device="cuda")
    hip_ops.preprocess_stack(dev(raw), dev(stack), out, dev(push), dev(reset))
    want = np.empty_like(stack)
    for e in range(N):
        fr = opre.rgb_to_gray(raw[e]) if rgb else raw[e]
        plane = opre.max_resize(fr)
        base = np.zeros_like(stack[e]) if reset[e] else stack[e]
        want[e] = opre.push_observation(base, plane) if push[e] else stack[e]
    assert np.array_equal(out.cpu().numpy(), want)
    # in place, no masks
    s2 = dev(stack)
    hip_ops.preprocess_stack(dev(raw), s2, s2)
    want2 = np.stack([opre.push_observation(stack[e], opre.max_resize(opre.rgb_to_gray(raw[e]) if rgb else raw[e]))
                      for e in range(N)])
    assert np.array_equal(s2.cpu().numpy(), want2)


@pytest.mark.parametrize("path", GOLDEN[:1], ids=[os.path.basename(p) for p in GOLDEN[:1]])
def test_preprocess_reproduces_reference_states(path):
    """Raw frames of the golden env (regenerated from its seeds) pushed through the HIP kernel reproduce the
    stacked states the reference's FramePool/ObservationPool + PIL produced."""
    from golden_env import GoldenEnv
    from paac_amd import hip_ops
    g = np.load(path)
    N, T, A = int(g["N"]), int(g["T"]), int(g["A"])
    envs = [GoldenEnv(i, A, opre.FramePoolOracle, opre.ObservationPoolOracle, opre.max_resize, float(g["terminal_p"]))
            for i in range(N)]
    for e in envs:
        e.raw_log = []
        e.get_initial_state()
    stack = torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device="cuda")
    for k in range(4):        # initial state = 4 pushes (atari_emulator.py:88-96)
        hip_ops.preprocess_stack(dev(np.stack([e.raw_log[k] for e in envs])), stack, stack)
    states0 = g["states"][0].reshape(T, N, 84, 84, 4)
    assert np.array_equal(stack.cpu().numpy(), states0[0])
    acts = np.argmax(g["actions"][0], axis=1).reshape(T, N)
    # one more step without resets: env e continues unless it hit a terminal
    nxt, over = [], []
    for i, e in enumerate(envs):
        e.raw_log = []
        obs, r, term = e.next(np.eye(A)[acts[0, i]])
        nxt.append(e.raw_log[0])
        over.append(term)
    hip_ops.preprocess_stack(dev(np.stack(nxt)), stack, stack)
    got = stack.cpu().numpy()
    for i in range(N):
        if not over[i]:
            assert np.array_equal(got[i], states0[1, i])


@pytest.mark.parametrize("raw_frames", [False, True])
def test_synthetic_env_matches_host_spec(raw_frames):
    from paac_amd import hip_ops
    from paac_amd.synthetic import SyntheticEnvironment, terminal_threshold
    N, A, seed, off, p = 6, 4, 3, 10, 0.2
    envs = [SyntheticEnvironment(off + i, A, seed=seed, terminal_p=p, raw_frames=raw_frames) for i in range(N)]
    host = np.stack([e.get_initial_state() for e in envs])
    raw = torch.zeros((N, 2, 210, 160), dtype=torch.uint8, device="cuda") if raw_frames else None
    a = torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device="cuda")
    b = torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device="cuda")
    hip_ops.synth_reset(seed, off, a, raw)
    assert np.array_equal(a.cpu().numpy(), host)
    tick = torch.zeros(1, dtype=torch.int64, device="cuda")
    rew = torch.zeros(N, device="cuda")
    msk = torch.zeros(N, device="cuda")
    ep_r = torch.zeros(N, device="cuda")
    ep_l = torch.zeros(N, dtype=torch.int32, device="cuda")
    fin = torch.zeros(hip_ops.FINISHED_RING_BYTES // 4, dtype=torch.int32, device="cuda")
    rs = np.random.RandomState(0)
    tot_r = np.zeros(N)
    tot_l = np.zeros(N, dtype=int)
    finished = []
    for step in range(25):
        acts = rs.randint(0, A, N).astype(np.int32)
        hip_ops.synth_step(seed, off, dev(acts), terminal_threshold(p), tick, step % 3, a, b, rew, msk, ep_r, ep_l, fin,
                           raw_scratch=raw)
        if step % 3 == 2:
            hip_ops.counter_add(tick, 3)
        want_s, want_r, want_m = [], [], []
        for i, e in enumerate(envs):
            obs, r, term = e.next(np.eye(A)[acts[i]])
            tot_r[i] += r
            tot_l[i] += 1
            if term:
                obs = e.get_initial_state()
                finished.append((tot_r[i], tot_l[i]))
                tot_r[i], tot_l[i] = 0, 0
            want_s.append(obs)
            want_r.append(oroll.rescale_reward(r))
            want_m.append(0.0 if term else 1.0)
        assert np.array_equal(b.cpu().numpy(), np.stack(want_s)), "step %d" % step
        assert np.array_equal(rew.cpu().numpy(), np.asarray(want_r, dtype=np.float32))
        assert np.array_equal(msk.cpu().numpy(), np.asarray(want_m, dtype=np.float32))
        a, b = b, a
    assert len(finished) > 0
    f = fin.cpu().numpy()
    assert int(f[0]) == len(finished)
    got = sorted(zip(f[2:2 + len(finished)].view(np.float32).tolist(), f[2 + 4096:2 + 4096 + len(finished)].tolist()))
    assert got == sorted((float(r), int(l)) for r, l in finished)
    assert np.array_equal(ep_r.cpu().numpy(), tot_r.astype(np.float32))
    assert np.array_equal(ep_l.cpu().numpy(), tot_l.astype(np.int32))


@pytest.mark.parametrize("managed", [False, True])
@pytest.mark.parametrize("arch,A,N", [("NATURE", 4, 32), ("NATURE", 6, 33), ("NIPS", 18, 16), ("NATURE", 18, 60),
                                      # the sampler workgroup finishes the heads in registers up to 32 environments x 7 actions
                                      # (misc.hip: FusedHeadsHook): its corners
                                      ("NATURE", 7, 32), ("NIPS", 2, 5), ("NATURE", 6, 17), ("NATURE", 8, 28),
                                      # the large shards (four launches; managed: the sampler's MT19937 doubles come from the
                                      # spare workgroup of the fc launch, csrc/mt_ahead.h): configs[4] and configs[2] per GPU
                                      ("NATURE", 18, 128), ("NATURE", 4, 256), ("NATURE", 9, 100)])
def test_act_step_equals_separate_calls(arch, A, N, managed):
    from paac_amd._lib import check as _lib_check
    """paac_act_step_mt (forward with the head contractions in the fc epilogue, then heads finish + MT19937 sampler +
    synthetic env step in ONE launch) == paac_forward + paac_sample_mt + paac_synth_step, bit for bit, over consecutive
    steps (probabilities, values, actions, stream position, stacks, rewards, masks, episode bookkeeping)."""
    from oracle import network as onet
    from paac_amd import hip_ops
    from paac_amd.synthetic import terminal_threshold
    arch_id = {"NIPS": 0, "NATURE": 1}[arch]
    ctx = hip_ops.Context(arch_id, A, max_batch=N)
    host = onet.init_params(arch, A, np.random.RandomState(1), dtype=np.float32)
    flat = np.zeros(ctx.layout["total"], dtype=np.float32)
    for t in ctx.layout["tensors"]:
        flat[t["offset"]:t["offset"] + t["size"]] = host[t["name"]].reshape(-1)
    p = torch.from_numpy(flat).cuda()
    if managed:       # the learner's mode: explicit weight packing, conv3 -> fc hand-off in fragment order, no kept activations
        ctx.set_managed_weights(True)
        ctx.pack_weights(p)
    env_seed, off, thr = 5, 3, terminal_threshold(0.15)
    rs = np.random.RandomState(77)

    def fresh():
        d = dict(s0=torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device="cuda"),
                 s1=torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device="cuda"),
                 act=torch.zeros(N, dtype=torch.int32, device="cuda"), probs=torch.zeros((N, A), device="cuda"),
                 val=torch.zeros(N, device="cuda"), rew=torch.zeros(N, device="cuda"), msk=torch.zeros(N, device="cuda"),
                 ep_r=torch.zeros(N, device="cuda"), ep_l=torch.zeros(N, dtype=torch.int32, device="cuda"),
                 fin=torch.zeros(hip_ops.FINISHED_RING_BYTES // 4, dtype=torch.int32, device="cuda"),
                 tick=torch.zeros(1, dtype=torch.int64, device="cuda"),
                 mt=hip_ops.mt_state_from_numpy(rs.get_state(), "cuda"))
        hip_ops.synth_reset(env_seed, off, d["s0"], None)
        return d

    a, b = fresh(), fresh()
    scratch = hip_ops.sample_mt_scratch(N, A, "cuda")
    walk = hip_ops.walk_scratch(N, A, "cuda") if N > hip_ops.ACT_STEP_MAX_ENVS else None
    for step in range(8):
        if step == 5 and walk is not None:
            # the stream moves between two steps (someone else drew from np.random): the next step's record is made from
            # the state it finds, whatever the previous step left (unmanaged contexts make no record: every sampler
            # workgroup builds its own blocks and doubles there)
            for d in (a, b):
                hip_ops.sample_mt(d["probs"], d["mt"], scratch, d["act"])
        if walk is not None and managed and step in (2, 6):
            # the sampler workgroups finish the heads of their own environments there and report exact zeros with their
            # tickets: make one report a zero it has not seen -- the last-ticket workgroup then finishes ALL heads and walks
            # the serial way; ordinary probabilities, so the result is unchanged
            _lib_check(ctx.lib.paac_debug_report_zero(0 if step == 2 else 3))
        try:
            ctx.act_step_mt(p, a["s0"], a["mt"], a["act"], a["probs"], a["val"], env_seed, off, thr, a["tick"], 0, a["s1"],
                            a["rew"], a["msk"], a["ep_r"], a["ep_l"], a["fin"], walk_scratch=walk)
            torch.cuda.synchronize()
        finally:
            if walk is not None and managed and step in (2, 6):
                _lib_check(ctx.lib.paac_debug_report_zero(-1))        # process-wide: never left on for the next test
        if walk is not None and managed and step in (2, 6):
            assert int(walk[:4].view(torch.int32).item()) == 0        # the ticket word is back at 0
        ctx.forward(p, b["s0"], probs=b["probs"], values=b["val"])
        hip_ops.sample_mt(b["probs"], b["mt"], scratch, b["act"])
        hip_ops.synth_step(env_seed, off, b["act"], thr, b["tick"], 0, b["s0"], b["s1"], b["rew"], b["msk"], b["ep_r"],
                           b["ep_l"], b["fin"])
        torch.cuda.synchronize()
        for k in ("probs", "val", "act", "mt", "rew", "msk", "ep_r", "ep_l", "s1"):
            assert torch.equal(a[k], b[k]), "step %d: %s differs" % (step, k)
        fa, fb = a["fin"].cpu().numpy(), b["fin"].cpu().numpy()
        n = int(fa[0])
        assert fa[0] == fb[0] and sorted(fa[2:2 + n].tolist()) == sorted(fb[2:2 + n].tolist())
        for d in (a, b):
            hip_ops.counter_add(d["tick"], 1)
            d["s0"], d["s1"] = d["s1"], d["s0"]
    assert int(a["fin"][0]) > 0
    # the probabilities are the oracle's (float64 restatement) within the forward tolerance
    states = b["s1"].cpu().numpy()        # after the swap: the stacks the LAST step observed
    ref = onet.forward(host, states, arch, dtype=np.float64)
    assert np.abs(a["probs"].cpu().numpy() - ref["pi"]).max() < 1e-5
    assert np.abs(a["val"].cpu().numpy() - ref["v"]).max() < 1e-4
    ctx.close()


@pytest.mark.parametrize("N,A,multi", [(32, 4, False), (256, 4, False), (128, 18, False), (96, 6, False),
                                       (256, 4, True), (128, 18, True), (96, 6, True), (200, 3, True), (65, 2, True),
                                       (32, 4, True), (100, 9, True)])      # 9 actions: the all-categories-at-once hop, J = 8
def test_fused_sampler_env_step_equals_separate_calls(N, A, multi):
    """paac_sample_mt_synth_step (small and large-LDS variants: 256 environments x 4 actions = two-level table chase with
    one shift workgroup per environment; 128 x 18 = lane walk; multi: the group walks of the large shards spread over
    several workgroups through the caller's walk scratch) == paac_sample_mt + paac_synth_step, bit for bit, and the actions
    are numpy's."""
    from paac_amd import hip_ops
    from paac_amd.synthetic import terminal_threshold
    env_seed, off, thr = 9, 2, terminal_threshold(0.2)
    rs = np.random.RandomState(31)
    gen = np.random.RandomState(N + A)

    def fresh():
        d = dict(s0=torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device="cuda"),
                 s1=torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device="cuda"),
                 s2=torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device="cuda"),
                 act=torch.zeros(N, dtype=torch.int32, device="cuda"), rew=torch.zeros(N, device="cuda"),
                 msk=torch.zeros(N, device="cuda"), ep_r=torch.zeros(N, device="cuda"),
                 ep_l=torch.zeros(N, dtype=torch.int32, device="cuda"),
                 fin=torch.zeros(hip_ops.FINISHED_RING_BYTES // 4, dtype=torch.int32, device="cuda"),
                 tick=torch.zeros(1, dtype=torch.int64, device="cuda"),
                 mt=hip_ops.mt_state_from_numpy(rs.get_state(), "cuda"))
        hip_ops.synth_reset(env_seed, off, d["s0"], None)
        return d

    a, b = fresh(), fresh()
    ref_rs = np.random.RandomState(31)
    scratch = hip_ops.sample_mt_scratch(N, A, "cuda")
    walk = hip_ops.walk_scratch(N, A, "cuda") if multi else None
    for step in range(6 if multi else 4):
        logits = gen.randn(N, A) * 1.5
        p = np.exp(logits - logits.max(1, keepdims=True))
        p = np.maximum(p / p.sum(1, keepdims=True), 1e-6).astype(np.float32)
        p = (p / p.sum(1, keepdims=True)).astype(np.float32)
        pd = dev(p)
        hip_ops.sample_mt_synth_step(pd, a["mt"], a["act"], env_seed, off, thr, a["tick"], 0, a["s0"], a["s1"], a["rew"],
                                     a["msk"], a["ep_r"], a["ep_l"], a["fin"], stack_out2=a["s2"], walk_scratch=walk)
        hip_ops.sample_mt(pd, b["mt"], scratch, b["act"])
        hip_ops.synth_step(env_seed, off, b["act"], thr, b["tick"], 0, b["s0"], b["s1"], b["rew"], b["msk"], b["ep_r"],
                           b["ep_l"], b["fin"], stack_out2=b["s2"])
        torch.cuda.synchronize()
        for k in ("act", "mt", "rew", "msk", "ep_r", "ep_l", "s1", "s2"):
            assert torch.equal(a[k], b[k]), "step %d: %s differs" % (step, k)
        assert torch.equal(a["s1"], a["s2"])
        want = osamp.sample_numpy_reference(p, ref_rs)
        assert np.array_equal(a["act"].cpu().numpy(), np.asarray(want, dtype=np.int32))
        if multi:       # the workgroup that took the last ticket put the ticket counter back: it never leaves [0, W]
            assert int(walk[:4].view(torch.int32).item()) == 0
        for d in (a, b):
            hip_ops.counter_add(d["tick"], 1)
            d["s0"], d["s1"] = d["s1"], d["s0"]
    assert hip_ops.mt_state_to_numpy(a["mt"])[2] == ref_rs.get_state()[2]


@pytest.mark.parametrize("N,A", [(256, 4), (128, 18), (96, 6)])
def test_multi_sampler_exact_zero_fallback(N, A):
    """Multi-workgroup sampler: a step in which some float32(p - epsneg) is EXACTLY zero (p == 2^-24; numpy draws nothing
    for that category) takes the serial path in workgroup 0 alone -- the other sampler workgroups leave without a ticket --
    and the steps around it take the spread walks on the same walk scratch: actions and stream position are numpy's in
    all of them, and the ticket counter is 0 after every launch."""
    from paac_amd import hip_ops
    from paac_amd.synthetic import terminal_threshold
    env_seed, off, thr = 4, 0, terminal_threshold(0.2)
    gen = np.random.RandomState(5 * N + A)
    ref_rs = np.random.RandomState(13)
    mt = hip_ops.mt_state_from_numpy(np.random.RandomState(13).get_state(), "cuda")
    s = [torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    hip_ops.synth_reset(env_seed, off, s[0], None)
    act = torch.zeros(N, dtype=torch.int32, device="cuda")
    rew, msk, ep_r = (torch.zeros(N, device="cuda") for _ in range(3))
    ep_l = torch.zeros(N, dtype=torch.int32, device="cuda")
    tick = torch.zeros(1, dtype=torch.int64, device="cuda")
    walk = hip_ops.walk_scratch(N, A, "cuda")
    eps = np.float32(2.0 ** -24)
    for step, zeros in enumerate([False, True, False, True, True, False]):
        logits = gen.randn(N, A) * 1.5
        p = np.exp(logits - logits.max(1, keepdims=True))
        p = np.maximum(p / p.sum(1, keepdims=True), 1e-6).astype(np.float32)
        p = (p / p.sum(1, keepdims=True)).astype(np.float32)
        if zeros:
            for e in gen.choice(N, 5, replace=False):
                j = gen.randint(0, A - 1)
                p[e, A - 1] += p[e, j] - eps          # keep the row sum; the last category is never drawn for
                p[e, j] = eps
            assert ((p[:, :A - 1] - eps) == 0).sum() == 5
        hip_ops.sample_mt_synth_step(dev(p), mt, act, env_seed, off, thr, tick, 0, s[0], s[1], rew, msk, ep_r, ep_l, None,
                                     walk_scratch=walk)
        torch.cuda.synchronize()
        want = osamp.sample_numpy_reference(p, ref_rs)
        assert np.array_equal(act.cpu().numpy(), np.asarray(want, dtype=np.int32)), "step %d" % step
        assert hip_ops.mt_state_to_numpy(mt)[2] == ref_rs.get_state()[2], "step %d: stream position" % step
        assert int(walk[:4].view(torch.int32).item()) == 0
        hip_ops.counter_add(tick, 1)
        s.reverse()
