"""-m gpu: the whole PAAC cycle through the drop-in surface vs the oracle restatement of paac.py:59-183."""
import argparse
import copy
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import network as onet
from oracle import rollout as oroll
from oracle import sampler as osamp


def make_args(**kw):
    from paac_amd import train
    args = train.get_arg_parser().parse_args([])
    args.debugging_folder = tempfile.mkdtemp(prefix="paac_test_")
    for k, v in kw.items():
        setattr(args, k, v)
    return args


def build_learner(args, params_seed=0):
    from paac_amd import train
    from paac_amd.paac import PAACLearner
    network_creator, env_creator = train.get_network_and_environment_creator(args)
    learner = PAACLearner(network_creator, env_creator, args)
    arch = "NIPS" if args.arch == "NIPS" else "NATURE"
    params = onet.init_params(arch, args.num_actions, np.random.RandomState(params_seed), dtype=np.float32)
    learner.network.set_parameters(params)
    learner.network.init = lambda folder, saver, session: 0      # keep the injected weights
    return learner, params, env_creator


class OracleLoop:
    """The reference loop restated on the CPU: same envs, same np.random sampler stream, fp32-param / fp64-math net.
    `p`, `ms`, `mom` ({tensor name: array}) may be replaced between cycles (the long device-loop case restarts every cycle
    from the weights and optimizer slots the device really holds)."""

    def __init__(self, args, params, env_creator, arch):
        self.args, self.arch = args, arch
        A, N, T = args.num_actions, args.emulator_counts, args.max_local_steps
        envs = [env_creator.create_environment(i) for i in range(N)]
        self.rs = np.random.RandomState(args.test_seed)
        self.p = {k: v.copy() for k, v in params.items()}
        self.ms, self.mom = onet.rmsprop_init(self.p)

        def policy_fn(states):
            out = onet.forward(self.p, states, arch, dtype=np.float64)
            return out["v"].astype(np.float32), out["pi"].astype(np.float32)

        self.ro = oroll.OracleRollout(envs, A, T, args.gamma, args.initial_lr, args.lr_annealing_steps, policy_fn,
                                      lambda pi: osamp.sample_mt_restated(pi, self.rs)[0])

    def cycle(self):
        args = self.args
        cyc = self.ro.cycle()
        L, g = onet.loss_and_grads(self.p, cyc["states"], cyc["actions"], cyc["y"].astype(np.float32),
                                   cyc["adv"].astype(np.float32), args.entropy_regularisation_strength, self.arch,
                                   dtype=np.float64)
        gc, gn = onet.clip_by_global_norm(g, args.clip_norm, args.clip_norm_type)
        self.p, self.ms, self.mom = onet.rmsprop_step(self.p, {k: v.astype(np.float32) for k, v in gc.items()}, self.ms,
                                                      self.mom, np.float32(cyc["lr"]), args.alpha, 0.0, args.e)
        cyc["params"] = {k: v.copy() for k, v in self.p.items()}
        cyc["gnorm"] = gn
        cyc["episodes"] = list(self.ro.finished_episodes)        # (global_step, reward, length) so far: paac.py:130-135
        return cyc


def oracle_cycles(args, params, env_creator, cycles, arch):
    loop = OracleLoop(args, params, env_creator, arch)
    return [loop.cycle() for _ in range(cycles)]


@pytest.mark.parametrize("game,arch,N,T", [("pong", "NIPS", 8, 5), ("breakout", "NATURE", 8, 5)])
def test_host_loop_matches_oracle(game, arch, N, T):
    """BASELINE config 1 (Pong / NIPS / 8 envs / t_max 5) and a Nature twin, host BaseEnvironment plugins."""
    cycles = 3
    feeds = []
    args = make_args(game=game, arch=arch, emulator_counts=N, emulator_workers=0, max_local_steps=T,
                     max_global_steps=cycles * N * T, host_environments=True, record_feeds=True,
                     feed_callback=feeds.append, synthetic_terminal_p=0.1, test_seed=42)
    learner, params, env_creator = build_learner(args)
    np.random.seed(args.test_seed)
    learner.train()
    want = oracle_cycles(args, params, env_creator, cycles, "NIPS" if arch == "NIPS" else "NATURE")
    assert len(feeds) == cycles
    for c in range(cycles):
        assert np.array_equal(feeds[c]["states"], want[c]["states"]), "states differ in cycle %d" % c
        assert np.array_equal(feeds[c]["actions"], np.argmax(want[c]["actions"], axis=1)), "actions differ in cycle %d" % c
        assert np.abs(feeds[c]["values"] - want[c]["values"]).max() < 1e-4
        assert np.abs(feeds[c]["y"] - want[c]["y"]).max() < 2e-4
        assert feeds[c]["lr"] == want[c]["lr"]
        assert feeds[c]["global_step"] == want[c]["global_step"]
    got = learner.network.get_parameters()
    for k, v in want[-1]["params"].items():
        assert np.abs(got[k] - v).max() < 2e-4, k
    # the episode records (paac.py:130-135: global_step at the moment the per-environment loop reaches the finished
    # environment, total unclipped reward, length) -- the host loop's bookkeeping is vectorised, the records must not move
    import json, os
    recs = [json.loads(l) for l in open(os.path.join(args.debugging_folder, "metrics.jsonl"))]
    got_eps = [(r["global_step"], r["reward"], r["length"]) for r in recs if r["kind"] == "episode"]
    want_eps = [(int(g), float(r), int(l)) for g, r, l in want[-1]["episodes"]]
    assert len(want_eps) >= 3 and got_eps == want_eps
    # sampler stream position was written back to the global np.random like the reference leaves it
    rs = np.random.RandomState(args.test_seed)
    for c in range(cycles):
        for t in range(T):
            osamp.sample_mt_restated(want[c]["pis"][t], rs)
    assert np.random.get_state()[2] == rs.get_state()[2]


@pytest.mark.parametrize("raw_frames,use_graph", [(False, False), (False, True), (True, True)])
def test_device_loop_matches_host_loop(raw_frames, use_graph):
    """Device-resident cycle (hipGraph replay, device envs) == the host loop on the same envs and sampler stream."""
    N, T, cycles = 8, 5, 4
    common = dict(game="breakout", arch="NATURE", emulator_counts=N, emulator_workers=0, max_local_steps=T,
                  max_global_steps=cycles * N * T, synthetic_terminal_p=0.1, synthetic_raw_frames=raw_frames)
    feeds = []
    a_host = make_args(host_environments=True, record_feeds=True, feed_callback=feeds.append, **common)
    host, params, _ = build_learner(a_host)
    np.random.seed(7)
    host.train()
    a_dev = make_args(sampler="numpy", use_graph=use_graph, **common)
    devl, _, _ = build_learner(a_dev)
    np.random.seed(7)
    devl.global_step = devl.init_network()
    from paac_amd.paac import DeviceRollout
    ro = DeviceRollout(devl, devl.environment_creator.device_env_spec, sampler="numpy", use_graph=use_graph)
    for c in range(cycles):
        ro.run_cycle()
        ro.synchronize()
        assert np.array_equal(ro.rollout_states().cpu().numpy(), feeds[c]["states"]), "cycle %d" % c
        assert np.array_equal(ro.actions.view(-1).cpu().numpy(), feeds[c]["actions"])
        assert np.allclose(ro.y.cpu().numpy(), feeds[c]["y"], atol=1e-5)
        assert abs(devl.lr_dev.item() - np.float32(feeds[c]["lr"])) == 0.0
    gh = host.network.get_parameters()
    gd = devl.network.get_parameters()
    for k in gh:
        assert np.abs(gh[k] - gd[k]).max() < 1e-5, k
    assert int(ro.global_step_dev.item()) == cycles * N * T
    ro.close()


@pytest.mark.parametrize("game,N,T,cycles,arch,raw", [
    ("breakout", 32, 5, 3, "NATURE", False),      # BASELINE configs[1] (the headline): fused numpy-parity sampler + env-step launch
    ("qbert", 32, 5, 2, "NATURE", False),         # BASELINE configs[3] per-GPU shard (A=6)
    ("seaquest", 128, 20, 1, "NATURE", False),    # BASELINE configs[4] per-GPU shard (A=18): 2176 draws -> the large-LDS sampler, lane walk
    ("breakout", 256, 5, 1, "NATURE", False),     # BASELINE configs[2]: 256 environments -> the large-LDS sampler, two-level table chase
    # ragged shapes (tools/stress_shapes.py runs more of them): one environment and one step; odd counts on both
    # networks; one past the 64-environment limit of the three-launch acting step
    ("breakout", 1, 1, 2, "NATURE", False),
    ("qbert", 7, 5, 2, "NIPS", False),
    ("seaquest", 5, 7, 2, "NATURE", False),
    ("breakout", 65, 5, 1, "NATURE", False),
    # path B (raw 210x160 screen pairs written by the step launch, max / PIL-nearest resize / history push by the preprocess
    # launch): BASELINE configs[2]'s "large-batch preprocess/HBM path" at its 256 environments, the headline shape, a ragged one
    ("breakout", 256, 5, 1, "NATURE", True),
    ("breakout", 32, 5, 2, "NATURE", True),
    ("qbert", 9, 3, 2, "NATURE", True),
    # twenty updates at the headline shape: the weights move (lr 0.0224, clip 3.0) and the actions stay bit-equal
    ("breakout", 32, 5, 20, "NATURE", False),
])
def test_device_loop_matches_oracle(game, N, T, cycles, arch, raw):
    """The device-resident cycle (hipGraph replay, numpy-parity sampler) against the CPU restatement of paac.py:99-165
    on the same synthetic environments and np.random stream: observations and actions bit for bit, values / returns /
    weights within the float tolerance."""
    from paac_amd import hip_ops
    from paac_amd.paac import DeviceRollout
    args = make_args(game=game, arch=arch, emulator_counts=N, emulator_workers=0, max_local_steps=T,
                     max_global_steps=1 << 40, synthetic_terminal_p=0.05, sampler="numpy", test_seed=11,
                     synthetic_raw_frames=raw)
    learner, params, env_creator = build_learner(args)
    A = args.num_actions
    assert N * (A - 1) <= hip_ops.FUSED_SAMPLE_MAX_DRAWS       # every BASELINE shard runs the fused sampler + env step
    np.random.seed(args.test_seed)
    learner.global_step = learner.init_network()
    ro = DeviceRollout(learner, env_creator.device_env_spec, sampler="numpy", use_graph=True)
    # Up to a few cycles the oracle runs on its own from the initial weights.  The LONG case restarts the oracle every cycle
    # from the weights and RMSProp slots the device really holds: the derivative of a ReLU network is discontinuous, so an
    # fp32 pre-activation that lands on the other side of 0 than the float64 one (a handful per million units and cycle)
    # moves a few weights by ~1e-6, and at lr 0.0224 the two trajectories then drift apart by a factor of 1.3-3 per update
    # whatever the kernels do (tools/probe_quarter.py prints the free-running drift: 1e-8 -> 1e-4..1e-3 over 20 cycles, a
    # different curve for every summation order).  Restarted, EVERY cycle is held to the single-cycle bars, at the weights
    # the run has really reached.
    forced = cycles > 5
    loop = OracleLoop(args, params, env_creator, arch)
    want = []
    for c in range(cycles):
        if forced:
            loop.p = learner.network.get_parameters()
            loop.ms = learner.network.get_parameters(learner.rms)
            loop.mom = learner.network.get_parameters(learner.mom)
        want.append(loop.cycle())
        ro.run_cycle()
        ro.synchronize()
        assert np.array_equal(ro.actions.view(-1).cpu().numpy(), np.argmax(want[c]["actions"], axis=1)), "cycle %d" % c
        assert np.array_equal(ro.rollout_states().cpu().numpy(), want[c]["states"]), "cycle %d" % c
        assert np.abs(ro.values.cpu().numpy() - want[c]["values"]).max() < 1e-4, "cycle %d" % c
        assert np.abs(ro.y.cpu().numpy() - want[c]["y"]).max() < 2e-4
        assert np.abs(ro.adv.cpu().numpy() - want[c]["adv"]).max() < 3e-4
        assert float(learner.lr_dev.item()) == float(np.float32(want[c]["lr"]))
        assert int(ro.global_step_dev.item()) == want[c]["global_step"]
        if forced:        # one update from the same weights: the bar of a single step (a flipped unit moves a weight by ~1e-6)
            got = learner.network.get_parameters()
            for k, v in want[c]["params"].items():
                assert np.abs(got[k] - v).max() < 2e-5, "cycle %d: %s" % (c, k)
    got = learner.network.get_parameters()
    for k, v in want[-1]["params"].items():
        assert np.abs(got[k] - v).max() < 2e-4, k
    rs = np.random.RandomState(args.test_seed)          # stream position after the last cycle
    for c in range(cycles):
        for t in range(T):
            osamp.sample_mt_restated(want[c]["pis"][t], rs)
    assert hip_ops.mt_state_to_numpy(ro.mt_state)[2] == rs.get_state()[2]
    ro.close()


def test_choose_next_actions_dropin():
    """Static helper used by the reference's test.py:77 -- same signature, numpy in/out, global np.random stream."""
    from paac_amd.paac import PAACLearner
    args = make_args(game="qbert", arch="NATURE", emulator_counts=4, max_local_steps=2, max_global_steps=0)
    learner, params, _ = build_learner(args)
    states = np.random.RandomState(0).randint(0, 256, (4, 84, 84, 4)).astype(np.uint8)
    np.random.seed(3)
    acts, v, pi = PAACLearner.choose_next_actions(learner.network, args.num_actions, states, learner.session)
    ref = onet.forward(params, states, "NATURE", dtype=np.float64)
    assert acts.shape == (4, args.num_actions) and np.all(acts.sum(1) == 1)
    assert np.abs(pi - ref["pi"]).max() < 1e-5 and np.abs(v - ref["v"]).max() < 1e-4
    rs = np.random.RandomState(3)
    want, _ = osamp.sample_mt_restated(pi, rs)
    assert list(np.argmax(acts, 1)) == want
    assert np.random.get_state()[2] == rs.get_state()[2]


@pytest.mark.parametrize("fmt", ["npz", "tf"])
def test_checkpoint_roundtrip(fmt):
    """fmt = tf: the reference's own container (TensorFlow V2 tensor bundle, paac_amd/tf_bundle.py); the network bundle
    then holds the optimizer slots too, like the reference's tf.train.Saver() over all variables."""
    import os
    args = make_args(game="pong", arch="NIPS", emulator_counts=4, max_local_steps=2, max_global_steps=16,
                     checkpoint_format=fmt)
    learner, params, _ = build_learner(args)
    learner.train()                       # 2 cycles on device envs, cleanup() saves
    saved = learner.network.get_parameters()
    files = sorted(os.listdir(os.path.join(args.debugging_folder, "checkpoints")))
    assert files == (["-16.npz"] if fmt == "npz" else ["-16.data-00000-of-00001", "-16.index", "checkpoint"])
    if fmt == "tf":
        from paac_amd import tf_bundle
        meta = tf_bundle.entries(os.path.join(args.debugging_folder, "checkpoints", "-16"))
        assert len(meta) == 30 and meta["local_learning_1/fc3_weights/OptimizerVariables_1"]["shape"] == (2592, 256)
    a2 = copy.copy(args)
    from paac_amd import train
    from paac_amd.paac import PAACLearner
    nc, ec = train.get_network_and_environment_creator(a2)
    l2 = PAACLearner(nc, ec, a2)
    step = l2.init_network()
    assert step == 16
    for k, v in l2.network.get_parameters().items():
        assert np.array_equal(v, saved[k])
    r1, r2 = learner.network.get_parameters(learner.rms), l2.network.get_parameters(l2.rms)
    for k in r1:
        assert np.array_equal(r1[k], r2[k]) and not np.all(r1[k] == 1.0)


def test_cleanup_right_after_async_cycles_saves_a_consistent_checkpoint():
    """cleanup() (what the signal handler ends in) straight after run_cycles(), with graph-replayed cycles still in
    flight on the rollout's own stream: the checkpoint must hold the weights and optimizer slots of the LAST finished
    update -- save_vars synchronises the rollout stream first."""
    from paac_amd.paac import DeviceRollout
    from paac_amd.session import Saver, checkpoint_key
    import os
    args = make_args(game="breakout", arch="NATURE", emulator_counts=8, emulator_workers=0, max_local_steps=5,
                     max_global_steps=1 << 40, synthetic_terminal_p=0.05)
    learner, _, env_creator = build_learner(args)
    learner.global_step = learner.init_network()
    learner.rollout = DeviceRollout(learner, env_creator.device_env_spec, sampler="philox", use_graph=True)
    learner.rollout.run_cycles(24)                 # asynchronous: returns while the GPU is still replaying
    learner.global_step += 24 * 40
    learner.cleanup()                              # no explicit synchronize before it
    torch.cuda.synchronize()
    want = learner.network.get_parameters()
    want_rms = learner.network.get_parameters(learner.rms)
    with np.load(Saver.latest_checkpoint(os.path.join(args.debugging_folder, "checkpoints"))) as z:
        for k, v in want.items():
            assert np.array_equal(z[checkpoint_key("local_learning", k)], v), k
    with np.load(Saver.latest_checkpoint(os.path.join(args.debugging_folder, "optimizer_checkpoints"))) as z:
        for k, v in want_rms.items():
            assert np.array_equal(z[checkpoint_key("local_learning", k, "OptimizerVariables")], v), k
        assert not np.all(z[checkpoint_key("local_learning", "fc4_weights", "OptimizerVariables")] == 1.0)


def test_cpu_device_is_rejected():
    args = make_args(device="/cpu:0", emulator_counts=2)
    with pytest.raises(RuntimeError):
        build_learner(args)


def test_batched_graph_launches_equal_single_cycles():
    """DeviceRollout.run_cycles(n) replays 4 cycles per hipGraph launch where it can: same trajectory, same weights,
    same counters as n single-cycle launches."""
    from paac_amd.paac import DeviceRollout
    N, T, cycles = 8, 5, 11
    outs = []
    for batched in (False, True):
        args = make_args(game="breakout", arch="NATURE", emulator_counts=N, emulator_workers=0, max_local_steps=T,
                         max_global_steps=1 << 40, synthetic_terminal_p=0.1)
        learner, _, _ = build_learner(args)
        np.random.seed(5)
        learner.global_step = learner.init_network()
        ro = DeviceRollout(learner, learner.environment_creator.device_env_spec, sampler="numpy", use_graph=True)
        if batched:
            ro.run_cycles(1)            # ring parity 1 from here: the batched graphs that START at parity 1
            ro.run_cycles(cycles - 1)   # 2 x 4 batched + 2 single
        else:
            for _ in range(cycles):
                ro.run_cycle()
        ro.synchronize()
        outs.append((learner.network.get_parameters(), ro.actions.cpu().numpy().copy(), int(ro.global_step_dev.item()),
                     float(learner.lr_dev.item()), ro.rollout_states().cpu().numpy().copy()))
        ro.close()
    (p0, a0, g0, lr0, s0), (p1, a1, g1, lr1, s1) = outs
    assert g0 == g1 == cycles * N * T and lr0 == lr1
    assert np.array_equal(a0, a1) and np.array_equal(s0, s1)
    for k in p0:
        assert np.array_equal(p0[k], p1[k]), k


@pytest.mark.parametrize("arch,A,N", [("NATURE", 4, 32), ("NIPS", 18, 8), ("NATURE", 6, 300)])
def test_philox_step_inside_heads_equals_separate_calls(arch, A, N):
    """paac_forward_sample_synth_step (row i's heads workgroup also does environment i's bookkeeping, extra workgroups
    shift the stacks) == paac_forward_sample followed by paac_synth_step, bit for bit, over consecutive steps."""
    from paac_amd import hip_ops
    from paac_amd.synthetic import terminal_threshold
    arch_id = {"NIPS": 0, "NATURE": 1}[arch]
    ctx = hip_ops.Context(arch_id, A, max_batch=N)
    host = onet.init_params(arch, A, np.random.RandomState(1), dtype=np.float32)
    flat = np.zeros(ctx.layout["total"], dtype=np.float32)
    for t in ctx.layout["tensors"]:
        flat[t["offset"]:t["offset"] + t["size"]] = host[t["name"]].reshape(-1)
    p = torch.from_numpy(flat).cuda()
    env_seed, sampler_seed, off, thr = 5, 77, 3, terminal_threshold(0.15)

    def fresh():
        d = dict(s0=torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device="cuda"),
                 s1=torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device="cuda"),
                 act=torch.zeros(N, dtype=torch.int32, device="cuda"), probs=torch.zeros((N, A), device="cuda"),
                 val=torch.zeros(N, device="cuda"), rew=torch.zeros(N, device="cuda"), msk=torch.zeros(N, device="cuda"),
                 ep_r=torch.zeros(N, device="cuda"), ep_l=torch.zeros(N, dtype=torch.int32, device="cuda"),
                 fin=torch.zeros(hip_ops.FINISHED_RING_BYTES // 4, dtype=torch.int32, device="cuda"),
                 tick=torch.zeros(1, dtype=torch.int64, device="cuda"))
        hip_ops.synth_reset(env_seed, off, d["s0"], None)
        return d

    a, b = fresh(), fresh()
    for step in range(10):
        ctx.forward_sample_synth_step(p, a["s0"], sampler_seed, a["tick"], 0, off, a["act"], env_seed, thr, a["s1"],
                                      a["rew"], a["msk"], a["ep_r"], a["ep_l"], a["fin"], probs=a["probs"], values=a["val"])
        ctx.forward_sample(p, b["s0"], sampler_seed, b["tick"], 0, off, b["act"], probs=b["probs"], values=b["val"])
        hip_ops.synth_step(env_seed, off, b["act"], thr, b["tick"], 0, b["s0"], b["s1"], b["rew"], b["msk"], b["ep_r"],
                           b["ep_l"], b["fin"])
        torch.cuda.synchronize()
        for k in ("probs", "val", "act", "rew", "msk", "ep_r", "ep_l", "s1"):
            assert torch.equal(a[k], b[k]), "step %d: %s differs" % (step, k)
        fa, fb = a["fin"].cpu().numpy(), b["fin"].cpu().numpy()
        n = int(fa[0])
        assert fa[0] == fb[0] and sorted(fa[2:2 + n].tolist()) == sorted(fb[2:2 + n].tolist())
        for d in (a, b):
            hip_ops.counter_add(d["tick"], 1)
            d["s0"], d["s1"] = d["s1"], d["s0"]
    assert int(a["fin"][0]) > 0
    ctx.close()
