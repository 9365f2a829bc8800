"""-m gpu: two ranks (gloo, both on cuda:0 -- the box has one GPU) run the env-sharded device loop with the split
graphs + gradient all-reduce, and must track a single-process run over the concatenated environments."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, out_dir, n_per_rank, cycles, mode="single", corrupt=False):
    os.environ["PAAC_ALLREDUCE"] = mode
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import network as onet
    from paac_amd import train
    from paac_amd.paac import DeviceRollout, PAACLearner
    args = train.get_arg_parser().parse_args([])
    args.game, args.arch = "breakout", "NATURE"
    args.emulator_counts, args.max_local_steps, args.emulator_workers = n_per_rank, 3, 0
    args.max_global_steps = 1 << 40
    args.synthetic_terminal_p = 0.1
    args.debugging_folder = tempfile.mkdtemp(prefix="paac_dp_")
    nc, ec = train.get_network_and_environment_creator(args)
    L = PAACLearner(nc, ec, args)
    L.network.set_parameters(onet.init_params("NATURE", args.num_actions, np.random.RandomState(0), dtype=np.float32))
    ro = DeviceRollout(L, ec.device_env_spec, sampler="philox", sampler_seed=9, env_offset=rank * n_per_rank, use_graph=True)
    acts = []
    ro.run_cycle()
    ro.synchronize()
    acts.append(ro.actions.cpu().numpy().copy())
    ro.run_cycles(cycles - 1)       # back to back: with two ranks the optimizer step rides in front of the next cycle
    ro.synchronize()
    acts.append(ro.actions.cpu().numpy().copy())
    p = L.network.get_parameters()
    verdict = {}
    if world > 1:
        from paac_amd import parallel
        assert ro.check_replicas("grad") and ro.check_replicas("weights")       # identical after `cycles` exchanged updates
        if corrupt:
            # one rank's weights drift by one element: EVERY rank must see the mismatch (the comparison is a collective),
            # also after more cycles
            if rank == 1:
                L.network.params[12345] += 1e-3
            for when in ("at once", "after two more cycles"):
                try:
                    ro.check_replicas("weights")
                    verdict[when] = "identical"
                except parallel.ReplicaMismatch as exc:
                    verdict[when] = str(exc)
                ro.run_cycles(2)
    np.savez(os.path.join(out_dir, "w%d_r%d.npz" % (world, rank)), actions=np.stack(acts),
             gstep=int(ro.global_step_dev.item()), lr=float(L.lr_dev.item()), verdict=np.array(sorted(verdict.items())), **p)
    ro.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["single", "split"])
def test_two_ranks_track_single_process(tmp_path, mode):
    """Both exchange forms: one all-reduce after the full backward (default), and the two-piece one (PAAC_ALLREDUCE=split)."""
    import torch.multiprocessing as mp
    N, cycles = 4, 5
    mp.spawn(_run, args=(2, _free_port(), str(tmp_path), N, cycles, mode), nprocs=2, join=True)
    mp.spawn(_run, args=(1, 0, str(tmp_path), 2 * N, cycles), nprocs=1, join=True)
    r0 = np.load(tmp_path / "w2_r0.npz")
    r1 = np.load(tmp_path / "w2_r1.npz")
    one = np.load(tmp_path / "w1_r0.npz")
    assert int(r0["gstep"]) == int(r1["gstep"]) == int(one["gstep"]) == 2 * N * 3 * cycles
    assert float(r0["lr"]) == float(one["lr"])
    # first cycle: identical weights -> identical actions per environment (philox is keyed by the global env id)
    assert np.array_equal(np.concatenate([r0["actions"][0], r1["actions"][0]], axis=1), one["actions"][0])
    for k in one.files:
        if k in ("actions", "gstep", "lr", "verdict"):
            continue
        assert np.array_equal(r0[k], r1[k]), "replicated weights diverged: %s" % k
        assert np.abs(r0[k] - one[k]).max() < 2e-5, k


def test_diverged_replicas_are_detected_on_every_rank(tmp_path):
    """DeviceRollout.check_replicas (what train.py and bench.py call after the first update and every
    PAAC_REPLICA_CHECK_CYCLES cycles): MIN / MAX all-reduce of 64-bit checksums of weights and optimizer slots.  Identical
    replicas pass; one element changed on ONE rank raises parallel.ReplicaMismatch on BOTH."""
    import torch.multiprocessing as mp
    mp.spawn(_run, args=(2, _free_port(), str(tmp_path), 4, 3, "single", True), nprocs=2, join=True)
    for r in (0, 1):
        verdict = dict(np.load(tmp_path / ("w2_r%d.npz" % r))["verdict"].tolist())
        assert set(verdict) == {"at once", "after two more cycles"}
        for when, text in verdict.items():
            assert "replicas diverged" in text and "params" in text, (r, when, text)


def test_bench_two_rank_rehearsal():
    """bench.py's own N > 1 path (torch.distributed.run, barriers, phased gradient exchange in the graph replay AND in
    the event-timed pass, max over ranks, one JSON line from rank 0) rehearsed with two ranks on this one GPU over
    gloo (PAAC_BENCH_REHEARSAL=1).  A rank-0-only pass with collectives in it would hang here."""
    import json
    import subprocess
    env = dict(os.environ, PAAC_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
           "12", "--warmup", "3"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 12 and out["scaling"] == "weak" and out["finite_params"]
    assert out["config"]["global_envs"] == 64 and out["cpu_baseline"] is None
    assert out["roofline"]["bound"] in ("mfma", "hbm") and out["value"] > 0
    # the replicas were compared (after the warm-up and after the timed windows) and the line says how the exchange ran:
    # gloo cannot be captured, so the requested graph mode reports the eager form it really used
    assert out["replicas_identical"] is True
    assert out["exchange"]["mode"] == "single" and out["exchange"]["requested"] == "graph"


_RCCL_SMOKE = r"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, %(root)r)
from paac_amd import parallel, train
args = train.get_arg_parser().parse_args([])
assert parallel.init_from_env(args) == 1 and args.device == "/gpu:0"
import torch
import torch.distributed as dist
assert dist.is_initialized() and dist.get_backend() == "nccl"
from paac_amd.paac import DeviceRollout, PAACLearner
args.game, args.arch = "breakout", "NATURE"
args.emulator_counts, args.max_local_steps, args.emulator_workers = 8, 5, 0
args.max_global_steps = 1 << 40
args.synthetic_terminal_p = 0.1
out = {}
for mode in ("plain", "split", "single", "graph", "graph_refused"):
    os.environ["PAAC_FORCE_COLLECTIVES"] = "0" if mode == "plain" else "1"
    os.environ["PAAC_ALLREDUCE"] = "graph" if mode == "graph_refused" else mode if mode != "plain" else "single"
    args.debugging_folder = tempfile.mkdtemp(prefix="paac_rccl_")
    nc, ec = train.get_network_and_environment_creator(args)
    L = PAACLearner(nc, ec, args)
    L.network.initialize(np.random.RandomState(0))
    np.random.seed(4)
    ro = DeviceRollout(L, ec.device_env_spec, sampler="numpy", use_graph=True)
    assert ro.phased == (mode != "plain")
    checked = []
    if mode == "graph":            # the replayed cycle is compared with the eager exchange before it is trusted ...
        inner = ro._replay_matches_eager
        ro._replay_matches_eager = lambda: checked.append(inner()) or checked[-1]
    if mode == "graph_refused":    # ... and a replay that does not reproduce it is refused: the eager form takes over
        ro._replay_matches_eager = lambda: False
    ro.run_cycles(7)
    ro.synchronize()
    out[mode] = (L.network.params.cpu().numpy().copy(), ro.actions.cpu().numpy().copy(), int(ro.global_step_dev.item()))
    assert ro.graph_exchange == (mode == "graph")
    assert ro.exchange_mode == {"plain": "none", "graph_refused": "single"}.get(mode, mode)
    if mode == "graph":
        assert checked == [True] and ro.exchange_fallback is None
    if mode == "graph_refused":
        assert "differs from the eager" in ro.exchange_fallback
    if mode != "plain":
        assert ro.check_replicas("weights") and ro.check_replicas("grad")
    assert (ro.graph_ua[0] is not None) == (mode in ("split", "single", "graph_refused"))
    assert (ro.graph_conv[0] is not None) == (mode == "split")
    assert (ro.graph_multi[0] is not None) == (mode in ("plain", "graph"))      # MULTI cycles per launch survive the exchange
    ro.close()
for mode in ("split", "single", "graph", "graph_refused"):
    assert np.array_equal(out[mode][0], out["plain"][0]), mode + ": weights differ from the unphased run"
    assert np.array_equal(out[mode][1], out["plain"][1]) and out[mode][2] == out["plain"][2]
parallel.shutdown()
print("RCCL_SMOKE_OK")
"""


def test_phased_exchange_runs_under_rccl_world_of_one():
    """The data-parallel cycle -- the all-reduce captured INTO the cycle graph (the default), and graph_a / graph_conv /
    graph_ua around stream-ordered eager RCCL all-reduces (DeviceRollout._exchange, both eager forms) -- executed under
    backend "nccl" on this one GPU (a world of one with the collectives forced on): the all-reduce of one rank is the
    identity, so the weights after 7 cycles must equal the plain single-process replay bit for bit."""
    import subprocess
    env = dict(os.environ, PAAC_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    res = subprocess.run([sys.executable, "-c", _RCCL_SMOKE % dict(root=ROOT)], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=600)
    if res.returncode != 0 and os.path.isdir(os.path.join(ROOT, "gpurun_out")):      # the whole log, for the post-mortem
        open(os.path.join(ROOT, "gpurun_out", "rccl_smoke_failure.txt"), "w").write(res.stdout + "\n--- stderr ---\n" + res.stderr)
    assert res.returncode == 0 and "RCCL_SMOKE_OK" in res.stdout, (res.stdout[-1500:], res.stderr[-3000:])


def test_train_module_under_torchrun_two_ranks():
    """`python -m torch.distributed.run --nproc-per-node 2 -m paac_amd.train ...` (both ranks on this one GPU, gradients
    over gloo): train.main joins the group before touching the GPU, rank 0 alone writes args.json, checkpoints and
    metrics, both ranks finish the same number of cycles, and the run resumes from the checkpoint."""
    import json
    import subprocess
    folder = tempfile.mkdtemp(prefix="paac_dp_train_")
    env = dict(os.environ, PAAC_DIST_BACKEND="gloo", PAAC_DIST_SINGLE_DEVICE="1")
    steps = 2 * 8 * 5 * 6                     # 2 ranks x 8 envs x t_max 5 x 6 cycles
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "-m", "paac_amd.train", "-g", "breakout", "--arch", "NATURE",
           "-ec", "8", "-ew", "0", "--max_global_steps", str(steps), "-df", folder]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert res.stdout.count("Starting training (2 data-parallel ranks)") == 2
    assert json.load(open(os.path.join(folder, "args.json")))["emulator_counts"] == 8
    assert os.listdir(os.path.join(folder, "checkpoints")) == ["-%d.npz" % steps]
    assert os.listdir(os.path.join(folder, "optimizer_checkpoints")) == ["-%d.npz" % steps]
    # resume: both ranks restore step `steps`, run 2 more cycles, rank 0 writes the next checkpoint
    cmd[cmd.index("--max_global_steps") + 1] = str(steps + 2 * 80)
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert "Starting training at Step %d" % steps in res.stdout
    assert sorted(os.listdir(os.path.join(folder, "checkpoints"))) == sorted(["-%d.npz" % steps, "-%d.npz" % (steps + 160)])


def test_host_plugin_loop_counts_global_steps_over_all_ranks():
    """`--host_environments true` under torchrun with two ranks: the host-plugin loop (PAACLearner._train_host) advances
    global_step by the environments of ALL ranks per step, like the device loop (paac.py:127 counts every environment of
    the one learner) -- it stops at max_global_steps on the global count (3 cycles here, not 6) and anneals lr on it."""
    import json
    import subprocess
    folder = tempfile.mkdtemp(prefix="paac_dp_host_")
    env = dict(os.environ, PAAC_DIST_BACKEND="gloo", PAAC_DIST_SINGLE_DEVICE="1")
    G, N, T, cycles = 2, 4, 5, 3
    steps = G * N * T * cycles
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(G), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "-m", "paac_amd.train", "-g", "breakout", "--arch", "NIPS",
           "-ec", str(N), "-ew", "2", "--host_environments", "true", "--max_global_steps", str(steps),
           "-lra", str(4 * steps), "-df", folder]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert os.listdir(os.path.join(folder, "checkpoints")) == ["-%d.npz" % steps]
    recs = [json.loads(l) for l in open(os.path.join(folder, "metrics.jsonl"))]
    assert all(r["global_step"] <= steps for r in recs if "global_step" in r)
    log = res.stdout + res.stderr
    assert log.count("Starting training (2 data-parallel ranks)") == 2
    # both ranks ran `cycles` updates (counting local environments only they would have run G times as many)
    assert log.count("Host-plugin loop: %d update cycles, global step %d" % (cycles, steps)) == 2, log[-3000:]
