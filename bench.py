#!/usr/bin/env python3
"""bench.py -- global env-steps/s of the PAAC hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one full PAAC cycle (paac.py:99-183): T x [policy forward -> sample -> env step] -> bootstrap
forward -> n-step returns -> forward/backward on the N*T batch -> (RCCL gradient all-reduce) -> clip +
RMSProp, on synthetic 84x84x4 uint8 frames generated on the device; value = env-steps of all ranks / the
slowest rank's wall time (weak scaling: 32 envs per GPU).  Workload at N=1 = BASELINE configs[1]
(Breakout action set, Nature net, 32 envs, t_max=5).

The timed region is repeated: consecutive windows of exactly K steps, each bracketed by barrier + synchronize on both
sides and reduced with MAX over ranks; `value` / `ms_per_step` are the MEDIAN window's, all windows are listed in
`windows_ms`.  `--windows` fixes their number; by default there are at least 9 and as many more as it takes to time a
quarter of a second (at most 51): a 20-step window is 4.5 ms and the GPU clock is still ramping through the first three
or four of them, so with nine short windows the median itself sat on the ramp (707 k against 719 k from 300-step windows
on the same box).

Extra objects on the JSON line:
  roofline     -- the kernel family with the largest share of the cycle, timed with HIP events attached to the
                  kernel dispatches (hipExtLaunchKernelGGL start/stop events on the launch stream) in a second,
                  eager (graph-free) pass over the same K steps; achieved = algorithmic FLOPs (or bytes) per
                  launch / average launch duration; traffic = PMC bytes per launch from profiles/ (separate passes).
  cpu_baseline -- oracle/cpu_learner.py (torch-CPU port of the reference loop; the reference's TF path cannot
                  run here) on a bounded sample, rank 0, N=1 only: thread count calibrated once, then the median of
                  three equal windows.
  host_plugin_loop (only with --host-envs) -- the same workload with the environments as host BaseEnvironment
                  plugins (PAACLearner._train_host: per step 32 x 28 KB observations H2D + the action indices D2H,
                  eager launches): the PCIe-inclusive rate a real emulator would see.  Reported beside `value`, never as it.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16; conv1 spends 3 exact bf16 products per fp32 multiply, split-bf16 ops 6
PEAK_HBM_GBS = 8000.0


def mfma_ceiling(bodies, mix):
    """TFLOP/s ceiling (in algorithmic fp32-equivalent FLOPs) of a launch whose contraction bodies `bodies` (FLOPs each) ran
    the instruction mixes `mix` (MFMA products issued per fp32 multiply, recorded by the library at launch time): 1 = fp32
    MFMA (157.3 TFLOP/s), 3 / 6 = exact / split bf16 on the dense-bf16 MFMA (2500 / products).  Harmonic over the bodies."""
    if not bodies or len(mix) != len(bodies) or sum(bodies) <= 0:
        return None
    t = sum(f / (PEAK_FP32_MFMA_TFLOPS if p <= 1 else PEAK_BF16_MFMA_TFLOPS / p) for f, p in zip(bodies, mix))
    return sum(bodies) / t


def family_work(name, batch, arch, A, P, raw=False):
    """Algorithmic work of one launch of a kernel family: (kind, amount) with kind 'flop' or 'byte'; for 'flop' the amount
    is the list of its contraction bodies' FLOPs in launch order (the library names paired launches for what they ran)."""
    if arch == "NATURE":
        C1, C2, C3, H, FLAT = 32, 64, 64, 512, 3136
        conv3 = 2.0 * batch * 49 * 576 * 64
    else:
        C1, C2, C3, H, FLAT = 16, 32, 32, 256, 2592
        conv3 = 0.0
    conv1 = 2.0 * batch * 400 * 256 * C1
    conv2 = 2.0 * batch * 81 * (16 * C1) * C2
    fc = 2.0 * batch * FLAT * H
    table = {"conv1_fwd": [conv1], "conv1_wgrad": [conv1], "conv2_fwd": [conv2], "conv2_wgrad": [conv2],
             "conv2_dgrad": [conv2], "conv3_fwd": [conv3], "conv3_wgrad": [conv3], "conv3_dgrad": [conv3], "fc_fwd": [fc],
             "fc_wgrad": [fc], "fc_dgrad": [fc],
             "conv_tower": [conv1, conv2 + conv3],       # the three conv layers in one launch (csrc/tower.h)
             "dgrad_tower": [conv3 + conv2],             # conv3 AND conv2 data gradients in one launch (csrc/dgrad_tower.h)
             "fc_conv3_wgrad": [fc, conv3],              # dmm_pair_kernel: what the launcher paired (net_bwd.hip)
             "conv2_conv1_wgrad": [conv2, conv1]}
    if name in table:
        return "flop", table[name]
    if name == "clip_rmsprop":      # read g (norm) + read g, ms, var + write ms, mom, var (momentum 0: slot not read)
        return "byte", 4.0 * P * 7  # (+ the packed copies of the conv / fc weights it leaves behind: 4 P more bytes written)
    if name == "heads_fwd":
        return "byte", 4.0 * batch * H * 2 + 4.0 * H * (A + 1)
    if name == "heads_bwd":
        return "byte", 4.0 * batch * H * 3 + 4.0 * H * (A + 1) * 2
    # kernels around the network: batch = environments per launch (nstep_returns: N*T elements)
    OBS, RAW = 28224.0, 2 * 33600.0
    if name in ("env_step", "sample_env_step"):      # read the stacks, write the shifted stacks (+ the sampler's [N,A] probs)
        if raw:                                       # path B: the step WRITES the two raw 210x160 screens
            return "byte", batch * (RAW + (4.0 * A if name == "sample_env_step" else 0.0) + 24.0)
        return "byte", batch * (2 * OBS + (4.0 * A if name == "sample_env_step" else 0.0) + 24.0)
    if name == "preprocess_stack":
        # SURVEY 8(d), row-granular gather: the 84 source rows (of 210) the nearest resize keeps, 160 B each, of both
        # screens; the stack in and the stack out
        return "byte", batch * (2 * 84 * 160.0 + 2 * OBS)
    if name in ("sample_mt", "sample_philox"):
        return "byte", batch * (4.0 * A + 4.0) + (2496.0 * 2 if name == "sample_mt" else 0.0)
    if name == "nstep_returns":                       # rewards, masks, values in; y, adv out
        return "byte", batch * 20.0
    return "byte", 0.0


def time_host_plugin_loop(a, train, PAACLearner, np, cycles=60, warm=10):
    """PAACLearner._train_host on SyntheticEnvironment host plugins, same workload: env-steps/s over the cycles after
    `warm` (per step: N x 28 KB observations H2D, one sync'ing D2H of the action indices, eager kernel launches)."""
    args = train.get_arg_parser().parse_args([])
    args.game, args.arch = a.game, a.arch
    args.emulator_counts, args.max_local_steps, args.emulator_workers = a.envs, a.tmax, 8    # the reference's default: 8 workers
    args.host_environments, args.metrics = True, False
    args.max_global_steps = cycles * a.envs * a.tmax
    args.debugging_folder = tempfile.mkdtemp(prefix="paac_bench_host_")
    stamps = []
    args.cycle_callback = lambda step: stamps.append(time.perf_counter())
    network_creator, env_creator = train.get_network_and_environment_creator(args)
    learner = PAACLearner(network_creator, env_creator, args)
    learner.network.init = lambda folder, saver, session: (learner.network.initialize(np.random.RandomState(0)), 0)[1]
    np.random.seed(42)
    learner.train()
    dt = stamps[-1] - stamps[warm - 1]
    return dict(value=round((cycles - warm) * a.envs * a.tmax / dt, 1), unit="env-steps/s", cycles=cycles - warm,
                note="host BaseEnvironment plugins (synthetic, numpy) stepped by 8 worker processes through shared memory "
                     "like the reference's runners, observations H2D (page-locked shared array) and actions D2H every step, "
                     "eager launches; PCIe-inclusive, not `value`")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=32, help="environments per GPU")
    ap.add_argument("--tmax", type=int, default=5)
    ap.add_argument("--arch", default="NATURE")
    ap.add_argument("--game", default="breakout")
    ap.add_argument("--sampler", default="numpy", choices=["numpy", "philox"])
    ap.add_argument("--raw-frames", action="store_true", help="path B: raw 210x160 frame pairs + GPU preprocess")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--windows", type=int, default=0,
                    help="consecutive K-step windows, the median one is reported; 0 = at least 9, more while they add up to "
                         "less than 0.25 s (at most 51)")
    ap.add_argument("--host-envs", action="store_true",
                    help="also time the host-plugin loop (PCIe-inclusive) on the same workload; reported beside value")
    a = ap.parse_args()

    # stdout carries ONE JSON line: whatever libraries print on the way (RCCL 2.26 writes a version banner to stdout when
    # its communicator comes up) goes to stderr -- file descriptor 1 points at stderr until the line is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    # Rehearsal knob for a one-GPU box: PAAC_BENCH_REHEARSAL=1 puts every rank on cuda:0 and exchanges gradients
    # over gloo, so the N>1 code path (barriers, phased all-reduce, max over ranks) can be exercised without N GPUs.
    # Never set by the driver; the numbers of such a run are meaningless.
    rehearsal = os.environ.get("PAAC_BENCH_REHEARSAL", "") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif os.environ.get("PAAC_DIST_FORCE", "") == "1":
        # one-GPU rehearsal of the data-parallel cycle: an RCCL group of ONE rank; with PAAC_FORCE_COLLECTIVES=1 the phased
        # graphs and the stream-ordered all-reduce calls really run (what they cost besides the wire time)
        from paac_amd import parallel
        parallel.init_from_env()
        import torch.distributed as dist

    from paac_amd import train
    from paac_amd.paac import DeviceRollout, PAACLearner

    args = train.get_arg_parser().parse_args([])
    args.game, args.arch = a.game, a.arch
    args.emulator_counts, args.max_local_steps = a.envs, a.tmax
    args.emulator_workers = 0
    args.device = "/gpu:%d" % local_rank
    args.max_global_steps = 1 << 60
    args.synthetic_raw_frames = bool(a.raw_frames)
    args.debugging_folder = tempfile.mkdtemp(prefix="paac_bench_")
    network_creator, env_creator = train.get_network_and_environment_creator(args)
    learner = PAACLearner(network_creator, env_creator, args)
    learner.network.initialize(np.random.RandomState(0))          # random-init weights (no checkpoints offline)
    np.random.seed(42 + rank)
    N, T, A = a.envs, a.tmax, args.num_actions
    ro = DeviceRollout(learner, env_creator.device_env_spec, sampler=a.sampler, sampler_seed=42, env_offset=rank * N,
                       use_graph=not a.no_graph)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if ro.use_graph:
        with torch.cuda.stream(ro.stream):
            ro.capture()                # recording the graphs executes nothing: kept out of the timed region even at W = 0
    ro.run_cycles(a.warmup)
    ro.synchronize()
    # data parallel: the replicas must hold bit-identical weights and optimizer slots -- compared (checksums, MIN / MAX over
    # ranks) after the warm-up, where the exchanged gradient of the last warm-up update is compared too, and again after the
    # timed windows; a mismatch is reported in the line and ends the run non-zero on every rank
    replicas_identical = None
    replica_error = None

    def check_replicas(with_grad):
        nonlocal replicas_identical, replica_error
        if not ro.phased or replica_error is not None:
            return
        from paac_amd import parallel
        try:
            if with_grad and a.warmup > 0:
                ro.check_replicas("grad")
            ro.check_replicas("weights")
            replicas_identical = True
        except parallel.ReplicaMismatch as exc:
            replicas_identical = False
            replica_error = str(exc)
            print("bench.py: %s" % exc, file=sys.stderr, flush=True)

    check_replicas(True)
    windows = []
    while True:
        if a.windows > 0 and len(windows) >= a.windows:
            break
        if a.windows <= 0 and len(windows) >= 9 and (sum(windows) >= 0.25 or len(windows) >= 51):
            break                       # (the window times are already MAX-reduced: every rank stops at the same count)
        barrier()
        t0 = time.perf_counter()
        ro.run_cycles(a.steps)          # exactly K cycles (graph replay batches them 4 per launch where it can)
        ro.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        windows.append(dt)
    check_replicas(False)
    elapsed = sorted(windows)[len(windows) // 2]          # the median window
    value = world * N * T * a.steps / elapsed
    finite = bool(torch.isfinite(learner.network.params).all().item())

    roofline = None
    kernels = None
    per = {}
    if not a.no_roofline:
        # every rank runs the event-timed pass (with N > 1 each cycle contains collectives); rank 0 reports
        learner.ctx.prof_enable(True)
        ro.use_graph = False
        steps_done = 0
        # Eager launches are host-bound (~7 us per ctypes call): keep the GPU busy with a ~1 ms memset train while
        # the host enqueues the next cycle, so the kernels then run back to back and each event pair brackets GPU
        # execution only (otherwise the start event waits for the host and inflates the small kernels).
        blocker = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        while steps_done < a.steps:
            chunk = min(a.steps - steps_done, 32)      # 8192-launch event table
            for c in range(chunk):
                with torch.cuda.stream(ro.stream):
                    for r in range(4):
                        blocker.fill_((c + r) & 255)
                ro.run_cycle()
            ro.synchronize()
            for name, batch, ms, mix in learner.ctx.prof_read(with_mix=True):
                d = per.setdefault((name, batch), [0.0, 0, mix])
                d[0] += ms
                d[1] += 1
            steps_done += chunk
        learner.ctx.prof_enable(False)
        ro.use_graph = not a.no_graph
        del blocker
    if rank == 0 and per:
        P = learner.network.layout["total_unpadded"]
        arch = "NIPS" if a.arch == "NIPS" else "NATURE"
        kernels = []
        for (name, batch), (ms, cnt, mix) in per.items():
            kind, amount = family_work(name, batch, arch, A, P, raw=a.raw_frames)
            bodies = amount if kind == "flop" else None
            amount = sum(bodies) if bodies else amount
            avg_us = 1000.0 * ms / cnt
            ach = amount / (avg_us * 1e-6) if avg_us > 0 else 0.0
            k = dict(kernel=name, batch=batch, launches_per_step=cnt / a.steps, avg_us=round(avg_us, 3),
                     us_per_step=round(1000.0 * ms / a.steps, 2),
                     achieved=(round(ach / 1e12, 3) if kind == "flop" else round(ach / 1e9, 1)) if amount > 0 else None,
                     unit="TFLOP/s" if kind == "flop" else "GB/s")
            if k["achieved"] is not None:
                # the ceiling of what the family ran: its recorded instruction mix on the MFMA, HBM otherwise
                peak = mfma_ceiling(bodies, mix) if kind == "flop" else PEAK_HBM_GBS
                if peak:
                    k["peak"] = round(peak, 1)
                    k["frac"] = round(k["achieved"] / peak, 4)
                if kind == "flop":
                    k["mfma_products_per_multiply"] = list(mix)
            kernels.append(k)
        kernels.sort(key=lambda k: -k["us_per_step"])
        dom = [k for k in kernels if k["achieved"] is not None][0]
        # HBM-side traffic of the dominant kernel from the committed PMC passes (collected separately, as rocprofv3
        # requires; profiles/*_traffic_by_family.json) when they were taken on this workload
        traffic = None
        try:
            import glob
            cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic_by_family.json")))
            key = "%s[batch=%d]" % (dom["kernel"], dom["batch"])
            want = "%s action set (A=%d), %s net, %d envs per GPU x 1 GPU, t_max=%d, %s" % (
                a.game, A, a.arch, N, T, "raw 210x160 frame pairs + GPU max/resize/stack" if a.raw_frames else
                "synthetic 84x84x4 u8 frames generated on device")
            for path in reversed(cand):               # the latest round's table taken on this workload
                tj = json.load(open(path))
                if tj.get("workload", "").split(", sampler")[0] == want and key in tj["bytes_per_launch"]:
                    traffic = tj["bytes_per_launch"][key]
                    break
        except Exception:
            traffic = None
        if dom["unit"] == "TFLOP/s":
            # frac: against the ceiling of the instruction mix the family ran (cannot exceed 1); frac_fp32_contract: the
            # same achieved figure against the fp32-MFMA peak SURVEY 8(d) prices the network at (a bf16-split family can
            # exceed 1 there: it is a different instruction)
            roofline = dict(bound="mfma", kernel="%s[batch=%d]" % (dom["kernel"], dom["batch"]), achieved=dom["achieved"],
                            peak=dom.get("peak", PEAK_FP32_MFMA_TFLOPS), unit="TFLOP/s",
                            frac=dom.get("frac", round(dom["achieved"] / PEAK_FP32_MFMA_TFLOPS, 4)),
                            frac_fp32_contract=round(dom["achieved"] / PEAK_FP32_MFMA_TFLOPS, 4),
                            peak_fp32_contract=PEAK_FP32_MFMA_TFLOPS,
                            mfma_products_per_multiply=dom.get("mfma_products_per_multiply"),
                            avg_launch_us=dom["avg_us"], traffic=traffic)
            if dom["kernel"] == "conv_tower":
                roofline["note"] = ("achieved = algorithmic fp32-equivalent FLOP/s of conv1+conv2+conv3 in one launch; peak = "
                                    "dense bf16 MFMA / products per multiply (3 in conv1, 6 in conv2/conv3), harmonic over "
                                    "the layers' FLOPs; with several regions per sample part of the conv1/conv2 arithmetic "
                                    "is recomputed and every workgroup streams all 466 KB of pre-split conv weights from L2")
        else:
            roofline = dict(bound="hbm", kernel="%s[batch=%d]" % (dom["kernel"], dom["batch"]), achieved=dom["achieved"],
                            peak=PEAK_HBM_GBS, unit="GB/s", frac=round(dom["achieved"] / PEAK_HBM_GBS, 4),
                            avg_launch_us=dom["avg_us"], traffic=traffic)

    cpu_baseline = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import cpu_learner, network as onet          # the oracle is the baseline leg, never the product
        envs = [env_creator.create_environment(i) for i in range(N)]
        params = onet.init_params(a.arch, A, np.random.RandomState(0), dtype=np.float32)
        res = cpu_learner.run(envs, a.arch, A, T, params, min_seconds=a.cpu_seconds)
        cpu_baseline = dict(value=round(res["steps_per_s"], 1), unit="env-steps/s", cores=res["cores"],
                            threads=res["cores"], host_cores=os.cpu_count(), kind="port",
                            sample="%d cycles of the same workload (%d envs x t_max %d, %s net) in %.1f s, torch-CPU fp32 port "
                                   "of the reference loop; median of %d windows (%s env-steps/s), %d threads fixed after calibration"
                                   % (res["cycles"], N, T, a.arch, res["seconds"], res["windows"],
                                      "/".join("%.0f" % r for r in res["window_rates"]), res["cores"]))

    host_loop = None
    if rank == 0 and world == 1 and a.host_envs:
        host_loop = time_host_plugin_loop(a, train, PAACLearner, np)

    if rank == 0:
        out = {
            "metric": "global env-steps/sec at 32 envs, t_max=5, Nature net; 1/2/4/8 MI355X",
            "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1000.0 * elapsed / a.steps, 4),
            "windows_ms": [round(1000.0 * w, 3) for w in windows], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s action set (A=%d), %s net, %d envs per GPU x %d GPU, t_max=%d, %s, sampler=%s, %s"
                                   % (a.game, A, a.arch, N, world, T,
                                      "raw 210x160 frame pairs + GPU max/resize/stack" if a.raw_frames else
                                      "synthetic 84x84x4 u8 frames generated on device",
                                      a.sampler, "hipGraph replay" if not a.no_graph else "eager launches"),
                       "envs_per_gpu": N, "t_max": T, "global_envs": N * world,
                       "arithmetic": "fp32 results everywhere (parity: logits/values within 1e-4). The Nature conv layers "
                                     "(forward tower, conv3/conv2 data gradients) and most >64-row contractions run on the "
                                     "bf16 MFMA with each fp32 operand split EXACTLY into 3 bf16 terms (u8 pixels are exact "
                                     "in one), fp32 accumulation; of the 9 partial products the 3 below 2^-23 of the leading "
                                     "one are dropped.  The acting fc layer (<= 64 rows) runs on the fp32 MFMA",
                       "parallelism": "env-sharded dp%d, %s" % (world, (
                           "no collective (one process)" if world == 1 and not ro.phased else
                           "ONE RCCL sum all-reduce of the flat gradient per update, after the full backward, captured into the "
                           "cycle's hipGraph" if ro.graph_exchange else
                           "ONE RCCL sum all-reduce of the flat gradient per update, after the full backward, issued eagerly on "
                           "the rollout stream" if ro.single_exchange else
                           "RCCL sum all-reduce of the flat gradient per update in two pieces (PAAC_ALLREDUCE=split: the fc/heads "
                           "tail overlaps the conv backward)"))},
            "finite_params": finite,
            "replicas_identical": replicas_identical,       # null with one process (nothing to compare)
            "exchange": {"mode": ro.exchange_mode, "requested": os.environ.get("PAAC_ALLREDUCE", "graph"),
                         "fallback_reason": ro.exchange_fallback, "replica_error": replica_error},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "host_plugin_loop": host_loop, "kernels": kernels,
        }
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)           # C stdio buffers of the libraries, while fd 1 is still stderr
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    ro.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if replica_error is not None:
        sys.exit(3)


if __name__ == "__main__":
    main()
