"""Diagnostic (stamped build only): phases of the sampler workgroup of the fused sampler + env-step launch.
PAAC_HIP_LIB=.../libpaac_hip_stamps.so python tools/probe_sampler.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from paac_amd import hip_ops, _lib
from paac_amd.synthetic import terminal_threshold
lib = _lib.load()
lib.paac_debug_set_misc_stamps.argtypes = [ctypes.c_void_p]
N, A = int(os.environ.get("PROBE_N", "32")), int(os.environ.get("PROBE_A", "4"))
dev = torch.device("cuda", 0)
stamps = torch.zeros(16, dtype=torch.int64, device=dev)
lib.paac_debug_set_misc_stamps(ctypes.c_void_p(stamps.data_ptr()))
probs = torch.softmax(torch.randn(N, A, device=dev), dim=1)
mt = hip_ops.mt_state_from_numpy(np.random.RandomState(1).get_state(), dev)
act = torch.zeros(N, dtype=torch.int32, device=dev)
s0 = torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device=dev); s1 = torch.zeros_like(s0)
rew = torch.zeros(N, device=dev); msk = torch.zeros(N, device=dev); epr = torch.zeros(N, device=dev)
epl = torch.zeros(N, dtype=torch.int32, device=dev); fin = torch.zeros(hip_ops.FINISHED_RING_BYTES // 4, dtype=torch.int32, device=dev)
tick = torch.zeros(1, dtype=torch.int64, device=dev)
walk = hip_ops.walk_scratch(N, A, dev) if os.environ.get("PROBE_MULTI", "") == "1" else None
# PROBE_ACT=1: through paac_act_step_mt (the large shards' sampler then finds its doubles made ahead, csrc/mt_ahead.h)
ctx = None
if os.environ.get("PROBE_ACT", "") == "1":
    ctx = hip_ops.Context(1, A, max_batch=N)
    params = torch.randn(ctx.layout["total"], device=dev) * 0.02
    ctx.set_managed_weights(os.environ.get("PROBE_MANAGED", "1") == "1")
    ctx.pack_weights(params)
    val = torch.zeros(N, device=dev)
    pr_out = torch.zeros((N, A), device=dev)
names = ["entry->loads landed", "phase 1 (cond. probabilities)", "phase 2 (state blocks)", "phase 3 (doubles)",
         "table fill", "chase", "write-back", "bookkeeping"]
acc = np.zeros(8)
reps = 50
fam = {}
if ctx is not None:
    ctx.prof_enable(True)
# PROBE_EVICT=1: 96 MB written and a few other kernels run between the steps -- what the first step of a cycle finds after
# the update (its L2 lines and instruction cache lines gone)
evict = torch.zeros(24 * 1024 * 1024, device=dev) if os.environ.get("PROBE_EVICT", "") == "1" else None
for r in range(reps + 5):
    probs = torch.softmax(torch.randn(N, A, device=dev), dim=1)    # cold-ish probabilities each time
    if evict is not None:
        evict.add_(1.0)
        (evict[:1 << 20].view(1024, 1024) @ evict[1 << 20:2 << 20].view(1024, 1024)).sum()
    torch.cuda.synchronize()
    if ctx is not None:
        ctx.act_step_mt(params, s0, mt, act, pr_out, val, 3, 0, terminal_threshold(0.01), tick, 0, s1, rew, msk, epr, epl, fin,
                        walk_scratch=walk)
    else:
        hip_ops.sample_mt_synth_step(probs, mt, act, 3, 0, terminal_threshold(0.01), tick, 0, s0, s1, rew, msk, epr, epl, fin,
                                     walk_scratch=walk)
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().astype(np.float64)
    if ctx is not None and r >= 5:
        for name, batch, ms in ctx.prof_read():
            fam[name] = fam.get(name, 0.0) + ms * 1000.0 / reps
    if r >= 5:
        acc += np.diff(st[:9])
for n, v in zip(names, acc / reps):
    print("%-32s %8.0f cycles" % (n, v))
print("%-32s %8.0f cycles" % ("total (entry -> end)", acc.sum() / reps))
print("launch durations (HIP events on the dispatches), us:", {k: round(v, 2) for k, v in fam.items()})
if walk is not None:      # stamped build: wall-clock ticks (10 ns) of the workgroup that took the last ticket
    w = st[[0, 1, 2, 3, 4, 9, 10, 11, 12, 6, 7, 8]]
    print("  p1 split: scan + ranges %d | phase 1 proper %d (10 ns ticks)" % (st[13] - st[1], st[2] - st[13]))
    print("multi (last call, 10 ns ticks): loads %d | p1 %d | p2 %d | p3 %d | walks %d | ticket %d | fence+exits %d | chase %d "
          "| actions %d | writeback %d | bookkeeping %d" % tuple(np.diff(w)))
