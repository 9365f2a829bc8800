"""Probe: per-kernel cost of small dependent launches inside the torch process (eager vs hipGraph)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from paac_amd import hip_ops

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
stream = torch.cuda.Stream()
ctr = torch.zeros(1, dtype=torch.int64, device=dev)
gs = torch.zeros(1, dtype=torch.int64, device=dev)
lr = torch.zeros(1, device=dev)
ctx = hip_ops.Context(1, 4, max_batch=160)
P = torch.randn(ctx.layout["total"], device=dev) * 0.01
S = torch.randint(0, 255, (160, 84, 84, 4), dtype=torch.uint8, device=dev)
probs = torch.zeros(32, 4, device=dev)
vals = torch.zeros(32, device=dev)


def timeit(name, fn, n_kernels, reps=200):
    with torch.cuda.stream(stream):
        for mode in ("eager", "graph"):
            if mode == "graph":
                g = hip_ops.Graph()
                g.begin(); fn(); g.end()
                run = g.launch
            else:
                run = fn
            for _ in range(10):
                run()
            stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            stream.synchronize()
            dt = (time.perf_counter() - t0) / reps * 1e6
            print("%-38s %-6s %8.2f us total  %6.2f us/kernel" % (name, mode, dt, dt / n_kernels), flush=True)


timeit("64x counter_add", lambda: [hip_ops.counter_add(ctr, 1) for _ in range(64)], 64)
timeit("32x (counter_add, lr_step)", lambda: [(hip_ops.counter_add(ctr, 1), hip_ops.lr_step(gs, 1, 0.1, 1000000000, lr)) for _ in range(32)], 64)
timeit("forward B=32 (5 kernels)", lambda: ctx.forward(P, S[:32], probs=probs, values=vals), 5)
timeit("4x forward B=32 (20 kernels)", lambda: [ctx.forward(P, S[:32], probs=probs, values=vals) for _ in range(4)], 20)
timeit("forward B=160 (5 kernels)", lambda: ctx.forward(P, S, None, None, None), 5)


# shader clock actually held while replaying the forward graph back to back
clk = torch.zeros((200, 2), dtype=torch.int64, device=dev)
with torch.cuda.stream(stream):
    g = hip_ops.Graph(); g.begin()
    for _ in range(4):
        ctx.forward(P, S[:32], probs=probs, values=vals)
    g.end()
    for i in range(200):
        g.launch()
        hip_ops.debug_clock(clk[i])
    stream.synchronize()
c = clk.cpu().numpy().astype(np.float64)
d = np.diff(c, axis=0)
mhz = d[:, 0] / d[:, 1] * 100.0
print("shader clock while replaying 4x forward graphs: median %.0f MHz (min %.0f, max %.0f); per-replay %.1f us"
      % (np.median(mhz[20:]), mhz[20:].min(), mhz[20:].max(), np.median(d[20:, 1]) / 100.0))
# and under a long dense kernel stream (backward at B=160)
grad = torch.zeros(ctx.layout["total"], device=dev)
acts = torch.zeros(160, dtype=torch.int32, device=dev)
yy = torch.randn(160, device=dev); aa = torch.randn(160, device=dev)
with torch.cuda.stream(stream):
    g2 = hip_ops.Graph(); g2.begin()
    ctx.loss_backward(P, S, acts, yy, aa, 0.02, grad)
    g2.end()
    for i in range(200):
        g2.launch()
        hip_ops.debug_clock(clk[i])
    stream.synchronize()
c = clk.cpu().numpy().astype(np.float64)
d = np.diff(c, axis=0)
mhz = d[:, 0] / d[:, 1] * 100.0
print("shader clock while replaying fwd+bwd B=160 graphs: median %.0f MHz; per-replay %.1f us" % (np.median(mhz[20:]), np.median(d[20:, 1]) / 100.0))
