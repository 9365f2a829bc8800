"""Replay forward(B=32) and forward+backward(B=160) graphs; meant to be run under rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paac_amd import hip_ops
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream()
A = int(os.environ.get("PROBE_A", "4"))
ctx = hip_ops.Context(1, A, max_batch=160)
P = torch.randn(ctx.layout["total"], device=dev) * 0.02
S = torch.randint(0, 255, (160, 84, 84, 4), dtype=torch.uint8, device=dev)
probs = torch.zeros(32, A, device=dev); vals = torch.zeros(32, device=dev)
grad = torch.zeros(ctx.layout["total"], device=dev)
acts = torch.zeros(160, dtype=torch.int32, device=dev)
yy = torch.randn(160, device=dev); aa = torch.randn(160, device=dev)
with torch.cuda.stream(stream):
    g = hip_ops.Graph(); g.begin(); ctx.forward(P, S[:32], probs=probs, values=vals); g.end()
    g2 = hip_ops.Graph(); g2.begin(); ctx.loss_backward(P, S, acts, yy, aa, 0.02, grad); g2.end()
    for _ in range(100):
        g.launch()
    for _ in range(100):
        g2.launch()
    stream.synchronize()
