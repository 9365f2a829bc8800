"""Diagnostic (stamped build, python -m paac_amd.build --stamps; PAAC_HIP_LIB=.../libpaac_hip_stamps.so): where the waves
of the fused conv tower spend their cycles.  PROBE_B = batch, PROBE_REGIONS = 1 / 2 / 4 / -1."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from paac_amd import _lib, hip_ops

lib = _lib.load()
lib.paac_debug_set_tower_stamps.argtypes = [ctypes.c_void_p]
B = int(os.environ.get("PROBE_B", "32"))
regions = int(os.environ.get("PROBE_REGIONS", "-1"))
ctx = hip_ops.Context(1, 4, max_batch=B)
for cls in (0, 1, 2):
    lib.paac_debug_set_tuning(ctx.handle, 11, cls, regions, 0, -1)
P = torch.randn(ctx.layout["total"], device="cuda") * 0.02
S = torch.randint(0, 255, (B, 84, 84, 4), dtype=torch.uint8, device="cuda")
probs = torch.zeros(B, 4, device="cuda")
ctx.set_managed_weights(True)
ctx.pack_weights(P)
for _ in range(20):
    ctx.forward(P, S, probs=probs)
torch.cuda.synchronize()
stamps = torch.zeros(12 * 8 * 9 * B + 1024, dtype=torch.int64, device="cuda")
lib.paac_debug_set_tower_stamps(ctypes.c_void_p(stamps.data_ptr()))
keep = os.environ.get("PROBE_KEEP", "0") == "1"
for rep in range(3):
    stamps.zero_()
    if keep:
        ctx.keep_next_forward(0)
    ctx.forward(P, S, probs=probs)
    torch.cuda.synchronize()
    st = stamps.cpu().numpy()[:-1024].reshape(-1, 12)
    st = st[st[:, 0] != 0].astype(np.float64)
    w0, w1 = st[:, 0], st[:, 11]
    print("waves %d | start spread %.2f us | lifetime med %.2f max %.2f us | first start -> last end %.2f us" % (
        len(st), (w0.max() - w0.min()) / 100, np.median(w1 - w0) / 100, (w1 - w0).max() / 100, (w1.max() - w0.min()) / 100))
    names = ["stage input", "barrier", "conv1 gemm", "conv1 epilogue", "barrier", "conv2 gemm", "conv2 reduce+epilogue+barrier",
             "conv3 gemm", "conv3 reduce+epilogue"]
    seg = np.diff(st[:, 1:11], axis=1)
    print("    " + " | ".join("%s %d" % (n, np.median(seg[:, i])) for i, n in enumerate(names)) + "  (median cycles; max: " +
          " ".join("%d" % seg[:, i].max() for i in range(seg.shape[1])) + ")")
lib.paac_debug_set_tower_stamps(None)
