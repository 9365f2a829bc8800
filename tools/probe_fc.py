"""Diagnostic (stamped build: python -m paac_amd.build --stamps; PAAC_HIP_LIB=.../libpaac_hip_stamps.so): wave timelines of
the dmm contractions of one training forward (PROBE_FWD rows) and one backward (PROBE_BWD rows) -- one line per GemmArgs
the launcher built, in launch order."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from paac_amd import _lib, hip_ops

dev = torch.device("cuda", 0)
lib = _lib.load()
lib.paac_debug_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
BF, BB = int(os.environ.get("PROBE_FWD", "192")), int(os.environ.get("PROBE_BWD", "160"))
ctx = hip_ops.Context(1, 4, max_batch=BF)
ctx.set_managed_weights(True)
P = torch.randn(ctx.layout["total"], device=dev) * 0.02
ctx.pack_weights(P)
S = torch.randint(0, 255, (BF, 84, 84, 4), dtype=torch.uint8, device=dev)
grad = torch.zeros(ctx.layout["total"], device=dev)
acts = torch.zeros(BB, dtype=torch.int32, device=dev)
yy = torch.randn(BB, device=dev)
aa = torch.randn(BB, device=dev)
stamps = torch.zeros(8 * 8 * 70000, dtype=torch.int64, device=dev)
names = ["invariants", "prologue loads issue", "main loop", "lds reduce", "epilogue"]


def cycle():
    ctx.train_forward_trunk(P, S)
    ctx.loss_backward(P, S[:BB], acts, yy, aa, 0.02, grad, forward_done=True)


for _ in range(10):
    cycle()
torch.cuda.synchronize()
for which in range(12):
    stamps.zero_()
    lib.paac_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()), which)
    cycle()
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(-1, 8)
    st = st[st[:, 0] != 0]
    if len(st) == 0:
        continue
    w0, w7 = st[:, 0].astype(np.float64), st[:, 7].astype(np.float64)
    print("call %2d  waves %5d | wave start spread %.2f us | wave lifetime med %.2f max %.2f us | first start -> last end %.2f us" % (
        which, len(st), (w0.max() - w0.min()) / 100, np.median(w7 - w0) / 100, (w7 - w0).max() / 100,
        (w7.max() - w0.min()) / 100))
    seg = np.diff(st[:, 1:7].astype(np.float64), axis=1)
    print("    " + " | ".join("%s %d" % (n, np.median(seg[:, i])) for i, n in enumerate(names)) + "  (median cycles)")
lib.paac_debug_set_stamps(None, -1)
