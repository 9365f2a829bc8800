"""Diagnostic (stamped build only): where one dmm wave spends its cycles.  PAAC_HIP_LIB=.../libpaac_hip_stamps.so"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from paac_amd import hip_ops, _lib
dev = torch.device("cuda", 0)
lib = _lib.load()
lib.paac_debug_set_stamps.argtypes = [ctypes.c_void_p]
B = int(os.environ.get("PROBE_B", "32"))
ctx = hip_ops.Context(1, 4, max_batch=160)
P = torch.randn(ctx.layout["total"], device=dev) * 0.02
S = torch.randint(0, 255, (160, 84, 84, 4), dtype=torch.uint8, device=dev)
stamps = torch.zeros(8 * 8 * 70000, dtype=torch.int64, device=dev)
probs = torch.zeros(B, 4, device=dev); vals = torch.zeros(B, device=dev)
for _ in range(20):
    ctx.forward(P, S[:B], probs=probs, values=vals)
torch.cuda.synchronize()
lib.paac_debug_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["invariants", "prologue loads issue", "main loop", "lds reduce", "epilogue"]
grad = torch.zeros(ctx.layout["total"], device=dev)
acts = torch.zeros(B, dtype=torch.int32, device=dev)
yy = torch.randn(B, device=dev); aa = torch.randn(B, device=dev)
mode = os.environ.get("PROBE_MODE", "fwd")
labels = ["conv1_fwd", "conv2_fwd", "conv3_fwd", "fc_fwd", "fc_wgrad", "fc_dgrad", "conv3_wgrad", "conv3_dgrad",
          "conv2_wgrad", "conv2_dgrad", "conv1_wgrad"]
for which in range(4 if mode == "fwd" else 11):
    stamps.zero_()
    lib.paac_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()), which)
    if mode == "fwd":
        ctx.forward(P, S[:B], probs=probs, values=vals)
    else:
        ctx.loss_backward(P, S[:B], acts, yy, aa, 0.02, grad)
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(-1, 8)
    st = st[st[:, 0] != 0]
    w0, w7 = st[:, 0].astype(np.float64), st[:, 7].astype(np.float64)
    print("%-12s waves %5d | wave start spread %.2f us | wave lifetime med %.2f max %.2f us | first start -> last end %.2f us" % (
        labels[which], len(st), (w0.max() - w0.min()) / 100, np.median(w7 - w0) / 100, (w7 - w0).max() / 100,
        (w7.max() - w0.min()) / 100))
    seg = np.diff(st[:, 1:7].astype(np.float64), axis=1)
    print("    " + " | ".join("%s %d" % (n, np.median(seg[:, i])) for i, n in enumerate(names)) + "  (median cycles)")
