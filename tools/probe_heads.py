import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from paac_amd import hip_ops, _lib
dev = torch.device("cuda", 0)
lib = _lib.load()
lib.paac_debug_set_heads_stamps.argtypes = [ctypes.c_void_p]
ctx = hip_ops.Context(1, 4, max_batch=160)
P = torch.randn(ctx.layout["total"], device=dev) * 0.02
S = torch.randint(0, 255, (160, 84, 84, 4), dtype=torch.uint8, device=dev)
st = torch.zeros(8 * 4096, dtype=torch.int64, device=dev)
probs = torch.zeros(32, 4, device=dev); vals = torch.zeros(32, device=dev); acts = torch.zeros(32, dtype=torch.int32, device=dev)
tick = torch.zeros(1, dtype=torch.int64, device=dev)
for _ in range(10):
    ctx.forward_sample(P, S[:32], 42, tick, 0, 0, acts, probs=probs, values=vals)
torch.cuda.synchronize()
lib.paac_debug_set_heads_stamps(ctypes.c_void_p(st.data_ptr()))
st.zero_()
ctx.forward_sample(P, S[:32], 42, tick, 0, 0, acts, probs=probs, values=vals)
torch.cuda.synchronize()
a = st.cpu().numpy().reshape(-1, 8)[:32].astype(np.float64)
d = np.diff(a[:, :5], axis=1)
print("heads_fwd B=32 (cycles, median over blocks): loads+dot %d | shuffle reduce %d | bias+sync %d | softmax+sample %d | total %d" % tuple(list(np.median(d, axis=0)) + [np.median(a[:, 4] - a[:, 0])]))
grad = torch.zeros(ctx.layout["total"], device=dev)
a32 = torch.zeros(160, dtype=torch.int32, device=dev); yy = torch.randn(160, device=dev); aa = torch.randn(160, device=dev)
for _ in range(3):
    ctx.loss_backward(P, S, a32, yy, aa, 0.02, grad)
st.zero_()
ctx.loss_backward(P, S, a32, yy, aa, 0.02, grad)
torch.cuda.synchronize()
a = st.cpu().numpy().reshape(-1, 8).astype(np.float64)
r1 = a[:160]; r2 = a[160:176]; r3 = a[176]
print("heads_bwd role1 (rows) median cycles %d" % np.median(r1[:, 1] - r1[:, 0]))
print("heads_bwd role2 (weights) accumulate %d | reduce+store %d" % (np.median(r2[:, 2] - r2[:, 0]), np.median(r2[:, 3] - r2[:, 2])))
print("heads_bwd role3 (bias+loss) %d" % (r3[4] - r3[0]))
