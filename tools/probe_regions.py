"""GPU probe: the conv towers' region layouts at one batch (launch durations from the dispatch events).
PROBE_ARCH = 0 (NIPS) / 1 (NATURE), PROBE_B = rows."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paac_amd import _lib, hip_ops
arch = int(os.environ.get("PROBE_ARCH", "0")); B = int(os.environ.get("PROBE_B", "32")); A = 6 if arch == 0 else 4
lib = _lib.load()
for regions in ((9, 4, 1) if arch == 0 else (8, 4, 2, 1)):
    ctx = hip_ops.Context(arch, A, max_batch=B)
    for cls in (0, 1, 2):
        lib.paac_debug_set_tuning(ctx.handle, 11, cls, regions, 0, -1)
    P = torch.randn(ctx.layout["total"], device="cuda") * 0.02
    S = torch.randint(0, 255, (B, 84, 84, 4), dtype=torch.uint8, device="cuda")
    probs = torch.zeros(B, A, device="cuda")
    ctx.set_managed_weights(True); ctx.pack_weights(P)
    g = hip_ops.Graph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            ctx.forward(P, S, probs=probs)
        g.begin()
        for _ in range(10):
            ctx.forward(P, S, probs=probs)
        g.end()
        st.synchronize()
        import time
        for _ in range(5): g.launch()
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): g.launch()
        st.synchronize()
        dt = (time.perf_counter() - t0) / 500 * 1e6
    print("arch %d B %d regions %d: %.2f us per forward (tower + fc + heads finish, replayed)" % (arch, B, regions, dt), flush=True)
    g.close(); ctx.close()
