"""Diagnostic: activation error statistics of the fused conv tower vs the per-layer kernels against the float64 oracle,
and timing of the forward at several batch sizes.  python tools/probe_tower.py [batch ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import network as onet
from paac_amd import _lib, hip_ops

A = 4


def run(B, tower, regions=-1, check=True):
    os.environ["PAAC_TOWER"] = "1" if tower else "0"
    rs = np.random.RandomState(1)
    params = onet.init_params("NATURE", A, rs, dtype=np.float32)
    states = rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)
    ctx = hip_ops.Context(1, A, max_batch=B)
    for cls in (0, 1, 2):
        _lib.check(ctx.lib.paac_debug_set_tuning(ctx.handle, 11, cls, regions, 0, -1), "tune")
    flat = np.zeros(ctx.layout["total"], dtype=np.float32)
    for t in ctx.layout["tensors"]:
        flat[t["offset"]:t["offset"] + t["size"]] = params[t["name"]].reshape(-1)
    p = torch.from_numpy(flat).cuda()
    s = torch.from_numpy(states).cuda()
    logits = torch.zeros((B, A), device="cuda")
    ctx.forward(p, s, logits=logits)
    torch.cuda.synchronize()
    out = {}
    if check:
        ref = onet.forward(params, states, "NATURE", dtype=np.float64, keep=True)
        for i in (1, 2, 3):
            got = ctx.debug_activation(i, B).cpu().numpy().astype(np.float64)
            want = ref["cache"]["a%d" % i].reshape(-1)
            e = np.abs(got - want)
            flips = int(((got > 0) != (want > 0)).sum())
            out["a%d" % i] = "max %.2e rms %.2e flips %d/%d" % (e.max(), np.sqrt((e ** 2).mean()), flips, e.size)
        out["logits"] = "max %.2e" % np.abs(logits.cpu().numpy() - ref["logits"]).max()
    # timing: managed mode (no per-call pack), graph of 20 forwards
    ctx.set_managed_weights(True)
    ctx.pack_weights(p)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            ctx.forward(p, s, logits=logits)
        g = hip_ops.Graph()
        g.begin()
        for _ in range(20):
            ctx.forward(p, s, logits=logits)
        g.end()
        g.launch()
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.launch()
        st.synchronize()
        out["us_per_forward"] = round((time.perf_counter() - t0) / 200 * 1e6, 2)
        g.close()
    ctx.close()
    return out


if __name__ == "__main__":
    batches = [int(x) for x in sys.argv[1:]] or [32, 160]
    for B in batches:
        print("B=%d per-layer :" % B, run(B, False, check=B <= 256), flush=True)
        for regions in (8, 4, 2, 1):
            if B * regions > 4096:
                continue
            print("B=%d tower r=%d :" % (B, regions), run(B, True, regions, check=B <= 256), flush=True)
