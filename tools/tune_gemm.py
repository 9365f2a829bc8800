"""Coordinate-descent sweep of the GEMM launch configurations on the benchmark shapes (MI355X).
Times the replayed forward (acting batch) and forward+backward (training batch) hipGraphs while changing one
op's (cfg, ksplit, xcd) at a time through paac_debug_set_tuning; prints the best table as C++ for
csrc/net_bwd.hip:default_tuning and as JSON (gpurun_out/tune.json)."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paac_amd import hip_ops, _lib

ARCH = int(os.environ.get("TUNE_ARCH", "1"))
A = int(os.environ.get("TUNE_A", "4"))
N = int(os.environ.get("TUNE_N", "32"))
T = int(os.environ.get("TUNE_T", "5"))
dev = torch.device("cuda", 0)
lib = _lib.load()
ctx = hip_ops.Context(ARCH, A, max_batch=N * (T + 1))
P = torch.randn(ctx.layout["total"], device=dev) * 0.02
S = torch.randint(0, 255, (N * (T + 1), 84, 84, 4), dtype=torch.uint8, device=dev)
probs = torch.zeros(N, A, device=dev); vals = torch.zeros(N, device=dev)
grad = torch.zeros(ctx.layout["total"], device=dev)
acts = torch.zeros(N * T, dtype=torch.int32, device=dev)
yy = torch.randn(N * T, device=dev); aa = torch.randn(N * T, device=dev)
stream = torch.cuda.Stream()
OPS = ["conv1_fwd", "conv2_fwd", "conv3_fwd", "fc_fwd", "fc_wgrad", "fc_dgrad", "conv3_wgrad", "conv3_dgrad",
       "conv2_wgrad", "conv2_dgrad", "conv1_wgrad"]
if ARCH == 0:
    OPS_ACTIVE = [0, 1, 3, 4, 5, 8, 9, 10]
else:
    OPS_ACTIVE = list(range(11))


def set_tune(op, cls, cfg, ks, xcd):
    _lib.check(lib.paac_debug_set_tuning(ctx.handle, op, cls, cfg, ks, xcd), "set_tuning")


REPS = int(os.environ.get("TUNE_REPS", "25"))
MARGIN = float(os.environ.get("TUNE_MARGIN", "0.3"))   # us a candidate must win by


def get_tune(op, cls):
    c, k, x = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.paac_debug_get_tuning(ctx.handle, op, cls, ctypes.byref(c), ctypes.byref(k), ctypes.byref(x)), "get_tuning")
    return (c.value, k.value, x.value)


def time_graph(fn, reps=40):
    with torch.cuda.stream(stream):
        g = hip_ops.Graph(); g.begin(); fn(); g.end()
        for _ in range(5):
            g.launch()
        stream.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                g.launch()
            stream.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps * 1e6)
        g.close()
    return best


def fwd_act():
    ctx.forward(P, S[:N], probs=probs, values=vals)


def train():     # what the device loop runs per update: forward over N*(T+1) rows, backward over N*T
    ctx.train_forward(P, S)
    ctx.loss_backward(P, S[:N * T], acts, yy, aa, 0.02, grad, forward_done=True)


def candidates(op, cls):
    fam = "fwd" if op <= 3 else ("dgrad" if op in (5, 7, 9) else "wgrad")
    out = []
    if fam == "fwd" and op != 3:
        out = [(c, 0, -1) for c in range(13)]
        if op == 0:
            out += [(100 + c, 0, -1) for c in range(13)]      # conv1 on the exact-bf16 path
        else:
            out += [(200 + c, 0, -1) for c in range(13)]      # six-product split-bf16 path
            out += [(300 + c, 0, -1) for c in (0, 1, 2, 3, 6)]  # half-width N tiles
    elif op == 3:
        for c in list(range(13)) + [200 + c for c in range(13)]:
            for ks in (1, 2, 4, 8):
                out.append((c, ks, -1))
                if ks == 8:
                    out.append((c, ks, 2))
    elif fam == "dgrad":
        out = [(c, 0, x) for c in list(range(12)) + [200 + c for c in range(12)] for x in (-1, 0, 1)]
    elif op == 4:
        out = [(c, 1, x) for c in list(range(9)) + [200 + c for c in range(9)] for x in (-1, 0)]
    else:
        cfgs = [0, 1, 2, 3, 6, 7, 8, 100, 101, 102, 103, 106, 107, 108] if op == 10 else list(range(9)) + [200 + c for c in range(9)]
        for c in cfgs:
            for ks in (8, 16, 32, 48, 64):
                out.append((c, ks, -1)); out.append((c, ks, 2))
    return out


best = {}
start = {(op, cls): get_tune(op, cls) for op in range(11) for cls in range(3)}
def batch_class(b):
    return 2 if b > 512 else (1 if b > 64 else 0)


# the library picks the table row by batch: acting batch N, training batches N*(T+1) / N*T (same class here)
for cls, fn, label in ((batch_class(N), fwd_act, "act B=%d" % N), (batch_class(N * T), train, "train B=%d" % (N * T))):
    if fn is train and batch_class(N * T) == batch_class(N):
        print("(acting and training batch share class %d: tuning the training step only would override it)" % cls)
    base = time_graph(fn)
    print("%s: graph with the library's table %.1f us" % (label, base), flush=True)
    ops = [o for o in OPS_ACTIVE if (o <= 3 or cls >= 1)]
    if os.environ.get("TUNE_OPS"):
        ops = [o for o in ops if str(o) in os.environ["TUNE_OPS"].split(",")]
    for rnd in range(2):
        for op in ops:
            cur = best.get((op, cls), start[(op, cls)])      # start from the library's own table
            results = []
            for cand in candidates(op, cls):
                set_tune(op, cls, *cand)
                results.append((time_graph(fn, reps=REPS), cand))
            set_tune(op, cls, *cur)
            ref = time_graph(fn, reps=REPS)
            results.sort()
            t_best, c_best = results[0]
            if t_best < ref - MARGIN:
                best[(op, cls)] = c_best
                set_tune(op, cls, *c_best)
            print("  round %d %-12s current %s %.1f us | best %s %.1f us | top3 %s" % (
                rnd, OPS[op], cur, ref, c_best, t_best, [(c, round(t, 1)) for t, c in results[:3]]), flush=True)
    print("%s: tuned graph %.1f us (was %.1f)" % (label, time_graph(fn), base), flush=True)

print("\n// default_tuning table (op, class) -> {cfg, ksplit, xcd}")
for (op, cls), (c, ks, x) in sorted(best.items()):
    print("  c->tune[%d][%d] = Tune{%d, %d, %d};   // %s" % (op, cls, c, ks, x, OPS[op]))
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"%s/%d" % (OPS[o], c): v for (o, c), v in best.items()}, open("gpurun_out/tune_%d_%d_%d.json" % (ARCH, N, T), "w"))
