"""Debugging aid: is one cycle from a restored state reproducible bit for bit -- eager vs eager, replay vs replay, eager vs
replay?  (The check DeviceRollout._replay_matches_eager relies on.)  World of one with the collectives forced on."""
import os, sys, tempfile
os.environ.update(PAAC_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
                  PAAC_FORCE_COLLECTIVES="1", PAAC_ALLREDUCE="graph", PAAC_VERIFY_EXCHANGE="0")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paac_amd import parallel, train
args = train.get_arg_parser().parse_args([])
parallel.init_from_env(args)
import torch
from paac_amd.paac import DeviceRollout, PAACLearner
args.game, args.arch = "breakout", "NATURE"
args.emulator_counts, args.max_local_steps, args.emulator_workers = int(os.environ.get("DBG_N", "8")), 5, 0
args.max_global_steps = 1 << 40
args.synthetic_terminal_p = 0.1
args.debugging_folder = tempfile.mkdtemp(prefix="paac_dbg_")
if os.environ.get("DBG_POISON", "") == "1":      # the library's hipMalloc'd buffers start out as NaN patterns instead of zeros
    x = torch.full((3 << 28,), float("nan"), device="cuda")
    torch.cuda.synchronize()
    del x
    torch.cuda.empty_cache()
nc, ec = train.get_network_and_environment_creator(args)
L = PAACLearner(nc, ec, args)
L.network.initialize(np.random.RandomState(0))
np.random.seed(4)
ro = DeviceRollout(L, ec.device_env_spec, sampler="numpy", use_graph=True)
with torch.cuda.stream(ro.stream):
    ro.capture()
    print("graph_exchange", ro.graph_exchange, "reuse", ro.reuse_acting)
    state = ro._cycle_state()
    saved = [t.clone() for t in state]
    def restore():
        for t, s in zip(state, saved):
            t.copy_(s)
        L.ctx.pack_weights(L.network.params)
    def eager():
        ro._rollout_and_backward(0); ro._exchange(None); g = L.grad.clone(); ro._update()
        out = [g] + [t.clone() for t in state]; restore(); return out
    def replay():
        ro.graph_a[0].launch(); out = [L.grad.clone()] + [t.clone() for t in state]; restore(); return out
    runs = {"e1": eager(), "e2": eager(), "r1": replay(), "r2": replay(), "e3": eager()}
    ro.stream.synchronize()
names = ["GRAD@exchange", "states", "actions", "values", "rewards", "masks", "probs", "y", "adv", "ep_reward", "ep_len", "finished",
         "tick", "global_step", "params", "rms", "mom", "grad", "lr", "gnorm", "loss", "x1", "x2", "x3"]
for a, b in (("e1", "e2"), ("r1", "r2"), ("e1", "r1"), ("e1", "e3")):
    diffs = []
    for i, (x, y) in enumerate(zip(runs[a], runs[b])):
        if not torch.equal(x, y):
            d = (x.double() - y.double()).abs()
            diffs.append("%s(max %.3g, n %d, nan %d)" % (names[i] if i < len(names) else i, float(d.max()), int((d > 0).sum()),
                                                         int(torch.isnan(x.double()).sum())))
    print(a, "vs", b, "->", "identical" if not diffs else "; ".join(diffs))
lay = L.network.layout
g1, g2 = runs["e1"][0].cpu().numpy(), runs["r1"][0].cpu().numpy()
for t in lay["tensors"]:
    a, b = g1[t["offset"]:t["offset"] + t["size"]], g2[t["offset"]:t["offset"] + t["size"]]
    if not np.array_equal(a, b):
        print("  grad tensor", t["name"], "differs: max", np.abs(a - b).max(), "of", np.abs(a).max())
parallel.shutdown()
