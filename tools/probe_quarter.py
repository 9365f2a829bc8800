"""GPU probe: the quarter-tile fc kernel against the whole-tile one and the float64 oracle (PAAC_FC_QUARTER is read at context
creation), managed and unmanaged, at 32 / 24 / 9 rows; and the per-cycle value drift of the 20-cycle loop test."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import network as onet
from paac_amd import hip_ops
from test_hip_network import make_case, upload_params, ARCH_ID

def run(arch, A, B, managed, quarter, scale):
    os.environ["PAAC_FC_QUARTER"] = "1" if quarter else "0"
    params, states, idx, y, adv = make_case(arch, A, B, seed=1, weight_scale=scale)
    ctx = hip_ops.Context(ARCH_ID[arch], A, max_batch=B)
    p = upload_params(ctx, params)
    if managed:
        ctx.set_managed_weights(True); ctx.pack_weights(p)
    st = torch.from_numpy(states).cuda()
    logits = torch.zeros((B, A), device="cuda"); probs = torch.zeros((B, A), device="cuda"); values = torch.zeros(B, device="cuda")
    ctx.forward(p, st, logits, probs, values)
    torch.cuda.synchronize()
    return params, states, logits.cpu().numpy(), values.cpu().numpy()

for arch, A, B in (("NATURE", 4, 32), ("NATURE", 6, 24), ("NATURE", 18, 9), ("NIPS", 6, 32), ("NIPS", 6, 64)):
    for managed in (True, False):
        for scale in (1.0, 3.5):
            params, states, l0, v0 = run(arch, A, B, managed, False, scale)
            _, _, l1, v1 = run(arch, A, B, managed, True, scale)
            ref = onet.forward(params, states, arch, dtype=np.float64)
            print("%s A=%d B=%d managed=%d scale %.1f: whole dlogit %.2e dv %.2e | quarter dlogit %.2e dv %.2e | quarter-whole %.2e %.2e (row of max %d)" % (
                arch, A, B, managed, scale, np.abs(l0 - ref["logits"]).max(), np.abs(v0 - ref["v"]).max(),
                np.abs(l1 - ref["logits"]).max(), np.abs(v1 - ref["v"]).max(), np.abs(l1 - l0).max(), np.abs(v1 - v0).max(),
                int(np.abs(v1 - v0).argmax())), flush=True)

if os.environ.get("PROBE_LOOP", "1") == "1":
    import test_learner_gpu as T
    from paac_amd.paac import DeviceRollout
    for quarter in (0, 1):
        os.environ["PAAC_FC_QUARTER"] = str(quarter)
        args = T.make_args(game="breakout", arch="NATURE", emulator_counts=32, emulator_workers=0, max_local_steps=5,
                           max_global_steps=1 << 40, synthetic_terminal_p=0.05, sampler="numpy", test_seed=11,
                           synthetic_raw_frames=False)
        learner, params, env_creator = T.build_learner(args)
        np.random.seed(args.test_seed)
        learner.global_step = learner.init_network()
        ro = DeviceRollout(learner, env_creator.device_env_spec, sampler="numpy", use_graph=True)
        want = T.oracle_cycles(args, params, env_creator, 20, "NATURE")
        out = []
        for c in range(20):
            ro.run_cycle(); ro.synchronize()
            same = np.array_equal(ro.actions.view(-1).cpu().numpy(), np.argmax(want[c]["actions"], axis=1))
            out.append("%d:%s%.1e" % (c, "" if same else "!", np.abs(ro.values.cpu().numpy() - want[c]["values"]).max()))
        got = learner.network.get_parameters()
        print("quarter=%d per-cycle max |dv| (! = actions differ):" % quarter, " ".join(out))
        print("   final params max diff", max(np.abs(got[k] - v).max() for k, v in want[-1]["params"].items()), flush=True)
        ro.close()
