"""Where do short bench windows lose time?  20-cycle windows bracketed like bench.py's: host wall time of each window against the
GPU time between two events recorded around the same launches, for window lengths 20 / 40 / 100."""
import os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paac_amd import train
from paac_amd.paac import DeviceRollout, PAACLearner
args = train.get_arg_parser().parse_args([])
args.game, args.arch = "breakout", "NATURE"
args.emulator_counts, args.max_local_steps, args.emulator_workers = 32, 5, 0
args.max_global_steps = 1 << 60
args.debugging_folder = tempfile.mkdtemp(prefix="paac_probe_")
nc, ec = train.get_network_and_environment_creator(args)
L = PAACLearner(nc, ec, args)
L.network.initialize(np.random.RandomState(0))
np.random.seed(42)
ro = DeviceRollout(L, ec.device_env_spec, sampler="numpy", sampler_seed=42)
with torch.cuda.stream(ro.stream):
    ro.capture()
ro.run_cycles(40)
ro.synchronize()
for K in (20, 40, 100, 20):
    host, gpu = [], []
    for w in range(9):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(ro.stream)
        ro.run_cycles(K)
        e1.record(ro.stream)
        ro.synchronize()
        torch.cuda.synchronize()
        host.append((time.perf_counter() - t0) * 1e3)
        gpu.append(e0.elapsed_time(e1))
    print("K=%3d  host ms/cycle %s | gpu ms/cycle %s" % (K, " ".join("%.4f" % (h / K) for h in host[2:]), " ".join("%.4f" % (g / K) for g in gpu[2:])))
