"""CPU port of the reference loop for bench.py's `cpu_baseline` leg (TEST INFRASTRUCTURE, see __init__).

The reference's own TF-1.0 CPU path (`-d '/cpu:0'`, README.md:16-17) cannot run here (tensorflow and ALE are
absent; nothing can be installed), so the timed CPU baseline is this port (`kind: "port"`): the same cycle as
paac.py:59-183 -- T+1 batch-N policy inferences, per-env np.random.multinomial sampling in a Python list
comprehension (paac.py:44), per-env Python bookkeeping (paac.py:121-138), the float64 n-step return scan
(paac.py:144-149), one batch-N*T update with clip_by_global_norm(3.0) and TF-semantics RMSProp -- with
torch-CPU (oneDNN) standing in for TF's Eigen kernels, fp32, all host cores.  Environments are the same
synthetic BaseEnvironment plugins the GPU path is measured on, stepped in-process.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import network as onet
from . import rollout as oroll
from . import sampler as osamp


class TorchNet(object):
    def __init__(self, arch, num_actions, params):
        self.arch = onet.arch_key(arch)
        self.convs, self.flat, self.fc = onet.layer_dims(self.arch)
        self.names = [n for n, _ in onet.param_shapes(self.arch, num_actions)]
        self.p = {k: torch.tensor(np.asarray(v, dtype=np.float32), requires_grad=True) for k, v in params.items()}
        self.ms = {k: torch.ones_like(v) for k, v in self.p.items()}
        self.mom = {k: torch.zeros_like(v) for k, v in self.p.items()}

    def forward(self, states_u8):
        x = torch.from_numpy(states_u8).float().mul(float(onet.INPUT_SCALE)).permute(0, 3, 1, 2)
        for i, L in enumerate(self.convs):
            w = self.p["conv%d_weights" % (i + 1)].permute(3, 2, 0, 1)
            x = F.relu(F.conv2d(x, w, self.p["conv%d_biases" % (i + 1)], stride=L["stride"]))
        n = len(self.convs) + 1
        xf = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
        h = F.relu(xf @ self.p["fc%d_weights" % n] + self.p["fc%d_biases" % n])
        logits = h @ self.p["actor_output_weights"] + self.p["actor_output_biases"]
        pi = torch.softmax(logits, dim=1)
        v = (h @ self.p["critic_output_weights"] + self.p["critic_output_biases"]).reshape(-1)
        return pi, v

    def policy(self, states_u8):
        with torch.no_grad():
            pi, v = self.forward(states_u8)
        return v.numpy(), pi.numpy()

    def train_step(self, states, onehot, y, adv, lr, beta=0.02, clip=3.0, decay=0.99, eps=0.1):
        pi, v = self.forward(states)
        lp = torch.log(pi + 1e-30)
        ent = -(pi * lp).sum(1)
        logp = (lp * torch.from_numpy(onehot.astype(np.float32))).sum(1)
        actor = (-(logp * torch.from_numpy(adv.astype(np.float32)) + beta * ent)).mean()
        critic = (0.25 * (torch.from_numpy(y.astype(np.float32)) - v) ** 2).mean()
        loss = 5.0 * (actor + critic)
        grads = torch.autograd.grad(loss, [self.p[k] for k in self.names])
        with torch.no_grad():
            gn = torch.sqrt(sum((g * g).sum() for g in grads))
            scale = clip * min(1.0 / float(gn), 1.0 / clip)
            for k, g in zip(self.names, grads):
                g = g * scale
                self.ms[k].add_((g * g - self.ms[k]) * (1.0 - decay))
                self.p[k].sub_(lr * g / torch.sqrt(self.ms[k] + eps))
        return float(loss.detach())


def run(envs, arch, num_actions, T, params, min_seconds=10.0, warmup_cycles=2, seed=42, gamma=0.99,
        initial_lr=0.0224, lr_annealing_steps=80000000, windows=3):
    """Time the port.  Returns dict(steps_per_s = median over `windows` windows, cycles, seconds, cores, ...)."""
    net = TorchNet(arch, num_actions, params)
    rs = np.random.RandomState(seed)
    def sample(pi):
        # paac.py:42-44 verbatim; modern numpy rejects p - epsneg < 0 (a saturated policy on the noise frames),
        # where 2017-era numpy drew and never selected the category -> fall back to the restatement with that behaviour.
        try:
            return osamp.sample_numpy_reference(pi, rs)
        except ValueError:
            return osamp.sample_mt_restated(pi, rs)[0]

    ro = oroll.OracleRollout(envs, num_actions, T, gamma, initial_lr, lr_annealing_steps, net.policy, sample)
    N = len(envs)

    def one_cycle():
        cyc = ro.cycle()
        net.train_step(cyc["states"], cyc["actions"], cyc["y"], cyc["adv"], cyc["lr"])

    for _ in range(warmup_cycles):
        one_cycle()
    # pick the intra-op thread count that runs this small-batch loop fastest (all cores is rarely it): 4 timed cycles per
    # candidate, then the count stays fixed for every timed window
    all_cores = torch.get_num_threads()
    best = (None, float("inf"))
    for nt in sorted({all_cores, max(1, all_cores // 2), max(1, all_cores // 4), min(all_cores, 16), min(all_cores, 8)}):
        torch.set_num_threads(nt)
        one_cycle()
        t = time.time()
        for _ in range(4):
            one_cycle()
        dt = time.time() - t
        if dt < best[1]:
            best = (nt, dt)
    torch.set_num_threads(best[0])
    # `windows` equal windows; the reported rate is the median window's
    rates, total_cycles, t_all = [], 0, time.time()
    for _ in range(windows):
        t0 = time.time()
        cycles = 0
        while True:
            one_cycle()
            cycles += 1
            dt = time.time() - t0
            if dt >= min_seconds / windows:
                break
        rates.append(cycles * N * T / dt)
        total_cycles += cycles
    rates.sort()
    return dict(steps_per_s=rates[len(rates) // 2], cycles=total_cycles, seconds=time.time() - t_all,
                cores=torch.get_num_threads(), windows=windows, window_rates=rates)
