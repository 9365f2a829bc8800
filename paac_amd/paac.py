"""PAAC learner (mirrors reference paac.py:13-187).

`PAACLearner(network_creator, environment_creator, args)`, `.train()`, `.cleanup()` and the static
`choose_next_actions(network, num_actions, states, session)` keep the reference's signatures.

train() has two executions of the SAME cycle (paac.py:99-183):
  * device-resident (environments with a device twin): T x [policy forward -> sample -> env step] ->
    bootstrap forward -> n-step returns -> forward/backward -> (RCCL all-reduce) -> clip + RMSProp, every
    stage a HIP kernel, the whole cycle replayed as a hipGraph; nothing crosses PCIe per step.
  * host environments (any BaseEnvironment plugin): the reference's loop, with session.run replaced by
    kernel launches and the numpy sampler/return maths by their HIP twins; actions are bit-identical to
    the reference's numpy sampler on the same np.random seed (paac_sample_mt).
"""
import logging
import os
import time

import numpy as np
import torch

from . import hip_ops, logger_utils, parallel
from .actor_learner import ActorLearner
from .runners import EmulatorRunner, RawEmulatorRunner, Runners


class DeviceRollout(object):
    """Device-resident rollout/update cycle for N local environments x T steps."""

    def __init__(self, learner, env_spec, sampler="philox", sampler_seed=42, env_offset=0, use_graph=True,
                 total_envs=None):
        L = learner
        self.L = L
        dev = L.torch_device
        N, T, A = L.emulator_counts, L.max_local_steps, L.num_actions
        self.N, self.T, self.A = N, T, A
        self.total_envs = total_envs if total_envs is not None else N * L._world()
        self.env_spec = env_spec
        self.env_offset = int(env_offset)
        self.sampler = sampler
        self.sampler_seed = int(sampler_seed)
        self.use_graph = use_graph
        # Observation buffer of 2T+1 slots: even cycles use slots 0..T, odd cycles T..2T, so the last observation of
        # an even cycle IS the first of the odd one, and the T+1 observations of a cycle are contiguous (one
        # forward at batch N*(T+1) serves the update and the bootstrap); the last environment step of an odd cycle
        # writes its new stacks to slot 2T AND to slot 0 (stack_out2).  Two captured graphs alternate.
        self.states = torch.zeros((2 * T + 1, N, 84, 84, 4), dtype=torch.uint8, device=dev)
        self.parity = 0
        self.actions = torch.zeros((T, N), dtype=torch.int32, device=dev)
        self.values = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self.rewards = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self.masks = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self.probs = torch.zeros((N, A), dtype=torch.float32, device=dev)
        self.y = torch.zeros((T * N,), dtype=torch.float32, device=dev)
        self.adv = torch.zeros((T * N,), dtype=torch.float32, device=dev)
        self.ep_reward = torch.zeros((N,), dtype=torch.float32, device=dev)
        self.ep_len = torch.zeros((N,), dtype=torch.int32, device=dev)
        self.finished = torch.zeros((hip_ops.FINISHED_RING_BYTES // 4,), dtype=torch.int32, device=dev)
        self.tick = torch.zeros((1,), dtype=torch.int64, device=dev)          # env steps taken (per env)
        self.global_step_dev = torch.full((1,), int(L.global_step), dtype=torch.int64, device=dev)
        self.raw = None
        # lets the large shards' sampler spread its walk over several workgroups (hip_ops.sample_mt_synth_step)
        self.walk_scratch = hip_ops.walk_scratch(N, A, dev) if N * (A - 1) <= hip_ops.FUSED_SAMPLE_MAX_DRAWS else None
        if env_spec.get("raw_frames"):
            self.raw = torch.zeros((N, 2, hip_ops.RAW_H, hip_ops.RAW_W), dtype=torch.uint8, device=dev)
        if sampler == "numpy":
            self.mt_state = hip_ops.mt_state_from_numpy(np.random.get_state(), dev)
            self.mt_scratch = hip_ops.sample_mt_scratch(N, A, dev)
        self.stream = torch.cuda.Stream(device=dev)
        self.graph_a = [None, None]
        self.graph_conv = [None, None]
        self.graph_b = None
        self.graph_multi = [None, None]                # MULTI consecutive cycles in one launch, by the ring parity they start at
        self.graph_multi_long = [None, None]           # MULTI_LONG of them
        self.graph_ua = [None, None]                   # data parallel: update of the previous cycle + graph_a
        self.pending_update = False
        # data parallel: the cycle contains the gradient exchange
        self.phased = parallel.collectives_active()
        # PAAC_ALLREDUCE = graph (default): ONE all-reduce of the whole flat gradient after the full backward, CAPTURED into
        #   the cycle's hipGraph between the backward and the optimizer step -- the data-parallel cycle is replayed exactly
        #   like the single-process one (MULTI cycles per graph launch, nothing issued eagerly); if the collective cannot be
        #   captured (backend gloo, or the capture raises) the loop falls back to `single`.
        # single: the same all-reduce issued eagerly on the rollout stream between two graph launches per cycle (measured
        #   with a one-rank RCCL group: +15 us per cycle over the plain replay).
        # split: the fc/heads tail (95 % of the bytes) goes out on the collective's stream while the conv backward still
        #   computes (+61 us per cycle of extra graph launch and cross-stream waits before any wire time: pays only when
        #   the large all-reduce takes longer than about 120 us)
        mode = os.environ.get("PAAC_ALLREDUCE", "graph")
        self.single_exchange = mode != "split"
        self.graph_exchange = self.phased and mode == "graph" and use_graph and parallel.backend() == "nccl"
        self.side_group = parallel.side_group() if (self.phased and not self.single_exchange) else None    # collective call
        # what the exchange really runs as (graph mode may fall back: capture()); PAAC_VERIFY_EXCHANGE=0 skips the check of
        # the replayed collective against the eager one
        self.exchange_mode = "none" if not self.phased else ("split" if not self.single_exchange else "single")
        self.exchange_fallback = None
        self.short_first = os.environ.get("PAAC_SHORT_GRAPH_FIRST", "1") != "0"
        self.spin_sync = os.environ.get("PAAC_SPIN_SYNC", "1") != "0"
        self.verify_exchange = os.environ.get("PAAC_VERIFY_EXCHANGE", "1") != "0"
        # flat gradient = [conv tensors | fc_w fc_b actor critic]; the tail is 95 % of the bytes
        self.tail_offset = [t["offset"] for t in L.network.layout["tensors"] if t["name"].startswith("fc")][0]
        # The update needs no forward of its own: weights are frozen inside a cycle, so the T acting forwards have already
        # computed the activations paac.py:163-165 recomputes (TensorFlow's feed/fetch model forces that second pass; nothing
        # else does).  Acting step t keeps its rows at [t*N, (t+1)*N) of the training activation set, the N bootstrap
        # observations (paac.py:140-142) run as one acting-shaped forward into rows [T*N, (T+1)*N), and the backward starts
        # from there.  PAAC_REUSE_ACTING=0 restores the recomputed training forward (both routes are tested).
        # PAAC_MT_AHEAD=0: the large shards' step as paac_forward + paac_sample_mt_synth_step (every sampler workgroup rebuilds
        # the MT19937 blocks and doubles itself)
        # (the fused conv launches -- csrc/tower.h, csrc/tower2.h -- exist for the reference's two stock trunks)
        towered = getattr(L.network, "ARCH", None) in ("NATURE", "NIPS") and os.environ.get("PAAC_TOWER", "1") != "0"
        self.act_step_large = os.environ.get("PAAC_MT_AHEAD", "1") != "0" and N <= hip_ops.ACT_STEP_MAX_ENVS_LARGE and towered
        self.reuse_acting = os.environ.get("PAAC_REUSE_ACTING", "1") != "0" and towered and N <= hip_ops.KEEP_FORWARD_MAX_ROWS
        hip_ops.synth_reset(env_spec["seed"], self.env_offset, self.states[0], self.raw)
        torch.cuda.synchronize(dev)

    # -- stages --------------------------------------------------------------------------------------
    def _slot(self, parity, t):
        return parity * self.T + t

    def rollout_states(self, parity=None):
        """[T*N,84,84,4] view of the states the LAST run cycle trained on (t-major, paac.py:151)."""
        parity = (self.parity ^ 1) if parity is None else parity
        T, N = self.T, self.N
        return self.states[parity * T:(parity + 1) * T].view(T * N, 84, 84, 4)

    def _rollout_and_backward(self, parity):
        L, T, N = self.L, self.T, self.N
        params = L.network.params
        st = [self.states[self._slot(parity, t)] for t in range(T + 1)]
        keep = self.reuse_acting
        for t in range(T):
            if keep:
                L.ctx.keep_next_forward(t * N)
            # the ring wraps after an odd cycle: its last step also writes slot 0 (the next even cycle's first slot)
            wrap = self.states[0] if (parity == 1 and t == T - 1) else None
            # (path B -- self.raw -- rides the same fused launches: they write the raw screen pairs instead of shifting the
            # stacks and the preprocess launch follows)
            fused = self.sampler == "numpy" and N * (self.A - 1) <= hip_ops.FUSED_SAMPLE_MAX_DRAWS
            if fused and (self.act_step_large or (N <= hip_ops.ACT_STEP_MAX_ENVS and N * (self.A - 1) <= hip_ops.ACT_STEP_MAX_DRAWS)):
                # the whole step in three launches: conv tower, fc + head partials, heads finish + sampler + env step -- or, for
                # the large shards, four (heads finish on its own; the sampler's MT19937 doubles made meanwhile by a spare
                # workgroup of the fc launch)
                L.ctx.act_step_mt(params, st[t], self.mt_state, self.actions[t], self.probs, self.values[t],
                                  self.env_spec["seed"], self.env_offset, self.env_spec["terminal_threshold"], self.tick, t,
                                  st[t + 1], self.rewards[t], self.masks[t], self.ep_reward, self.ep_len, self.finished,
                                  stack_out2=wrap, raw_scratch=self.raw, walk_scratch=self.walk_scratch)
                continue
            if fused:
                # numpy-parity sampler and env step in one launch (the frame shift does not need the action)
                L.ctx.forward(params, st[t], probs=self.probs, values=self.values[t])
                hip_ops.sample_mt_synth_step(self.probs, self.mt_state, self.actions[t], self.env_spec["seed"],
                                             self.env_offset, self.env_spec["terminal_threshold"], self.tick, t,
                                             st[t], st[t + 1], self.rewards[t], self.masks[t], self.ep_reward,
                                             self.ep_len, self.finished, stack_out2=wrap, walk_scratch=self.walk_scratch,
                                             raw_scratch=self.raw)
                continue
            if self.sampler == "numpy":
                L.ctx.forward(params, st[t], probs=self.probs, values=self.values[t])
                hip_ops.sample_mt(self.probs, self.mt_state, self.mt_scratch, self.actions[t])
            elif self.raw is None:      # counter-based sampler AND the env step inside the heads kernel
                L.ctx.forward_sample_synth_step(params, st[t], self.sampler_seed, self.tick, t, self.env_offset,
                                                self.actions[t], self.env_spec["seed"],
                                                self.env_spec["terminal_threshold"], st[t + 1], self.rewards[t],
                                                self.masks[t], self.ep_reward, self.ep_len, self.finished,
                                                probs=self.probs, values=self.values[t])
                if wrap is not None:          # this launch has no second output
                    wrap.copy_(st[t + 1])
                continue
            else:       # counter-based sampler fused into the heads kernel
                L.ctx.forward_sample(params, st[t], self.sampler_seed, self.tick, t, self.env_offset,
                                     self.actions[t], probs=self.probs, values=self.values[t])
            hip_ops.synth_step(self.env_spec["seed"], self.env_offset, self.actions[t],
                               self.env_spec["terminal_threshold"], self.tick, t, st[t], st[t + 1],
                               self.rewards[t], self.masks[t], self.ep_reward, self.ep_len, self.finished,
                               stack_out2=wrap, raw_scratch=self.raw)
        # training forward over the T*N rollout rows with the N bootstrap observations appended (paac.py:140-142)
        # (the forward stops after the fc layer: the heads of the rollout rows AND the value head of the bootstrap rows are
        # finished inside the backward's first launch)
        if keep:
            L.ctx.bootstrap_forward_trunk(params, st[T], T * N)
        else:
            L.ctx.train_forward_trunk(params, self.states[parity * T:(parity + 1) * T + 1].view((T + 1) * N, 84, 84, 4))
        # n-step returns + global_step/lr schedule + frame counter (paac.py:127,144-156) ride in the backward's first
        # launch.  One process: whole backward here, with the slab reduction of the conv weight gradients left to the norm
        # pass of the optimizer step that follows (phase 3: one launch less); data parallel: the complete gradient
        # (phase 0: the all-reduce needs it) or, in the split form, heads + fc only (phase 1), so that the all-reduce of
        # the fc/heads gradient tail overlaps the conv backward (phase 2, _backward_conv)
        L.ctx.loss_backward_returns(params, self.rollout_states(parity), self.actions.view(-1), None, self.rewards,
                                    self.masks, self.values, L.gamma, self.y, self.adv, L.entropy_beta, L.grad,
                                    L.loss_dev, forward_done=True,
                                    phase=(0 if self.single_exchange else 1) if self.phased else 3,
                                    global_step_dev=self.global_step_dev, increment=self.total_envs * T,
                                    initial_lr=L.initial_lr, lr_annealing_steps=L.lr_annealing_steps, lr_out_dev=L.lr_dev,
                                    tick_dev=self.tick, tick_inc=T)

    def _backward_conv(self, parity):
        L = self.L
        L.ctx.loss_backward(L.network.params, self.rollout_states(parity), self.actions.view(-1), self.y, self.adv,
                            L.entropy_beta, L.grad, L.loss_dev, forward_done=True, phase=2)

    def _update(self):
        L = self.L
        L.ctx.clip_rmsprop(L.network.params, L.grad, L.rms, L.mom, L.lr_dev, L.alpha, L.momentum, L.e, L.clip_norm,
                           L.clip_mode, L._grad_scale(), L.gnorm_dev)

    def capture(self):
        """Capture the cycle into hipGraphs: one per observation-ring parity (and a separate update graph when a
        gradient all-reduce sits between backward and the optimizer step)."""
        def captured(fn):
            g = hip_ops.Graph()
            g.begin()
            try:
                fn()
            except Exception:
                g.abort()
                raise
            g.end()
            return g

        def cycle(parity, with_update):
            self._rollout_and_backward(parity)
            if with_update:
                if self.graph_exchange:
                    self._exchange(None)          # recorded into the graph like the kernels around it
                self._update()

        if self.graph_exchange:
            # the communicator must exist before a capture can record its collective: one eager all-reduce of a scratch
            # word first; a capture that raises leaves the eager form in charge
            reason = None
            try:
                parallel.allreduce_sum_(torch.zeros(1, dtype=torch.float32, device=self.L.torch_device))
                torch.cuda.synchronize(self.L.torch_device)
                with torch.cuda.stream(self.stream):
                    for parity in (0, 1):
                        self.graph_a[parity] = captured(lambda: cycle(parity, True))
                    for p0 in (0, 1):
                        self.graph_multi[p0] = captured(lambda: [cycle((k + p0) & 1, True) for k in range(self.MULTI)])
                        if self.MULTI_LONG > self.MULTI:
                            self.graph_multi_long[p0] = captured(lambda: [cycle((k + p0) & 1, True) for k in range(self.MULTI_LONG)])
            except Exception as exc:      # noqa: BLE001 -- whatever the runtime / RCCL raised: keep training, eagerly
                reason = "the capture raised: %s" % (exc,)
            # every rank takes the same route: one that kept replaying graphs beside a peer issuing its collectives eagerly
            # would pair the wrong calls
            if not parallel.all_ranks(reason is None, self.L.torch_device):
                reason = reason or "the capture failed on another rank"
            elif self.verify_exchange:
                # ... and before the replayed collective is trusted: the same cycle, from the same state, once with the
                # eager exchange and once replayed -- the exchanged gradients must agree bit for bit, on every rank
                with torch.cuda.stream(self.stream):
                    same = self._replay_matches_eager()
                if not parallel.all_ranks(same, self.L.torch_device):
                    reason = "a replayed cycle's exchanged gradient differs from the eager exchange's" + ("" if not same else " on another rank")
            if reason is None:
                self.exchange_mode = "graph"
                return
            logging.warning("PAAC_ALLREDUCE=graph not used (%s): the all-reduce is issued eagerly between two graph launches", reason)
            self.exchange_fallback = reason
            self.close()
            self.graph_exchange = False
        with torch.cuda.stream(self.stream):
            for parity in (0, 1):
                self.graph_a[parity] = captured(lambda: cycle(parity, not self.phased))
                if self.phased:
                    if not self.single_exchange:
                        self.graph_conv[parity] = captured(lambda: self._backward_conv(parity))
                    # the optimizer step of cycle k rides in front of cycle k+1's graph: two graph launches per
                    # cycle around the exchange instead of three (synchronize() flushes a pending step)
                    self.graph_ua[parity] = captured(lambda: (self._update(), cycle(parity, False)))
            if self.phased:
                self.graph_b = captured(self._update)
            else:
                for p0 in (0, 1):
                    self.graph_multi[p0] = captured(lambda: [cycle((k + p0) & 1, True) for k in range(self.MULTI)])
                    if self.MULTI_LONG > self.MULTI:
                        self.graph_multi_long[p0] = captured(lambda: [cycle((k + p0) & 1, True) for k in range(self.MULTI_LONG)])

    # -- trust, but verify: the captured exchange ---------------------------------------------------------
    def _cycle_state(self):
        """Every tensor one cycle reads and writes (the rollout's and the learner's): a snapshot of them is a restart point."""
        L = self.L
        ts = [self.states, self.actions, self.values, self.rewards, self.masks, self.probs, self.y, self.adv, self.ep_reward,
              self.ep_len, self.finished, self.tick, self.global_step_dev, L.network.params, L.rms, L.mom, L.grad, L.lr_dev,
              L.gnorm_dev, L.loss_dev]
        for name in ("raw", "walk_scratch", "mt_state"):
            t = getattr(self, name, None)
            if t is not None:
                ts.append(t)
        return ts

    def _replay_matches_eager(self):
        """One cycle from the current state with the all-reduce issued eagerly, the state put back, the same cycle replayed
        from the captured graph (all-reduce inside), the state put back again: True when the two exchanged gradients are
        bit-identical.  Nothing of either run survives: training starts from the state this was called in."""
        L = self.L
        state = self._cycle_state()
        saved = [t.clone() for t in state]

        def restore():
            for t, s in zip(state, saved):
                t.copy_(s)
            L.ctx.pack_weights(L.network.params)          # the optimizer step re-packed the updated weights

        self._rollout_and_backward(0)
        self._exchange(None)
        eager = L.grad.clone()
        self._update()                                    # (consumes what the backward left pending in the ctx)
        restore()
        self.graph_a[0].launch()                          # backward -> captured all-reduce -> update; grad stays as exchanged
        replayed = L.grad.clone()
        restore()
        self.stream.synchronize()
        return bool(torch.equal(eager, replayed))

    def check_replicas(self, what="weights"):
        """Data parallel: all ranks must hold bit-identical weights and optimizer slots (identical start, identical
        all-reduced gradient, identical update).  Raises parallel.ReplicaMismatch on every rank when they do not."""
        L = self.L
        self.synchronize()
        names = ("params", "rms", "mom") if what == "weights" else ("grad",)
        tensors = [L.network.params, L.rms, L.mom] if what == "weights" else [L.grad]
        with torch.cuda.stream(self.stream):
            ok, same = parallel.replicas_identical(tensors)
        if not ok:
            bad = [n for n, s in zip(names, same) if not s]
            raise parallel.ReplicaMismatch("data-parallel replicas diverged: %s differ between ranks (exchange mode %s, rank %d of %d)"
                                           % (", ".join(bad), self.exchange_mode, parallel.rank(), parallel.world_size()))
        return True

    # cycles per launch of graph_multi / graph_multi_long (even: the ring parity is back at 0 afterwards).  Measured at the
    # headline configuration, cycles per launch 4 / 8 / 16 / 32: 643.5 / 645.6 / 648.6 / 648.1 k env-steps/s.
    MULTI = max(2, int(os.environ.get("PAAC_CYCLES_PER_LAUNCH", "4")) // 2 * 2)
    MULTI_LONG = max(MULTI, int(os.environ.get("PAAC_CYCLES_PER_LONG_LAUNCH", "16")) // 2 * 2)

    def run_cycles(self, count):
        """`count` cycles.  Graph replay batches them MULTI_LONG or MULTI per hipGraph launch where it can: the gap between
        two graph launches on the GPU (about 8 us) is paid once per batch instead of every cycle."""
        count = int(count)
        if self.use_graph and self.graph_a[0] is None and count > 0:
            with torch.cuda.stream(self.stream):
                self.capture()            # first: a captured exchange that is refused changes which graphs exist
        while count > 0:
            if self.use_graph and (not self.phased or self.graph_exchange) and count >= self.MULTI:
                # (an even number of cycles per launch: the ring parity is the same afterwards; there is a graph per starting
                # parity -- a caller that has run an odd number of cycles, e.g. a 5-cycle warm-up, keeps the batched launches)
                with torch.cuda.stream(self.stream):
                    # the SHORT graph first: submitting a replay costs host time in proportion to its nodes, and an idle GPU
                    # waits for it -- behind MULTI cycles already running, the long graphs' submissions are hidden
                    if self.graph_multi_long[self.parity] is not None and count >= self.MULTI_LONG and (
                            self.short_first is False or count % self.MULTI_LONG < self.MULTI):
                        self.graph_multi_long[self.parity].launch()
                        count -= self.MULTI_LONG
                        continue
                    self.graph_multi[self.parity].launch()
                count -= self.MULTI
            else:
                self.run_cycle()
                count -= 1

    def run_cycle(self):
        with torch.cuda.stream(self.stream):
            if self.use_graph:
                if self.graph_a[0] is None:
                    self.capture()
                if self.phased and not self.graph_exchange:
                    (self.graph_ua if self.pending_update else self.graph_a)[self.parity].launch()
                    self._exchange(None if self.single_exchange else self.graph_conv[self.parity].launch)
                    self.pending_update = True
                else:
                    self.graph_a[self.parity].launch()
            else:
                self._rollout_and_backward(self.parity)
                if self.phased:
                    self._exchange(None if self.single_exchange else (lambda: self._backward_conv(self.parity)))
                self._update()
        self.parity ^= 1

    def _exchange(self, conv_backward):
        """Sum all-reduce of the flat gradient: one collective on our stream after the full backward (conv_backward is
        None), or in two pieces -- the fc/heads tail goes out on the collective's own stream while `conv_backward` still
        computes the conv head on ours; the update waits for both."""
        grad = self.L.grad
        if conv_backward is None:          # PAAC_ALLREDUCE=single: the backward is complete, one collective
            parallel.allreduce_sum_(grad)  # a blocking-style call runs ON our stream: no cross-stream event hops
            return
        tail = parallel.allreduce_sum_async(grad[self.tail_offset:])
        conv_backward()
        # the small conv part follows the conv backward on our own stream (nothing to overlap it with) and on a second
        # communicator (it does not queue behind the 6.4 MB one); the update then waits for the large one, which has been
        # travelling on the collective's stream meanwhile
        parallel.allreduce_sum_(grad[:self.tail_offset], group=self.side_group)
        if tail is not None:
            tail.wait()

    def synchronize(self):
        """Completes everything issued so far, including a data-parallel optimizer step still waiting to ride in
        front of the next cycle: afterwards weights, optimizer state and buffers are those of the last cycle."""
        if self.pending_update:
            with torch.cuda.stream(self.stream):
                self.graph_b.launch()
            self.pending_update = False
        if self.spin_sync:
            # poll instead of sleeping on the stream: the wake-up of a blocking synchronize costs tens of microseconds, which a
            # caller that brackets short runs (bench.py's 20-cycle windows are 5 ms) pays every time
            done = torch.cuda.Event()
            done.record(self.stream)
            while not done.query():
                pass
            return
        self.stream.synchronize()

    def finished_episodes(self):
        """Drain the device ring of finished episodes -> list of (reward, length)."""
        host = self.finished.cpu().numpy()
        count = int(host[0])
        n = min(count, 4096)
        rewards = host[2:2 + 4096].view(np.float32)
        lens = host[2 + 4096:2 + 8192]
        idx = [(count - n + i) & 4095 for i in range(n)]
        return count, [(float(rewards[i]), int(lens[i])) for i in idx]

    def close(self):
        for g in (self.graph_a[0], self.graph_a[1], self.graph_conv[0], self.graph_conv[1], self.graph_b, self.graph_multi[0],
                  self.graph_multi[1], self.graph_multi_long[0], self.graph_multi_long[1],
                  self.graph_ua[0], self.graph_ua[1]):
            if g is not None:
                g.close()
        self.graph_a = [None, None]
        self.graph_conv = [None, None]
        self.graph_b = None
        self.graph_multi = [None, None]
        self.graph_multi_long = [None, None]
        self.graph_ua = [None, None]


class DeviceObservations(object):
    """Observations of host environments that emit raw screens, built on the GPU: the screen pairs travel through
    pinned staging buffers, paac_preprocess_stack does max + nearest resize + history push for all environments in
    one launch per slot (one slot per step; HISTORY slots for the environments that were just reset)."""

    def __init__(self, n_envs, slots, dev):
        self.staging = torch.empty((slots, n_envs, 2, hip_ops.RAW_H, hip_ops.RAW_W), dtype=torch.uint8).pin_memory()
        self.mask_staging = torch.empty((slots, n_envs), dtype=torch.uint8).pin_memory()
        self.d_raw = torch.empty((slots, n_envs, 2, hip_ops.RAW_H, hip_ops.RAW_W), dtype=torch.uint8, device=dev)
        self.d_mask = torch.empty((slots, n_envs), dtype=torch.uint8, device=dev)
        self.current = torch.zeros((n_envs, 84, 84, 4), dtype=torch.uint8, device=dev)

    def update(self, raw, counts):
        """raw u8 [N,slots,2,210,160], counts [N]: environment i pushes its first counts[i] slots."""
        if tuple(raw.shape[2:]) != (2, hip_ops.RAW_H, hip_ops.RAW_W):
            raise ValueError("device preprocessing expects %dx%d screens, got %s" %
                             (hip_ops.RAW_H, hip_ops.RAW_W, tuple(raw.shape[3:])))
        counts = np.asarray(counts).astype(np.int64)
        for j in range(int(counts.max())):
            self.staging[j].copy_(torch.from_numpy(raw[:, j]))
            self.mask_staging[j].copy_(torch.from_numpy((counts > j).astype(np.uint8)))
            self.d_raw[j].copy_(self.staging[j], non_blocking=True)
            self.d_mask[j].copy_(self.mask_staging[j], non_blocking=True)
            hip_ops.preprocess_stack(self.d_raw[j], self.current, self.current, push_mask=self.d_mask[j])
        return self.current


class PAACLearner(ActorLearner):
    def __init__(self, network_creator, environment_creator, args):
        super(PAACLearner, self).__init__(network_creator, environment_creator, args)
        self.workers = args.emulator_workers
        self.args = args
        self.runners = None
        self.rollout = None
        self.metrics = None
        self.stop_requested = False      # set by the signal handler (train.py); honoured at the next cycle boundary

    def _open_metrics(self):
        """metrics.jsonl in the debugging folder (rank 0 only): what the reference sends to TensorBoard."""
        if self.metrics is None and parallel.rank() == 0 and getattr(self.args, "metrics", True):
            self.metrics = logger_utils.MetricsWriter(self.debugging_folder)
        return self.metrics

    def _progress_record(self, steps_per_s, steps_per_s_avg, last_ten):
        if self.metrics is None:
            return
        loss = self.loss_dev.cpu().numpy()
        stats = self.ctx.grad_stats(self.clip_norm, self.clip_mode)      # actor_learner.py:85-87 summaries
        self.metrics.write("gradients", global_step=int(self.global_step), **stats)
        self.metrics.write("progress", global_step=int(self.global_step), steps_per_s=float(steps_per_s),
                           steps_per_s_avg=float(steps_per_s_avg), last_10_rewards_avg=float(last_ten),
                           lr=float(self.lr_dev.item()), grad_norm=float(self.gnorm_dev.item()), loss=float(loss[0]),
                           actor_loss=float(loss[1]), critic_loss=float(loss[2]), entropy=float(loss[3]))
        self.metrics.flush()

    @staticmethod
    def choose_next_actions(network, num_actions, states, session):
        network_output_v, network_output_pi = session.run(
            [network.output_layer_v, network.output_layer_pi],
            feed_dict={network.input_ph: states})
        action_indices = PAACLearner._sample_policy_action(network_output_pi)
        new_actions = np.eye(num_actions)[action_indices]
        return new_actions, network_output_v, network_output_pi

    @staticmethod
    def _sample_policy_action(probs):
        """paac.py:34-45 executed by paac_sample_mt on the global np.random MT19937 stream: the stream is
        uploaded, advanced on the device exactly as numpy's multinomial would advance it, and written back."""
        probs = np.ascontiguousarray(np.asarray(probs, dtype=np.float32))
        dev = torch.device("cuda", torch.cuda.current_device())
        p = torch.from_numpy(probs).to(dev)
        state = hip_ops.mt_state_from_numpy(np.random.get_state(), dev)
        scratch = hip_ops.sample_mt_scratch(p.shape[0], p.shape[1], dev)
        actions = torch.empty((p.shape[0],), dtype=torch.int32, device=dev)
        hip_ops.sample_mt(p, state, scratch, actions)
        np.random.set_state(hip_ops.mt_state_to_numpy(state))
        return [int(a) for a in actions.cpu().numpy()]

    def _use_device_envs(self):
        spec = getattr(self.environment_creator, "device_env_spec", None)
        return spec is not None and not getattr(self.args, "host_environments", False)

    def train(self):
        """Main actor learner loop for parallel advantage actor critic learning (paac.py:59-183)."""
        self.global_step = self.init_network()
        logging.debug("Starting training at Step {}".format(self.global_step))
        if self._use_device_envs():
            self._train_device()
        else:
            self._train_host()
        self.cleanup()

    # -- device-resident loop ------------------------------------------------------------------------
    def _train_device(self):
        args = self.args
        world = self._world()
        rank = 0
        if world > 1:
            import torch.distributed as dist
            rank = dist.get_rank()
        N, T = self.emulator_counts, self.max_local_steps
        self.rollout = DeviceRollout(self, self.environment_creator.device_env_spec,
                                     sampler=getattr(args, "sampler", "philox"),
                                     sampler_seed=getattr(args, "sampler_seed", 42), env_offset=rank * N,
                                     use_graph=getattr(args, "use_graph", True))
        counter = 0
        global_step_start = self.global_step
        start_time = time.time()
        log_every = 2048 / self.emulator_counts            # paac.py:172 (float division, as upstream)
        steps_per_cycle = N * T * world
        metrics = self._open_metrics()
        episodes_seen = 0
        from .actor_learner import CHECKPOINT_INTERVAL
        since_stop_check = 0
        # data parallel: the replicas are compared (checksums of weights and optimizer slots, MIN / MAX over ranks) right
        # after the first update -- with the exchanged gradient -- and then every PAAC_REPLICA_CHECK_CYCLES cycles; a
        # mismatch ends the run on every rank (parallel.ReplicaMismatch -> non-zero exit)
        check_every = max(1, int(os.environ.get("PAAC_REPLICA_CHECK_CYCLES", "1024")))
        next_check = 1 if parallel.collectives_active() else None
        while self.global_step < self.max_global_steps:
            if next_check is not None and counter >= next_check:
                if next_check == 1:
                    self.rollout.check_replicas("grad")
                self.rollout.check_replicas("weights")
                next_check = counter + check_every
            if world == 1:
                if self.stop_requested:
                    break
            elif since_stop_check >= 16:       # all ranks must leave at the same cycle (one collective per >= 16 cycles)
                since_stop_check = 0
                if parallel.any_rank(self.stop_requested, self.torch_device):
                    break
            loop_start_time = time.time()
            # as many cycles per call as fit before the next event the reference checks every cycle: the end of
            # training, the progress line (paac.py:172) and the checkpoint (actor_learner.py:89-93)
            chunk = 1
            if float(log_every).is_integer():
                until_end = -(-(self.max_global_steps - self.global_step) // steps_per_cycle)
                until_log = int(log_every) - counter % int(log_every)
                until_save = -(-(self.last_saving_step + CHECKPOINT_INTERVAL - self.global_step) // steps_per_cycle)
                chunk = max(1, min(until_end, until_log, until_save))
            if next_check is not None:
                chunk = max(1, min(chunk, next_check - counter))
            self.rollout.run_cycles(chunk)
            self.global_step += chunk * steps_per_cycle
            counter += chunk
            since_stop_check += chunk
            if counter % log_every == 0:
                self.rollout.synchronize()
                curr_time = time.time()
                count, eps = self.rollout.finished_episodes()
                last_ten = 0.0 if len(eps) < 1 else np.mean([r for r, _ in eps[-10:]])
                logging.info("Ran {} steps, at {} steps/s ({} steps/s avg), last 10 rewards avg {}"
                             .format(self.global_step, chunk * steps_per_cycle / (curr_time - loop_start_time),
                                     (self.global_step - global_step_start) / (curr_time - start_time), last_ten))
                if metrics is not None:
                    for r, l in eps[max(0, len(eps) - (count - episodes_seen)):]:     # new since the last drain
                        metrics.write("episode", global_step=int(self.global_step), reward=float(r), length=int(l))
                    episodes_seen = count
                    self._progress_record(chunk * steps_per_cycle / (curr_time - loop_start_time),
                                          (self.global_step - global_step_start) / (curr_time - start_time), last_ten)
            self.save_vars()          # _sync_device() flushes the rollout first when a checkpoint is due
        self.rollout.synchronize()
        if next_check is not None:
            self.rollout.check_replicas("weights")

    # -- host-environment loop (the reference's structure, kernels instead of session.run) ------------
    def _train_host(self):
        dev = self.torch_device
        N, T, A = self.emulator_counts, self.max_local_steps, self.num_actions
        counter = 0
        global_step_start = self.global_step
        total_rewards = []
        raw_mode = bool(getattr(self.args, "device_preprocess", False))
        if raw_mode:
            # environments hand out raw screen pairs; observations are built on the GPU (DeviceObservations)
            if not all(hasattr(e, "next_raw") and hasattr(e, "initial_raw") for e in self.emulators):
                raise Exception("--device_preprocess needs environments with initial_raw() / next_raw()")
            first = np.stack([emulator.initial_raw() for emulator in self.emulators]).astype(np.uint8)
            variables = [first, np.full(N, first.shape[1], dtype=np.float32), np.zeros(N, dtype=np.float32),
                         np.asarray([False] * N, dtype=np.float32), np.zeros((N, A), dtype=np.float32)]
            self.runners = Runners(RawEmulatorRunner, self.emulators, self.workers, variables)
            self.runners.start()
            shared_raw, shared_counts, shared_rewards, shared_episode_over, shared_actions = \
                self.runners.get_shared_variables()
            observations = DeviceObservations(N, first.shape[1], dev)
            observations.update(shared_raw, shared_counts)
            current_states = lambda: observations.current
        else:
            variables = [(np.asarray([emulator.get_initial_state() for emulator in self.emulators], dtype=np.uint8)),
                         (np.zeros(N, dtype=np.float32)),
                         (np.asarray([False] * N, dtype=np.float32)),
                         (np.zeros((N, A), dtype=np.float32))]
            self.runners = Runners(EmulatorRunner, self.emulators, self.workers, variables)
            self.runners.start()
            shared_states, shared_rewards, shared_episode_over, shared_actions = self.runners.get_shared_variables()
            host_states = torch.from_numpy(shared_states)
            # page-lock the shared observation array in place: the per-step H2D is then one DMA straight out of the
            # memory the emulator workers write, instead of a staged pageable copy
            pinned = hip_ops.pin_host_array(host_states)
            current_states = lambda: host_states

        emulator_steps = np.zeros(N, dtype=np.int64)
        total_episode_rewards = np.zeros(N, dtype=np.float64)
        d_states = torch.zeros((T, N, 84, 84, 4), dtype=torch.uint8, device=dev)
        d_cur = torch.zeros((N, 84, 84, 4), dtype=torch.uint8, device=dev)
        d_actions = torch.zeros((T, N), dtype=torch.int32, device=dev)
        d_values = torch.zeros((T, N), dtype=torch.float32, device=dev)
        d_rewards = torch.zeros((T, N), dtype=torch.float32, device=dev)
        d_masks = torch.zeros((T, N), dtype=torch.float32, device=dev)
        d_probs = torch.zeros((N, A), dtype=torch.float32, device=dev)
        d_vboot = torch.zeros((N,), dtype=torch.float32, device=dev)
        d_y = torch.zeros((T * N,), dtype=torch.float32, device=dev)
        d_adv = torch.zeros((T * N,), dtype=torch.float32, device=dev)
        mt_state = hip_ops.mt_state_from_numpy(np.random.get_state(), dev)
        mt_scratch = hip_ops.sample_mt_scratch(N, A, dev)
        rewards = np.zeros((T, N), dtype=np.float32)
        masks = np.zeros((T, N), dtype=np.float32)
        # the sampled action indices come back through a page-locked buffer: an asynchronous copy + an event instead of a
        # synchronising .cpu(), so the host does the previous step's bookkeeping while the GPU runs this step's forward
        h_actions = torch.zeros((N,), dtype=torch.int32).pin_memory()
        actions_ready = torch.cuda.Event()
        env_index = np.arange(N)
        params = self.network.params
        start_time = time.time()
        self.last_feed = None
        metrics = self._open_metrics()
        # data parallel: global_step counts the environment steps of ALL ranks (paac.py:127 counts every environment of
        # the one learner), like the device loop -- lr anneals and max_global_steps ends on the global count
        world = self._world()
        check_every = max(1, int(os.environ.get("PAAC_REPLICA_CHECK_CYCLES", "1024")))

        stock_rescale = type(self).rescale_reward is ActorLearner.rescale_reward

        def bookkeeping(t, step_rewards, step_overs):
            """paac.py:119-138 for one step, vectorised over the environments that did not end an episode; the finished
            ones are visited in index order with the global_step the reference's per-environment loop would show them."""
            masks[t] = 1.0 - step_overs
            total_episode_rewards[:] += step_rewards
            if stock_rescale:
                rewards[t] = np.clip(step_rewards, -1.0, 1.0)      # rescale_reward, actor_learner.py:95-101
            else:                                                  # a subclass's own rescale_reward, per reward like upstream
                rewards[t] = [self.rescale_reward(r) for r in step_rewards]
            emulator_steps[:] += 1
            before = self.global_step
            self.global_step += N * world
            for e in np.nonzero(step_overs)[0]:
                total_rewards.append(total_episode_rewards[e])
                if metrics is not None:                      # paac.py:130-135
                    metrics.write("episode", global_step=int(before + (e + 1) * world),
                                  reward=float(total_episode_rewards[e]), length=int(emulator_steps[e]))
                total_episode_rewards[e] = 0
                emulator_steps[e] = 0

        # One step's GPU work -- observations H2D, policy forward, numpy-parity sampler, action indices D2H -- as ONE hipGraph
        # launch per step where it can be captured (page-locked shared observations, up to 64 environments on a stock trunk:
        # paac_act_step_mt's sampling-only form, three kernels), instead of seven API calls whose host cost is what this
        # loop is made of; PAAC_HOST_GRAPH=0, or anything that cannot be captured, keeps the eager calls
        fused_act = (N <= hip_ops.ACT_STEP_MAX_ENVS and N * (A - 1) <= hip_ops.ACT_STEP_MAX_DRAWS
                     and getattr(self.network, "ARCH", None) in ("NATURE", "NIPS"))
        host_stream = torch.cuda.Stream(device=dev)
        host_stream.wait_stream(torch.cuda.current_stream(dev))
        step_graphs = [None] * T

        def step_gpu(t):
            d_states[t].copy_(current_states(), non_blocking=True)
            if fused_act:
                self.ctx.act_mt(params, d_states[t], mt_state, d_actions[t], d_probs, d_values[t])
            else:
                self.ctx.forward(params, d_states[t], probs=d_probs, values=d_values[t])
                hip_ops.sample_mt(d_probs, mt_state, mt_scratch, d_actions[t])
            h_actions.copy_(d_actions[t], non_blocking=True)

        if (os.environ.get("PAAC_HOST_GRAPH", "1") != "0" and not raw_mode and pinned and fused_act):
            with torch.cuda.stream(host_stream):
                try:
                    for t in range(T):
                        g = hip_ops.Graph()
                        g.begin()
                        try:
                            step_gpu(t)
                        except Exception:
                            g.abort()
                            raise
                        g.end()
                        step_graphs[t] = g
                except Exception as exc:      # noqa: BLE001 -- a copy or launch the runtime would not record: eager calls
                    logging.debug("host-plugin loop: per-step graph not captured (%s), issuing the calls eagerly", exc)
                    for g in step_graphs:
                        if g is not None:
                            g.close()
                    step_graphs = [None] * T
        ctx_stream = torch.cuda.stream(host_stream)
        ctx_stream.__enter__()          # everything below is issued on the loop's own stream
        while self.global_step < self.max_global_steps and not parallel.any_rank(self.stop_requested, dev):
            loop_start_time = time.time()
            pending = None
            for t in range(T):
                if step_graphs[t] is not None:
                    step_graphs[t].launch()
                else:
                    step_gpu(t)
                actions_ready.record()
                if pending is not None:            # the previous step's records, while the GPU is busy with this step
                    bookkeeping(*pending)
                actions_ready.synchronize()
                shared_actions[...] = 0.0          # one-hot rows (the convention of runners.py: workers take the argmax)
                shared_actions[env_index, h_actions.numpy()] = 1.0
                self.runners.update_environments()
                self.runners.wait_updated()
                if raw_mode:
                    observations.update(shared_raw, shared_counts)
                pending = (t, shared_rewards.astype(np.float32), shared_episode_over.astype(np.float32))
            bookkeeping(*pending)
            d_cur.copy_(current_states())
            self.ctx.forward(params, d_cur, values=d_vboot)
            d_rewards.copy_(torch.from_numpy(rewards))
            d_masks.copy_(torch.from_numpy(masks))
            hip_ops.nstep_returns(d_vboot, d_rewards, d_masks, d_values, self.gamma, d_y, d_adv)
            lr = self.get_lr()
            self.lr_dev.fill_(float(np.float32(lr)))
            self.ctx.loss_backward(params, d_states.view(T * N, 84, 84, 4), d_actions.view(-1), d_y, d_adv,
                                   self.entropy_beta, self.grad, self.loss_dev)
            self._allreduce_grad()
            self.ctx.clip_rmsprop(params, self.grad, self.rms, self.mom, self.lr_dev, self.alpha, self.momentum, self.e,
                                  self.clip_norm, self.clip_mode, self._grad_scale(), self.gnorm_dev)
            if getattr(self.args, "record_feeds", False):
                self.last_feed = dict(states=d_states.view(T * N, 84, 84, 4).cpu().numpy(), y=d_y.cpu().numpy(),
                                      adv=d_adv.cpu().numpy(), actions=d_actions.view(-1).cpu().numpy(), lr=lr,
                                      values=d_values.cpu().numpy(), global_step=self.global_step)
                if getattr(self.args, "feed_callback", None):
                    self.args.feed_callback(self.last_feed)
            if getattr(self.args, "cycle_callback", None):      # bench hook: one call per finished cycle, nothing copied
                self.args.cycle_callback(self.global_step)
            counter += 1
            if parallel.collectives_active() and (counter == 1 or counter % check_every == 0):
                ok, same = parallel.replicas_identical([params, self.rms, self.mom])
                if not ok:
                    raise parallel.ReplicaMismatch("data-parallel replicas diverged after %d updates: %s differ between ranks"
                                                   % (counter, ", ".join(n for n, s in zip(("params", "rms", "mom"), same) if not s)))
            if counter % (2048 / self.emulator_counts) == 0:
                curr_time = time.time()
                last_ten = 0.0 if len(total_rewards) < 1 else np.mean(total_rewards[-10:])
                logging.info("Ran {} steps, at {} steps/s ({} steps/s avg), last 10 rewards avg {}"
                             .format(self.global_step, T * N * world / (curr_time - loop_start_time),
                                     (self.global_step - global_step_start) / (curr_time - start_time), last_ten))
                self._progress_record(T * N * world / (curr_time - loop_start_time),
                                      (self.global_step - global_step_start) / (curr_time - start_time), last_ten)
            self.save_vars()
        host_stream.synchronize()
        ctx_stream.__exit__(None, None, None)
        for g in step_graphs:
            if g is not None:
                g.close()
        logging.debug("Host-plugin loop: %d update cycles, global step %d", counter, self.global_step)
        np.random.set_state(hip_ops.mt_state_to_numpy(mt_state))

    def _sync_device(self):
        """Everything issued so far has completed -- graph-replayed cycles still running on the rollout's own stream and
        a data-parallel optimizer step still waiting to ride in front of the next cycle included -- so a checkpoint
        taken now holds weights, rms and mom of one and the same update."""
        if self.rollout is not None:
            self.rollout.synchronize()
        super(PAACLearner, self)._sync_device()

    def cleanup(self):
        super(PAACLearner, self).cleanup()       # save_vars(True) -> _sync_device() first
        if self.runners is not None:
            self.runners.stop()
            self.runners = None
        if self.rollout is not None:
            self.rollout.close()
        if self.metrics is not None:
            self.metrics.close()
            self.metrics = None
