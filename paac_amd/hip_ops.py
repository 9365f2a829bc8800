"""Thin, shape-checked Python wrappers over the C-ABI (include/paac_hip.h).

torch is plumbing only (device memory + streams): every wrapper validates dtype / device / contiguity /
extent on the host BEFORE the launch (a kernel that faults can take the whole node down), then hands raw
device pointers to libpaac_hip.so on torch's current HIP stream.
"""
import ctypes
import gc

import numpy as np
import torch

from . import _lib

OBS_SHAPE = (84, 84, 4)
RAW_H, RAW_W = 210, 160
FINISHED_RING_BYTES = 8 + 4096 * 4 + 4096 * 4


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, dtype, numel=None, name="tensor", optional=False):
    if t is None:
        if optional:
            return ctypes.c_void_p(0)
        raise ValueError("%s is required" % name)
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError("%s must be a CUDA/HIP torch tensor" % name)
    if t.dtype != dtype:
        raise ValueError("%s: dtype %s, expected %s" % (name, t.dtype, dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if numel is not None and t.numel() < numel:
        raise ValueError("%s: %d elements, need >= %d" % (name, t.numel(), numel))
    return ctypes.c_void_p(t.data_ptr())


class Context(object):
    """Owns one paac_ctx (activation workspace for one network on one GPU)."""

    def __init__(self, arch, num_actions, max_batch, device_index=0):
        self.lib = _lib.load()
        self.arch = int(arch)
        self.num_actions = int(num_actions)
        self.max_batch = int(max_batch)
        self.layout = _lib.param_layout(arch, num_actions)
        cfg = _lib.Cfg(device=int(device_index), arch=self.arch, num_actions=self.num_actions, max_batch=self.max_batch)
        h = ctypes.c_void_p()
        _lib.check(self.lib.paac_create(ctypes.byref(cfg), ctypes.byref(h)), "paac_create")
        self.handle = h
        self.device = torch.device("cuda", device_index)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.paac_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- network ---------------------------------------------------------------------------------
    def _check_states(self, states):
        B = states.shape[0]
        if tuple(states.shape[1:]) != OBS_SHAPE:
            raise ValueError("states must be [B,84,84,4] uint8, got %s" % (tuple(states.shape),))
        if not (0 < B <= self.max_batch):
            raise ValueError("batch %d outside (0, %d]" % (B, self.max_batch))
        return B

    def forward(self, params, states, logits=None, probs=None, values=None):
        B = self._check_states(states)
        A = self.num_actions
        _lib.check(self.lib.paac_forward(self.handle, _ptr(params, torch.float32, self.layout["total"], "params"),
                                         _ptr(states, torch.uint8, B * 28224, "states"), B,
                                         _ptr(logits, torch.float32, B * A, "logits", True),
                                         _ptr(probs, torch.float32, B * A, "probs", True),
                                         _ptr(values, torch.float32, B, "values", True), _stream()), "paac_forward")

    def forward_sample(self, params, states, seed, step_base_dev, step_offset, env_offset, actions, probs=None,
                       values=None):
        B = states.shape[0]
        if tuple(states.shape[1:]) != OBS_SHAPE:
            raise ValueError("states must be [B,84,84,4] uint8, got %s" % (tuple(states.shape),))
        if not (0 < B <= self.max_batch):
            raise ValueError("batch %d outside (0, %d]" % (B, self.max_batch))
        A = self.num_actions
        _lib.check(self.lib.paac_forward_sample(self.handle, _ptr(params, torch.float32, self.layout["total"], "params"),
                                                _ptr(states, torch.uint8, B * 28224, "states"), B,
                                                _ptr(probs, torch.float32, B * A, "probs", True),
                                                _ptr(values, torch.float32, B, "values", True), int(seed),
                                                _ptr(step_base_dev, torch.int64, 1, "step_base", True), int(step_offset),
                                                int(env_offset), _ptr(actions, torch.int32, B, "actions"), _stream()),
                   "paac_forward_sample")

    def forward_sample_synth_step(self, params, states, seed, step_base_dev, step_offset, env_offset, actions, env_seed,
                                  terminal_threshold, stack_out, rewards_out, masks_out, ep_reward, ep_len, finished=None,
                                  probs=None, values=None):
        """forward + counter-based sampler + synthetic env step (path A) in the forward's five launches."""
        B = states.shape[0]
        A = self.num_actions
        if tuple(states.shape[1:]) != OBS_SHAPE or tuple(stack_out.shape) != tuple(states.shape):
            raise ValueError("states / stack_out must be [B,84,84,4] uint8, got %s / %s" %
                             (tuple(states.shape), tuple(stack_out.shape)))
        if not (0 < B <= self.max_batch):
            raise ValueError("batch %d outside (0, %d]" % (B, self.max_batch))
        if states.data_ptr() == stack_out.data_ptr():
            raise ValueError("the step cannot shift the stacks in place")
        if finished is not None and finished.numel() * finished.element_size() < FINISHED_RING_BYTES:
            raise ValueError("finished ring too small")
        _lib.check(self.lib.paac_forward_sample_synth_step(
            self.handle, _ptr(params, torch.float32, self.layout["total"], "params"),
            _ptr(states, torch.uint8, B * 28224, "states"), B, _ptr(probs, torch.float32, B * A, "probs", True),
            _ptr(values, torch.float32, B, "values", True), int(seed), _ptr(step_base_dev, torch.int64, 1, "step_base", True),
            int(step_offset), int(env_offset), _ptr(actions, torch.int32, B, "actions"), int(env_seed),
            int(terminal_threshold), _ptr(stack_out, torch.uint8, B * 28224, "stack_out"),
            _ptr(rewards_out, torch.float32, B, "rewards_out"), _ptr(masks_out, torch.float32, B, "masks_out"),
            _ptr(ep_reward, torch.float32, B, "ep_reward"), _ptr(ep_len, torch.int32, B, "ep_len"),
            ctypes.c_void_p(finished.data_ptr()) if finished is not None else ctypes.c_void_p(0), _stream()),
            "paac_forward_sample_synth_step")

    def train_forward(self, params, states, values=None):
        B = states.shape[0]
        if tuple(states.shape[1:]) != OBS_SHAPE:
            raise ValueError("states must be [B,84,84,4] uint8, got %s" % (tuple(states.shape),))
        if not (0 < B <= self.max_batch):
            raise ValueError("batch %d outside (0, %d]" % (B, self.max_batch))
        _lib.check(self.lib.paac_train_forward(self.handle, _ptr(params, torch.float32, self.layout["total"], "params"),
                                               _ptr(states, torch.uint8, B * 28224, "states"), B,
                                               _ptr(values, torch.float32, B, "values", True), _stream()),
                   "paac_train_forward")

    def train_forward_trunk(self, params, states):
        """Training forward without the heads: the next loss_backward[_returns](forward_done=True) finishes them, inside
        its first launch where it can (include/paac_hip.h: paac_train_forward_trunk)."""
        B = self._check_states(states)
        _lib.check(self.lib.paac_train_forward_trunk(self.handle, _ptr(params, torch.float32, self.layout["total"], "params"),
                                                     _ptr(states, torch.uint8, B * 28224, "states"), B, _stream()),
                   "paac_train_forward_trunk")

    def loss_backward(self, params, states, actions, y, adv, entropy_beta, grad, loss_out=None, forward_done=False,
                      phase=0):
        B = states.shape[0]
        if tuple(states.shape[1:]) != OBS_SHAPE:
            raise ValueError("states must be [B,84,84,4] uint8, got %s" % (tuple(states.shape),))
        if not (0 < B <= self.max_batch):
            raise ValueError("batch %d outside (0, %d]" % (B, self.max_batch))
        _lib.check(self.lib.paac_loss_backward(self.handle, _ptr(params, torch.float32, self.layout["total"], "params"),
                                               _ptr(states, torch.uint8, B * 28224, "states"),
                                               _ptr(actions, torch.int32, B, "actions"),
                                               _ptr(y, torch.float32, B, "y"), _ptr(adv, torch.float32, B, "adv"), B,
                                               float(entropy_beta),
                                               _ptr(grad, torch.float32, self.layout["total"], "grad"),
                                               _ptr(loss_out, torch.float32, 4, "loss_out", True),
                                               1 if forward_done else 0, int(phase), _stream()),
                   "paac_loss_backward")

    def loss_backward_returns(self, params, states, actions, v_boot, rewards, masks, values, gamma, y_out, adv_out,
                              entropy_beta, grad, loss_out=None, forward_done=False, phase=0, global_step_dev=None,
                              increment=0, initial_lr=0.0, lr_annealing_steps=1, lr_out_dev=None, tick_dev=None, tick_inc=0):
        """n-step returns (+ the cycle's schedule bookkeeping) inside the backward's first launch
        (include/paac_hip.h: paac_loss_backward_returns) == nstep_returns_tick followed by loss_backward.
        v_boot=None: the bootstrap values are rows [B, B + N) of the training forward that has already run."""
        B = self._check_states(states)
        T, N = rewards.shape
        if T * N != B:
            raise ValueError("rollout records are [%d,%d] but the batch has %d rows" % (T, N, B))
        ret = _lib.Returns(
            v_boot=_ptr(v_boot, torch.float32, N, "v_boot", True), rewards=_ptr(rewards, torch.float32, B, "rewards"),
            masks=_ptr(masks, torch.float32, B, "masks"), values=_ptr(values, torch.float32, B, "values"), T=T, N=N,
            gamma=float(gamma), y_out=_ptr(y_out, torch.float32, B, "y_out"), adv_out=_ptr(adv_out, torch.float32, B, "adv_out"),
            global_step_dev=_ptr(global_step_dev, torch.int64, 1, "global_step", True), increment=int(increment),
            initial_lr=float(initial_lr), lr_annealing_steps=int(lr_annealing_steps),
            lr_out_dev=_ptr(lr_out_dev, torch.float32, 1, "lr_out", True),
            tick_dev=_ptr(tick_dev, torch.int64, 1, "tick", True), tick_inc=int(tick_inc))
        _lib.check(self.lib.paac_loss_backward_returns(
            self.handle, _ptr(params, torch.float32, self.layout["total"], "params"),
            _ptr(states, torch.uint8, B * 28224, "states"), _ptr(actions, torch.int32, B, "actions"), ctypes.byref(ret), B,
            float(entropy_beta), _ptr(grad, torch.float32, self.layout["total"], "grad"),
            _ptr(loss_out, torch.float32, 4, "loss_out", True), 1 if forward_done else 0, int(phase), _stream()),
            "paac_loss_backward_returns")

    def clip_rmsprop(self, params, grad, ms, mom, lr_dev, decay, momentum, eps, clip_norm, clip_mode, grad_scale=1.0,
                     gnorm_out=None):
        n = self.layout["total"]
        _lib.check(self.lib.paac_clip_rmsprop(self.handle, _ptr(params, torch.float32, n, "params"),
                                              _ptr(grad, torch.float32, n, "grad"), _ptr(ms, torch.float32, n, "ms"),
                                              _ptr(mom, torch.float32, n, "mom"), n,
                                              _ptr(lr_dev, torch.float32, 1, "lr_dev"), float(decay), float(momentum),
                                              float(eps), float(clip_norm), int(clip_mode), float(grad_scale),
                                              _ptr(gnorm_out, torch.float32, 1, "gnorm_out", True), _stream()),
                   "paac_clip_rmsprop")

    def keep_next_forward(self, train_row):
        """The next acting forward also leaves its rows' activations at rows [train_row, train_row + batch) of the training
        activation set (include/paac_hip.h: paac_keep_next_forward); -1 cancels."""
        _lib.check(self.lib.paac_keep_next_forward(self.handle, int(train_row)), "paac_keep_next_forward")

    def bootstrap_forward_trunk(self, params, states, train_row):
        """Acting-shaped forward (conv tower + fc) of the bootstrap observations, kept at rows [train_row, ...) of the training
        set the acting steps have filled: the next loss_backward[_returns](forward_done=True) needs no training forward."""
        B = self._check_states(states)
        _lib.check(self.lib.paac_bootstrap_forward_trunk(self.handle, _ptr(params, torch.float32, self.layout["total"], "params"),
                                                         _ptr(states, torch.uint8, B * 28224, "states"), B, int(train_row), _stream()),
                   "paac_bootstrap_forward_trunk")

    def act_step_mt(self, params, states, mt_state, actions, probs_out, values_out, env_seed, env_offset,
                    terminal_threshold, step_base_dev, step_offset, stack_out, rewards_out, masks_out, ep_reward, ep_len,
                    finished=None, stack_out2=None, raw_scratch=None, walk_scratch=None):
        """One acting step in three launches: policy forward, then heads finish + numpy-parity sampler + synthetic
        environment step in one (include/paac_hip.h: paac_act_step_mt).  raw_scratch ([N,2,210,160] u8): path B -- the
        step launch writes the raw screen pairs, a fourth launch (max, PIL-nearest resize, history push) builds the stacks."""
        N, A = self._check_states(states), self.num_actions
        if N > ACT_STEP_MAX_ENVS_LARGE or N * (A - 1) > FUSED_SAMPLE_MAX_DRAWS:
            raise ValueError("act_step_mt supports N <= %d and N*(A-1) <= %d" % (ACT_STEP_MAX_ENVS_LARGE, FUSED_SAMPLE_MAX_DRAWS))
        if tuple(stack_out.shape) != (N,) + OBS_SHAPE:
            raise ValueError("stack_out must be [%d,84,84,4], got %s" % (N, tuple(stack_out.shape)))
        if states.data_ptr() == stack_out.data_ptr() or (stack_out2 is not None and states.data_ptr() == stack_out2.data_ptr()):
            raise ValueError("the step cannot shift the stacks in place")
        if finished is not None and finished.numel() * finished.element_size() < FINISHED_RING_BYTES:
            raise ValueError("finished ring too small")
        _lib.check(self.lib.paac_act_step_mt(
            self.handle, _ptr(params, torch.float32, self.layout["total"], "params"),
            _ptr(states, torch.uint8, N * 28224, "states"), N, _ptr(mt_state, torch.int32, 625, "mt_state"),
            _ptr(actions, torch.int32, N, "actions"), _ptr(probs_out, torch.float32, N * A, "probs_out"),
            _ptr(values_out, torch.float32, N, "values_out"), int(env_seed), int(env_offset), int(terminal_threshold),
            _ptr(step_base_dev, torch.int64, 1, "step_base", True), int(step_offset),
            _ptr(stack_out, torch.uint8, N * 28224, "stack_out"), _ptr(stack_out2, torch.uint8, N * 28224, "stack_out2", True),
            _ptr(rewards_out, torch.float32, N, "rewards_out"),
            _ptr(masks_out, torch.float32, N, "masks_out"), _ptr(ep_reward, torch.float32, N, "ep_reward"),
            _ptr(ep_len, torch.int32, N, "ep_len"),
            ctypes.c_void_p(finished.data_ptr()) if finished is not None else ctypes.c_void_p(0),
            _ptr(raw_scratch, torch.uint8, N * 2 * RAW_H * RAW_W, "raw_scratch", True),
            ctypes.c_void_p(walk_scratch.data_ptr()) if walk_scratch is not None else ctypes.c_void_p(0),
            int(walk_scratch.numel()) if walk_scratch is not None else 0, _stream()),
            "paac_act_step_mt")

    def act_mt(self, params, states, mt_state, actions, probs_out, values_out):
        """Policy forward + numpy-parity sampler in three launches, no environment step (paac_act_step_mt with stack_out =
        NULL): what the host-plugin loop runs per step (paac.py:104-110).  N <= ACT_STEP_MAX_ENVS, N*(A-1) <= 1024."""
        N, A = self._check_states(states), self.num_actions
        z = ctypes.c_void_p(0)
        _lib.check(self.lib.paac_act_step_mt(
            self.handle, _ptr(params, torch.float32, self.layout["total"], "params"),
            _ptr(states, torch.uint8, N * 28224, "states"), N, _ptr(mt_state, torch.int32, 625, "mt_state"),
            _ptr(actions, torch.int32, N, "actions"), _ptr(probs_out, torch.float32, N * A, "probs_out"),
            _ptr(values_out, torch.float32, N, "values_out"), 0, 0, 0, z, 0, z, z, z, z, z, z, z, z, z, 0, _stream()),
            "paac_act_step_mt")

    def pack_weights(self, params):
        """Refresh the ctx's pre-split copy of the conv weights (include/paac_hip.h: paac_pack_weights)."""
        _lib.check(self.lib.paac_pack_weights(self.handle, _ptr(params, torch.float32, self.layout["total"], "params"),
                                              _stream()), "paac_pack_weights")

    def set_managed_weights(self, on=True):
        """Managed mode: only clip_rmsprop / pack_weights refresh the pre-split copy; acting forwards keep no conv1 /
        conv2 activations (include/paac_hip.h: paac_set_managed_weights)."""
        _lib.check(self.lib.paac_set_managed_weights(self.handle, 1 if on else 0), "paac_set_managed_weights")

    def grad_stats(self, clip_norm, clip_mode):
        """Gradient summaries of the last clip_rmsprop (actor_learner.py:85-87, logger_utils.py:23-33): dict with
        mean / stddev / max / min of the raw and of the clipped flat gradient, and global_norm.  Synchronises."""
        out = torch.zeros(8, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.paac_grad_stats(self.handle, ctypes.c_void_p(out.data_ptr()), _stream()), "paac_grad_stats")
        s, ss, mx, mn = [float(v) for v in out.cpu().numpy()[:4].astype(np.float64)]
        n = float(self.layout["total_unpadded"])
        mean = s / n
        std = max(ss / n - mean * mean, 0.0) ** 0.5
        gn = ss ** 0.5
        f = 1.0
        if clip_mode == _lib.CLIP_GLOBAL and gn > 0.0:
            f = clip_norm * min(1.0 / gn, 1.0 / clip_norm)
        return {"global_norm": gn,
                "raw_gradients": {"mean": mean, "stddev": std, "max": mx, "min": mn},
                "clipped_gradients": {"mean": mean * f, "stddev": std * f, "max": mx * f, "min": mn * f}}

    def debug_activation(self, what, batch):
        out = torch.empty(batch * 20 * 20 * 64, dtype=torch.float32, device=self.device)
        n = self.lib.paac_debug_activation(self.handle, int(what), int(batch), ctypes.c_void_p(out.data_ptr()), _stream())
        _lib.check(n, "paac_debug_activation")
        return out[:n].clone()

    # -- timing hooks ----------------------------------------------------------------------------
    def prof_enable(self, on=True):
        _lib.check(self.lib.paac_prof_enable(self.handle, 1 if on else 0), "paac_prof_enable")

    def prof_read(self, with_mix=False):
        """-> list of (family name, batch, milliseconds), one per kernel-family launch since the last read; with_mix adds
        the launch's instruction mix as a tuple of MFMA products per fp32 multiply, one entry per contraction body
        (include/paac_hip.h: paac_prof_read_mix)."""
        cap = 8192
        fam = (ctypes.c_int32 * cap)()
        bat = (ctypes.c_int32 * cap)()
        ms = (ctypes.c_float * cap)()
        mix = (ctypes.c_int32 * cap)()
        if with_mix:
            _lib.check(self.lib.paac_prof_read_mix(self.handle, mix, cap), "paac_prof_read_mix")
        n = self.lib.paac_prof_read(self.handle, fam, bat, ms, cap)
        _lib.check(n, "paac_prof_read")
        out = [(self.lib.paac_prof_name(fam[i]).decode(), int(bat[i]), float(ms[i])) for i in range(n)]
        if with_mix:
            out = [o + (tuple(b for b in ((mix[i] >> (8 * k)) & 255 for k in range(4)) if b),) for i, o in enumerate(out)]
        return out


# -- context-free entry points -------------------------------------------------------------------
def lr_step(global_step_dev, increment, initial_lr, lr_annealing_steps, lr_out_dev):
    lib = _lib.load()
    _lib.check(lib.paac_lr_step(_ptr(global_step_dev, torch.int64, 1, "global_step"), int(increment), float(initial_lr),
                                int(lr_annealing_steps), _ptr(lr_out_dev, torch.float32, 1, "lr_out"), _stream()),
               "paac_lr_step")


def nstep_returns(v_boot, rewards, masks, values, gamma, y, adv):
    T, N = rewards.shape
    lib = _lib.load()
    _lib.check(lib.paac_nstep_returns(_ptr(v_boot, torch.float32, N, "v_boot"), _ptr(rewards, torch.float32, T * N, "rewards"),
                                      _ptr(masks, torch.float32, T * N, "masks"), _ptr(values, torch.float32, T * N, "values"),
                                      T, N, float(gamma), _ptr(y, torch.float32, T * N, "y"),
                                      _ptr(adv, torch.float32, T * N, "adv"), _stream()), "paac_nstep_returns")


def nstep_returns_tick(v_boot, rewards, masks, values, gamma, y, adv, global_step_dev, increment, initial_lr,
                       lr_annealing_steps, lr_out_dev, tick_dev=None, tick_inc=0):
    T, N = rewards.shape
    lib = _lib.load()
    _lib.check(lib.paac_nstep_returns_tick(_ptr(v_boot, torch.float32, N, "v_boot"), _ptr(rewards, torch.float32, T * N, "rewards"),
                                           _ptr(masks, torch.float32, T * N, "masks"), _ptr(values, torch.float32, T * N, "values"),
                                           T, N, float(gamma), _ptr(y, torch.float32, T * N, "y"),
                                           _ptr(adv, torch.float32, T * N, "adv"),
                                           _ptr(global_step_dev, torch.int64, 1, "global_step"), int(increment),
                                           float(initial_lr), int(lr_annealing_steps),
                                           _ptr(lr_out_dev, torch.float32, 1, "lr_out"),
                                           _ptr(tick_dev, torch.int64, 1, "tick", True), int(tick_inc), _stream()),
               "paac_nstep_returns_tick")


def sample_mt_scratch(N, A, device):
    nbytes = _lib.load().paac_sample_mt_scratch_bytes(int(N), int(A))
    return torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=device)


def sample_mt(probs, mt_state, scratch, actions):
    N, A = probs.shape
    lib = _lib.load()
    need = lib.paac_sample_mt_scratch_bytes(int(N), int(A))
    if scratch.numel() * scratch.element_size() < need:
        raise ValueError("sample_mt scratch too small: %d < %d bytes" % (scratch.numel() * scratch.element_size(), need))
    _lib.check(lib.paac_sample_mt(_ptr(probs, torch.float32, N * A, "probs"), N, A,
                                  _ptr(mt_state, torch.int32, 625, "mt_state"), ctypes.c_void_p(scratch.data_ptr()),
                                  _ptr(actions, torch.int32, N, "actions"), _stream()), "paac_sample_mt")


def mt_state_from_numpy(state, device):
    """np.random.get_state() tuple -> device int32[625] (key + pos)."""
    assert state[0] == "MT19937"
    arr = np.concatenate([np.asarray(state[1], dtype=np.uint32), np.array([state[2]], dtype=np.uint32)])
    return torch.from_numpy(arr.view(np.int32).copy()).to(device)


def mt_state_to_numpy(mt_state):
    arr = mt_state.detach().cpu().numpy().view(np.uint32)
    return ("MT19937", arr[:624].copy(), int(arr[624]), 0, 0.0)


def sample_philox(probs, seed, step_base_dev, step_offset, env_offset, actions):
    N, A = probs.shape
    lib = _lib.load()
    _lib.check(lib.paac_sample_philox(_ptr(probs, torch.float32, N * A, "probs"), N, A, int(seed),
                                      _ptr(step_base_dev, torch.int64, 1, "step_base", True), int(step_offset),
                                      int(env_offset), _ptr(actions, torch.int32, N, "actions"), _stream()),
               "paac_sample_philox")


def debug_clock(out2_dev):
    _lib.check(_lib.load().paac_debug_clock(_ptr(out2_dev, torch.int64, 2, "out2"), _stream()), "paac_debug_clock")


def counter_add(counter_dev, inc):
    _lib.check(_lib.load().paac_counter_add(_ptr(counter_dev, torch.int64, 1, "counter"), int(inc), _stream()),
               "paac_counter_add")


def preprocess_stack(raw, stack_in, stack_out, push_mask=None, reset_mask=None):
    N = raw.shape[0]
    if tuple(raw.shape) == (N, 2, RAW_H, RAW_W):
        rgb = 0
    elif tuple(raw.shape) == (N, 2, RAW_H, RAW_W, 3):
        rgb = 1
    else:
        raise ValueError("raw must be [N,2,210,160] or [N,2,210,160,3] uint8, got %s" % (tuple(raw.shape),))
    for nm, t in (("stack_in", stack_in), ("stack_out", stack_out)):
        if tuple(t.shape) != (N,) + OBS_SHAPE:
            raise ValueError("%s must be [%d,84,84,4], got %s" % (nm, N, tuple(t.shape)))
    _lib.check(_lib.load().paac_preprocess_stack(_ptr(raw, torch.uint8, None, "raw"), rgb, N,
                                                 _ptr(stack_in, torch.uint8, N * 28224, "stack_in"),
                                                 _ptr(stack_out, torch.uint8, N * 28224, "stack_out"),
                                                 _ptr(push_mask, torch.uint8, N, "push_mask", True),
                                                 _ptr(reset_mask, torch.uint8, N, "reset_mask", True), _stream()),
               "paac_preprocess_stack")


def synth_reset(seed, env_offset, stack_out, raw_scratch=None):
    N = stack_out.shape[0]
    if tuple(stack_out.shape) != (N,) + OBS_SHAPE:
        raise ValueError("stack_out must be [N,84,84,4]")
    _lib.check(_lib.load().paac_synth_reset(int(seed), int(env_offset), N, _ptr(stack_out, torch.uint8, N * 28224, "stack_out"),
                                            _ptr(raw_scratch, torch.uint8, N * 2 * RAW_H * RAW_W, "raw_scratch", True),
                                            _stream()), "paac_synth_reset")


def synth_step(seed, env_offset, actions, terminal_threshold, step_base_dev, step_offset, stack_in, stack_out,
               rewards_out, masks_out, ep_reward, ep_len, finished=None, stack_out2=None, raw_scratch=None):
    N = actions.shape[0]
    for nm, t in (("stack_in", stack_in), ("stack_out", stack_out)):
        if tuple(t.shape) != (N,) + OBS_SHAPE:
            raise ValueError("%s must be [%d,84,84,4], got %s" % (nm, N, tuple(t.shape)))
    if finished is not None and finished.numel() * finished.element_size() < FINISHED_RING_BYTES:
        raise ValueError("finished ring too small")
    _lib.check(_lib.load().paac_synth_step(int(seed), int(env_offset), N, _ptr(actions, torch.int32, N, "actions"),
                                           int(terminal_threshold), _ptr(step_base_dev, torch.int64, 1, "step_base", True),
                                           int(step_offset), _ptr(stack_in, torch.uint8, N * 28224, "stack_in"),
                                           _ptr(stack_out, torch.uint8, N * 28224, "stack_out"),
                                           _ptr(stack_out2, torch.uint8, N * 28224, "stack_out2", True),
                                           _ptr(rewards_out, torch.float32, N, "rewards_out"),
                                           _ptr(masks_out, torch.float32, N, "masks_out"),
                                           _ptr(ep_reward, torch.float32, N, "ep_reward"),
                                           _ptr(ep_len, torch.int32, N, "ep_len"),
                                           ctypes.c_void_p(finished.data_ptr()) if finished is not None else ctypes.c_void_p(0),
                                           _ptr(raw_scratch, torch.uint8, N * 2 * RAW_H * RAW_W, "raw_scratch", True),
                                           _stream()), "paac_synth_step")


FUSED_SAMPLE_MAX_DRAWS = 2304
ACT_STEP_MAX_DRAWS = 1024
ACT_STEP_MAX_ENVS = 64
ACT_STEP_MAX_ENVS_LARGE = 256    # paac_act_step_mt's four-launch form (large shards)
KEEP_FORWARD_MAX_ROWS = 256      # paac_keep_next_forward: acting forwards of up to this many rows (csrc/fc_heads.h)


def walk_scratch(N, A, device):
    """Zero-initialised scratch that lets the large shards' sampler spread its walk over several workgroups
    (include/paac_hip.h: paac_sample_mt_synth_step); lend the same tensor to every call of one (N, A)."""
    return torch.zeros(int(_lib.load().paac_walk_scratch_bytes(int(N), int(A))), dtype=torch.uint8, device=device)


def sample_mt_synth_step(probs, mt_state, actions, seed, env_offset, terminal_threshold, step_base_dev, step_offset,
                         stack_in, stack_out, rewards_out, masks_out, ep_reward, ep_len, finished=None, stack_out2=None,
                         walk_scratch=None, raw_scratch=None):
    N, A = probs.shape
    if N * (A - 1) > FUSED_SAMPLE_MAX_DRAWS:
        raise ValueError("fused sampler+env step supports N*(A-1) <= %d" % FUSED_SAMPLE_MAX_DRAWS)
    for nm, t in (("stack_in", stack_in), ("stack_out", stack_out)):
        if tuple(t.shape) != (N,) + OBS_SHAPE:
            raise ValueError("%s must be [%d,84,84,4], got %s" % (nm, N, tuple(t.shape)))
    if finished is not None and finished.numel() * finished.element_size() < FINISHED_RING_BYTES:
        raise ValueError("finished ring too small")
    _lib.check(_lib.load().paac_sample_mt_synth_step(
        _ptr(probs, torch.float32, N * A, "probs"), A, _ptr(mt_state, torch.int32, 625, "mt_state"),
        _ptr(actions, torch.int32, N, "actions"), int(seed), int(env_offset), N, int(terminal_threshold),
        _ptr(step_base_dev, torch.int64, 1, "step_base", True), int(step_offset),
        _ptr(stack_in, torch.uint8, N * 28224, "stack_in"), _ptr(stack_out, torch.uint8, N * 28224, "stack_out"),
        _ptr(stack_out2, torch.uint8, N * 28224, "stack_out2", True),
        _ptr(rewards_out, torch.float32, N, "rewards_out"), _ptr(masks_out, torch.float32, N, "masks_out"),
        _ptr(ep_reward, torch.float32, N, "ep_reward"), _ptr(ep_len, torch.int32, N, "ep_len"),
        ctypes.c_void_p(finished.data_ptr()) if finished is not None else ctypes.c_void_p(0),
        ctypes.c_void_p(walk_scratch.data_ptr()) if walk_scratch is not None else ctypes.c_void_p(0),
        int(walk_scratch.numel()) if walk_scratch is not None else 0,
        _ptr(raw_scratch, torch.uint8, N * 2 * RAW_H * RAW_W, "raw_scratch", True), _stream()),
        "paac_sample_mt_synth_step")


def pin_host_array(t):
    """Page-lock the memory of a CPU tensor in place (hipHostRegister through torch's runtime handle) so that copies to
    the device are asynchronous DMAs.  Returns True when the registration succeeded; a failure only costs speed."""
    try:
        rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel() * t.element_size(), 0)
        return int(rc) == 0
    except Exception:
        return False


class Graph(object):
    """hipGraph captured from the launches issued on torch's current stream between begin() and end()."""

    def __init__(self):
        self.lib = _lib.load()
        self.handle = None

    def begin(self):
        # A finalizer that frees device memory (a collected Context or tensor) in the middle of a capture
        # invalidates it: collect now and keep the collector off until end().
        gc.collect()
        self._gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            _lib.check(self.lib.paac_graph_begin(_stream()), "paac_graph_begin")
        except Exception:
            self._restore_gc()
            raise

    def _restore_gc(self):
        if getattr(self, "_gc_was_enabled", False):
            gc.enable()
        self._gc_was_enabled = False

    def end(self):
        h = ctypes.c_void_p()
        try:
            _lib.check(self.lib.paac_graph_end(_stream(), ctypes.byref(h)), "paac_graph_end")
        finally:
            self._restore_gc()
        self.handle = h

    def abort(self):
        """Leave capture mode after a failed capture, dropping whatever was recorded."""
        h = ctypes.c_void_p()
        try:
            if self.lib.paac_graph_end(_stream(), ctypes.byref(h)) == 0 and h:
                self.lib.paac_graph_destroy(h)
        finally:
            self._restore_gc()

    def launch(self):
        _lib.check(self.lib.paac_graph_launch(self.handle, _stream()), "paac_graph_launch")

    def close(self):
        if self.handle:
            self.lib.paac_graph_destroy(self.handle)
            self.handle = None
