"""Data parallelism of the PAAC hot path: environments are sharded contiguously over
ranks (one process per GPU, torch.distributed, backend "nccl" = RCCL over xGMI), every rank runs its own
rollout / forward / backward on its shard, and ONE sum all-reduce of the flat fp32 gradient per update --
before global-norm clipping, like the reference clips the full-batch gradient (actor_learner.py:56-59) --
keeps the replicated weights identical.  The reference loss is a mean over the batch
(policy_v_network.py:49-53), so with equal shards the global gradient is (1/G) * sum_r grad_r; the 1/G is
folded into paac_clip_rmsprop's grad_scale.  Nothing else is exchanged on the data path.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(args=None):
    """One process per GPU under `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment): picks this rank's GPU, rewrites args.device to '/gpu:<LOCAL_RANK>' and joins the process group --
    backend "nccl" (= RCCL over xGMI) bound to that GPU -- BEFORE anything else touches the device.  A plain
    `python -m paac_amd.train` (no WORLD_SIZE) stays a single process with no process group.  Returns the world size.

    Knobs (tests and one-GPU rehearsals only): PAAC_DIST_BACKEND=gloo exchanges through the host instead of RCCL;
    PAAC_DIST_SINGLE_DEVICE=1 puts every rank on GPU 0; PAAC_DIST_FORCE=1 joins a group even at WORLD_SIZE=1 (with
    PAAC_FORCE_COLLECTIVES=1 the exchange then really issues its collectives, see `collectives_active`)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    forced = os.environ.get("PAAC_DIST_FORCE", "") == "1"
    if world <= 1 and not forced:
        return 1
    if dist.is_initialized():
        return dist.get_world_size()
    local_rank = 0 if os.environ.get("PAAC_DIST_SINGLE_DEVICE", "") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("PAAC_DIST_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if args is not None:
        args.device = "/gpu:%d" % local_rank          # train.py:80 syntax; emulator_counts stays per GPU (weak scaling)
    torch.cuda.set_device(local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist.get_world_size()


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def collectives_active():
    """True when the gradient exchange has to issue collectives: more than one rank, or a joined group with
    PAAC_FORCE_COLLECTIVES=1 (the one-GPU RCCL smoke: a world of one still runs the stream-ordered all-reduce calls)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("PAAC_FORCE_COLLECTIVES", "") == "1"


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def broadcast_(tensor, src=0):
    """In-place broadcast from rank `src` (replicas start from rank 0's weights and optimizer slots)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(tensor, src=src)
    return tensor


def any_rank(flag, device):
    """Logical OR of a host flag over all ranks (a stop request must be honoured by every rank at the same cycle, or
    the ranks would disagree on the number of gradient exchanges and hang)."""
    if world_size() <= 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32,
                     device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def all_ranks(flag, device):
    """Logical AND of a host flag over all ranks (a fallback decision every rank must take together)."""
    if world_size() <= 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32,
                     device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


_weights = {}


def checksum64(t):
    """Position-weighted 64-bit checksum of a tensor's bits (device scalar, wraps modulo 2^64): equal tensors give equal sums,
    a flipped bit or two swapped elements change it."""
    bits = t.detach().contiguous().view(-1)
    bits = bits.view(torch.int32) if bits.element_size() == 4 else bits.to(torch.int32)
    key = (bits.numel(), bits.device)
    if key not in _weights:
        _weights[key] = (torch.arange(bits.numel(), dtype=torch.int64, device=bits.device) % 1000003) * 2 + 1
    return (bits.to(torch.int64) * _weights[key]).sum()


def replicas_identical(tensors):
    """Do all ranks hold bit-identical copies of `tensors`?  MIN and MAX all-reduce of their checksums; every rank gets the
    same answer.  -> (identical, [per-tensor bool]).  One process: trivially true."""
    sums = torch.stack([checksum64(t) for t in tensors])
    if world_size() <= 1:
        return True, [True] * len(tensors)
    if dist.get_backend() != "nccl":
        sums = sums.cpu()
    lo, hi = sums.clone(), sums.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same = (lo == hi).cpu().tolist()
    return all(same), same


class ReplicaMismatch(RuntimeError):
    """The data-parallel replicas no longer hold identical weights: the run cannot continue."""


def backend():
    """Backend of the default process group ("nccl" = RCCL, "gloo"), None without a group."""
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_range(total_envs, r=None, world=None):
    """Contiguous shard [lo, hi) of `total_envs` environments owned by rank r (equal shards required)."""
    world = world_size() if world is None else world
    r = rank() if r is None else r
    if total_envs % world != 0:
        raise ValueError("emulator count %d is not divisible by the world size %d" % (total_envs, world))
    per = total_envs // world
    return r * per, (r + 1) * per


def allreduce_sum_(flat_grad, group=None):
    """In-place sum all-reduce of the flat gradient (no-op for a single process).  With RCCL a blocking-style call
    (async_op=False) is launched on the CURRENT stream: it is ordered with the kernels around it like any other launch,
    without the two cross-stream event waits of an asynchronous collective."""
    if collectives_active():
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


_side_group = None


def side_group():
    """A second process group over the same ranks (collective call: every rank must make it at the same point).
    Its collectives have their own communicator and stream, so a small all-reduce issued on it does not queue behind a
    large one still in flight on the default group."""
    global _side_group
    if _side_group is None and collectives_active():
        _side_group = dist.new_group(ranks=list(range(world_size())))
    return _side_group


def allreduce_sum_async(flat_grad, group=None):
    """Start an in-place sum all-reduce; returns the work handle (None for a single process).  With RCCL the
    collective runs on the process group's stream after the work already queued on the current stream, and
    `work.wait()` makes the current stream wait for it -- no host synchronisation."""
    if collectives_active():
        return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


def grad_scale():
    return 1.0 / world_size()


def global_steps_per_cycle(envs_per_rank, t_max):
    """global_step advances by one per environment per step (paac.py:127), over ALL ranks."""
    return envs_per_rank * t_max * world_size()
