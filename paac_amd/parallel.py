"""Data parallelism of the PAAC hot path: environments are sharded contiguously over
ranks (one process per GPU, torch.distributed, backend "nccl" = RCCL over xGMI), every rank runs its own
rollout / forward / backward on its shard, and ONE sum all-reduce of the flat fp32 gradient per update --
before global-norm clipping, like the reference clips the full-batch gradient (actor_learner.py:56-59) --
keeps the replicated weights identical.  The reference loss is a mean over the batch
(policy_v_network.py:49-53), so with equal shards the global gradient is (1/G) * sum_r grad_r; the 1/G is
folded into paac_clip_rmsprop's grad_scale.  Nothing else is exchanged on the data path.
"""
import torch
import torch.distributed as dist


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


def allreduce_sum_(flat_grad):
    """In-place sum all-reduce of the flat gradient (no-op for a single process)."""
    if world_size() > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


_side_group = None


def side_group():
    """A second process group over the same ranks (collective call: every rank must make it at the same point).
    Its collectives have their own communicator and stream, so a small all-reduce issued on it does not queue behind a
    large one still in flight on the default group."""
    global _side_group
    if _side_group is None and world_size() > 1:
        _side_group = dist.new_group(ranks=list(range(world_size())))
    return _side_group


def allreduce_sum_async(flat_grad, group=None):
    """Start an in-place sum all-reduce; returns the work handle (None for a single process).  With RCCL the
    collective runs on the process group's stream after the work already queued on the current stream, and
    `work.wait()` makes the current stream wait for it -- no host synchronisation."""
    if world_size() > 1:
        return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


def grad_scale():
    return 1.0 / world_size()


def global_steps_per_cycle(envs_per_rank, t_max):
    """global_step advances by one per environment per step (paac.py:127), over ALL ranks."""
    return envs_per_rank * t_max * world_size()
