"""world_size-2 gloo test (CPU): sharding + one sum all-reduce of the flat gradient + 1/G scale reproduces the
single-process gradient of the concatenated batch, and replicated TF-RMSProp steps stay
bit-identical across ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import network as onet
    from paac_amd import parallel
    arch, A, N, T = "NIPS", 4, 8, 2
    rs = np.random.RandomState(0)
    params = onet.init_params(arch, A, rs, dtype=np.float64)
    B = N * T
    states = rs.randint(0, 256, (T, N, 84, 84, 4)).astype(np.uint8)
    idx = rs.randint(0, A, (T, N))
    y = rs.randn(T, N)
    adv = rs.randn(T, N)
    lo, hi = parallel.shard_range(N)
    assert (lo, hi) == (rank * N // world, (rank + 1) * N // world)
    assert parallel.global_steps_per_cycle(hi - lo, T) == N * T
    sh = lambda a: a[:, lo:hi].reshape((T * (hi - lo),) + a.shape[2:])
    _, g = onet.loss_and_grads(params, sh(states), np.eye(A)[sh(idx)], sh(y), sh(adv), 0.02, arch)
    flat = torch.from_numpy(onet.flatten_params(g, arch, A).copy())
    parallel.allreduce_sum_(flat)
    flat *= parallel.grad_scale()
    # replicated optimizer step
    gd = {}
    off = 0
    for name, shape in onet.param_shapes(arch, A):
        n = int(np.prod(shape))
        gd[name] = flat.numpy()[off:off + n].reshape(shape)
        off += n
    gc, gn = onet.clip_by_global_norm(gd, 3.0)
    ms, mom = onet.rmsprop_init(params)
    p2, _, _ = onet.rmsprop_step({k: v.copy() for k, v in params.items()}, gc, ms, mom, 0.0224)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), flat=flat.numpy(), gn=gn, w=onet.flatten_params(p2, arch, A))
    if rank == 0:
        full = lambda a: a.reshape((T * N,) + a.shape[2:])
        _, gf = onet.loss_and_grads(params, full(states), np.eye(A)[full(idx)], full(y), full(adv), 0.02, arch)
        np.savez(os.path.join(out_dir, "full.npz"), flat=onet.flatten_params(gf, arch, A))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gradient_equals_full_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    full = np.load(tmp_path / "full.npz")
    assert np.array_equal(r0["flat"], r1["flat"])              # all-reduce result identical on every rank
    assert np.array_equal(r0["w"], r1["w"])                    # so replicated weights stay bit-identical
    scale = np.abs(full["flat"]).max()
    assert np.abs(r0["flat"] - full["flat"]).max() / scale < 1e-12   # == gradient of the concatenated batch


def _checksum_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from paac_amd import parallel
    rs = np.random.RandomState(3)
    w = torch.from_numpy(rs.randn(100003).astype(np.float32))
    slots = torch.ones(777)
    res = [parallel.replicas_identical([w, slots])]
    if rank == 1:
        w[99999] = w[99999] + 1e-7 * (1 + abs(w[99999]))          # one element, the last bit or two
    res.append(parallel.replicas_identical([w, slots]))
    w2 = w.clone()
    if rank == 1:
        w2[5], w2[6] = w[6].clone(), w[5].clone()                 # two elements swapped: the checksum is position-weighted
    res.append(parallel.replicas_identical([slots, w2]))
    flags = [parallel.all_ranks(True, "cpu"), parallel.all_ranks(rank == 0, "cpu"), parallel.any_rank(rank == 1, "cpu")]
    np.save(os.path.join(out_dir, "cs%d.npy" % rank), np.array([str(res), str(flags)]))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_checksums_agree_on_every_rank(tmp_path):
    """parallel.replicas_identical / all_ranks: what keeps a diverged or half-fallen-back data-parallel run from going on."""
    world = 2
    mp.spawn(_checksum_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "cs0.npy"), np.load(tmp_path / "cs1.npy")
    assert list(a) == list(b)                                      # every rank reaches the same verdicts
    assert a[0] == str([(True, [True, True]), (False, [False, True]), (False, [True, False])])
    assert a[1] == str([True, False, True])


def test_shard_range_validation():
    from paac_amd import parallel
    assert parallel.shard_range(32, 3, 8) == (12, 16)
    with pytest.raises(ValueError):
        parallel.shard_range(30, 0, 8)
    assert parallel.world_size() == 1 and parallel.rank() == 0 and parallel.grad_scale() == 1.0
