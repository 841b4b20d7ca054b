"""Multi-process logic of the data-parallel trainer on CPU (gloo, world_size 2): the index
sharding and the two-bucket gradient all-reduce.  The device kernels are not involved -- the
optimiser step after the reduce is the oracle's Adam (test infrastructure only)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import st_oracle as orc
        from pca_hip.trainer import ShardedIndexStream, allreduce_buckets
        # ---- sharding: same permutation on all ranks, disjoint interleaved shares ----
        n, B = 1003, 50
        st = ShardedIndexStream(n, B, rank, world, seed=7, shuffle=True)
        mine = torch.cat([st.next() for _ in range((n // world) // B)])
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        allidx = torch.cat(gathered)
        ok_disjoint = allidx.unique().numel() == allidx.numel()
        g = torch.Generator().manual_seed(7)
        perm = torch.randperm(n, generator=g)
        ok_share = torch.equal(mine, perm[rank:(n // world) * world:world][:mine.numel()])
        e0 = st.epoch
        st.next()                                  # wraps into the next epoch
        ok_epoch = st.epoch == e0 + 1
        # ---- two-bucket all-reduce + Adam(grad / world) == single process on the mean ----
        nparam, split = 1000, 300
        gen = torch.Generator().manual_seed(100 + rank)
        grads = torch.randn(nparam, generator=gen)
        local = grads.clone()
        second = allreduce_buckets(grads, split)
        tail_done = grads[split:].clone()
        head_before = torch.equal(grads[:split], local[:split])     # not yet reduced
        second()
        all_local = [torch.zeros(nparam) for _ in range(world)]
        dist.all_gather(all_local, local)
        total = sum(all_local)
        ok_sum = torch.allclose(grads, total, atol=1e-6) and torch.allclose(tail_done, total[split:], atol=1e-6)
        w0 = torch.linspace(-1, 1, nparam)
        pa = {"w": w0.clone()}
        orc.AdamState(pa).step(pa, {"w": grads / world})
        pb = {"w": w0.clone()}
        orc.AdamState(pb).step(pb, {"w": total / world})
        ok_adam = torch.equal(pa["w"], pb["w"])
        q.put((rank, ok_disjoint, ok_share, ok_epoch, head_before, ok_sum, ok_adam))
    finally:
        dist.destroy_process_group()


def test_sharding_and_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def test_index_stream_single_rank():
    sys.path.insert(0, PKG)
    from pca_hip.trainer import ShardedIndexStream
    st = ShardedIndexStream(10, 4, shuffle=False)
    assert st.next().tolist() == [0, 1, 2, 3] and st.next().tolist() == [4, 5, 6, 7]
    assert st.next().tolist() == [0, 1, 2, 3]                 # tail of 2 dropped, next epoch
    with pytest.raises(ValueError):
        ShardedIndexStream(10, 8, rank=0, world=2)
