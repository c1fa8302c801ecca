"""Data-parallel gradient exchange, world_size 2, gloo on CPU (the collective logic only: no kernel runs here).

The product averages each model's flat gradient buffer across ranks before the optimizer step
(finetune/ppo.py::_DataParallel) and the 10 logged metrics with one packed all-reduce; replicas are made identical
by a rank-0 broadcast in main().  RCCL cannot be exercised in this container (no GPU); the same code path runs with
backend "nccl" on the GPU box.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeHead:
    """Stands in for an Actor/Critic: only the flat gradient buffer matters to the exchange."""

    def __init__(self, flat, split=0):
        self._flat_grad = flat
        self._bucket_split = split


class _FakeWs:
    def planes(self, name, rows, cols):
        from lr2ppo_amd.ops import Planes
        p = Planes.__new__(Planes)
        p.buf, p.rows, p.cols, p.lo_off = torch.zeros(2 * rows * cols, dtype=torch.int16), rows, cols, rows * cols
        return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lr2ppo_amd.finetune import misc, ppo
    assert misc.get_world_size() == world and misc.get_rank() == rank and misc.is_main_process() == (rank == 0)
    dp = ppo._DataParallel()
    assert dp.world == world
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(1000, generator=g)
    mine = flat.clone()
    head = _FakeHead(flat, split=300)
    dp.reduce(head)
    gathered = [torch.zeros(1000) for _ in range(world)]
    dist.all_gather(gathered, mine)
    want = torch.stack(gathered).mean(0)
    # the tail is averaged, the out_layer.fc1.weight block in front is left alone (it travels as factors)
    ok_avg = torch.allclose(head._flat_grad[300:], want[300:], atol=1e-6) and torch.equal(head._flat_grad[:300], mine[:300])
    # factor all-gather: planes [rows, cols] of every rank concatenated along rows, hi and lo planes separately
    from lr2ppo_amd.ops import Planes
    rows, cols = 3, 8
    pl = Planes.__new__(Planes)
    pl.rows, pl.cols, pl.lo_off = rows, cols, rows * cols
    pl.buf = (torch.arange(2 * rows * cols, dtype=torch.int16) + 1000 * rank)
    out = dp.gather_planes_finish(dp.gather_planes_start(pl, _FakeWs(), "x"))
    exp_hi = torch.cat([torch.arange(rows * cols, dtype=torch.int16) + 1000 * r for r in range(world)])
    exp_lo = torch.cat([torch.arange(rows * cols, 2 * rows * cols, dtype=torch.int16) + 1000 * r for r in range(world)])
    ok_avg = ok_avg and out.rows == world * rows and torch.equal(out.buf[:world * rows * cols], exp_hi) \
        and torch.equal(out.buf[out.lo_off:out.lo_off + world * rows * cols], exp_lo)
    # the identity the scheme rests on: sum_r dZ_r^T X_r == concat(dZ)^T concat(X)
    dz, xx = torch.randn(4, 5, generator=g), torch.randn(4, 7, generator=g)
    dzs, xxs = [torch.zeros(4, 5) for _ in range(world)], [torch.zeros(4, 7) for _ in range(world)]
    dist.all_gather(dzs, dz)
    dist.all_gather(xxs, xx)
    local = dz.t() @ xx
    dist.all_reduce(local)
    ok_avg = ok_avg and torch.allclose(local, torch.cat(dzs).t() @ torch.cat(xxs), atol=1e-5)
    # replicas initialised per rank (the reference's quirk 17) become identical after the rank-0 broadcast
    torch.manual_seed(7 + rank)
    p = torch.randn(64)
    dist.broadcast(p, src=0)
    ref = [torch.zeros(64) for _ in range(world)]
    dist.all_gather(ref, p)
    ok_bcast = all(torch.equal(ref[0], r) for r in ref)
    # packed metric all-reduce == mean over ranks
    m = torch.arange(10, dtype=torch.float32) + rank
    m.div_(world)
    dist.all_reduce(m)
    ok_metrics = torch.allclose(m, torch.arange(10, dtype=torch.float32) + (world - 1) / 2)
    q.put((rank, ok_avg, ok_bcast, ok_metrics))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_average_broadcast_and_metric_reduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] and r[3] for r in res), res


def test_sampler_shards_are_disjoint_and_cover():
    from torch.utils.data.distributed import DistributedSampler
    from lr2ppo_amd.finetune.ppo import SyntheticMovieNet
    ds = SyntheticMovieNet(10, 2)
    seen = []
    for r in range(2):
        s = DistributedSampler(ds, num_replicas=2, rank=r, shuffle=True)
        s.set_epoch(3)
        seen.append(list(iter(s)))
    assert len(seen[0]) == len(seen[1]) == 5 and sorted(seen[0] + seen[1]) == list(range(10))
