"""CPU, 2 processes over gloo: the per-mouse data-parallel exchange (v1t_amd/dist.py). Each rank fills
the core gradient arena with the contribution of ITS mice; after reduce_core every rank must hold the
single-process sum over all mice (SUM, not mean — reference train.py:97-111), and per-mouse arenas stay
local. With world > n_mice the replicated mouse's arena is reduced inside its 2-rank group."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from v1t_amd.dist import MouseSharding
from v1t_amd.flat import FlatArena


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _mouse_grad(mouse_index: int, n: int, sl=None) -> torch.Tensor:
    g = torch.Generator().manual_seed(100 + mouse_index)
    per_image = torch.randn(16, n, generator=g)  # contribution of each image of the mouse's batch
    return per_image[sl if sl is not None else slice(None)].sum(0)


def _worker(rank, world, port, mice, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000
    core = FlatArena.from_params([torch.nn.Parameter(torch.zeros(n))])
    core.flatten()
    sh = MouseSharding(mice, rank=rank, world=world, batch_size=16)
    arenas = {}
    for m, sl in sh.local_units():
        core.grad += _mouse_grad(mice.index(m), n, sl)
        a = FlatArena.from_params([torch.nn.Parameter(torch.zeros(50))])
        a.flatten()
        a.grad += _mouse_grad(50 + mice.index(m), 50, sl)
        arenas[m] = a
    sh.reduce_core(core)
    for m in sh.shared_mice():
        sh.reduce_mouse(m, arenas[m])
    q.put((rank, core.grad.clone(), {m: a.grad.clone() for m, a in arenas.items()}))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, mice):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, mice, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    expect = sum(_mouse_grad(i, 1000) for i in range(len(mice)))
    for rank, core, arenas in res:
        assert torch.allclose(core, expect, atol=1e-4), rank
        for m, g in arenas.items():
            assert torch.allclose(g, _mouse_grad(50 + mice.index(m), 50), atol=1e-4), (rank, m)


def test_two_ranks_seven_mice():
    _run(2, list("ABCDEFG"))


def test_three_ranks_one_mouse_cut():
    # 7 mice on 3 ranks: 38 / 38 / 36 images, mice C and E cut across neighbouring ranks (their arenas reduced in 2-rank groups)
    loads = []
    for r in range(3):
        sh = MouseSharding(list("ABCDEFG"), rank=r, world=3, batch_size=16, make_groups=False)
        loads.append(sum(16 if sl is None else sl.stop - sl.start for _, sl in sh.local_units()))
    assert loads == [38, 38, 36]
    _run(3, list("ABCDEFG"))


def test_four_and_eight_ranks_balanced_images():
    # a rank runs the shared core once over all its pieces, so what counts is its image total: 28 each on 4 ranks
    # (quarters of a mouse-batch), 14 each on 8 (eighths); every mouse keeps its 16 images
    for world, want in ((4, 28), (8, 14)):
        loads, seen = [], {}
        for r in range(world):
            sh = MouseSharding(list("ABCDEFG"), rank=r, world=world, batch_size=16, make_groups=False)
            loads.append(sum(16 if sl is None else sl.stop - sl.start for _, sl in sh.local_units()))
            for m, sl in sh.local_units():
                seen.setdefault(m, []).extend(range(16)[sl] if sl is not None else range(16))
        assert loads == [want] * world
        assert all(sorted(v) == list(range(16)) for v in seen.values()) and len(seen) == 7


def test_four_ranks_three_cut_mice_collectives():
    _run(4, list("ABCDEFG"))  # B, D and F are shared by neighbouring ranks: three 2-rank groups + the core all-reduce


def test_two_ranks_balanced_halves():
    loads = []
    for r in range(2):
        sh = MouseSharding(list("ABCDEFG"), rank=r, world=2, batch_size=16, make_groups=False)
        loads.append(sum(16 if sl is None else sl.stop - sl.start for _, sl in sh.local_units()))
    assert loads == [56, 56]


def test_replica_rank_shares_a_mouse():
    _run(2, ["A"])  # world > n_mice: both ranks own mouse A and split its batch 8 + 8


def test_eight_ranks_every_mouse_cut_collectives():
    _run(8, list("ABCDEFG"))  # 14 images per rank: all seven mice shared by neighbouring ranks (seven 2-rank groups)


# ---- overlapped (bucketed, asynchronous) exchange and gather_mice on the real core arena ---------------------------------
def _core_worker(rank, world, port, q):
    import numpy as np

    import v1t_amd
    from v1t_amd.synthetic import default_args, make_ds

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mice = list("ABCDEFG")
    neurons = {m: 20 + i for i, m in enumerate(mice)}
    args = default_args(input_shape=(1, 36, 64), resize_image=0, num_blocks=3, emb_dim=32, mlp_dim=48, num_heads=2)
    args.output_shapes = {m: (n,) for m, n in neurons.items()}
    torch.manual_seed(0)
    model = v1t_amd.Model(args, make_ds(neurons))  # CPU: constructors + flat arenas only, no kernels
    core = model.core
    core._arena.ensure()
    core._arena.attach_grads()
    sh = MouseSharding(mice, rank=rank, world=world, batch_size=16)
    buckets = core.grad_buckets()
    assert [k for k, _, _ in buckets[:3]] == [2, 1, 0] and all(k == -1 for k, _, _ in buckets[3:])  # blocks complete last to first
    n = core._arena.param_floats
    g = torch.Generator().manual_seed(1000 + rank)
    local = torch.randn(n, generator=g)
    core._arena.grad[:n] = local
    sh.wait_all(sh.reduce_core_overlapped(core))
    bucketed = core._arena.grad[:n].clone()
    core._arena.grad[:n] = local
    sh.reduce_core(core._arena)
    single = core._arena.grad[:n].clone()
    # cut mice: concurrent group reductions == sequential ones
    arenas = {m: model.mouse_arena(m) for m in sh.local_mice()}
    for m, a in arenas.items():
        a.attach_grads()
        a.grad[:] = float(rank + 1) * (1 + mice.index(m))
    sh.wait_all(sh.reduce_mice_overlapped(arenas))
    shared = {m: float(arenas[m].grad[0]) for m in sh.shared_mice()}
    # gather_mice: every rank ends up with the owner's values and step count
    for m in mice:
        a = model.mouse_arena(m)
        with torch.no_grad():
            a.data.fill_(100.0 * (sh.owners[m][0] + 1) if rank == sh.owners[m][0] else -1.0)
        mom, var = a.moments()
        mom.fill_(float(rank))
        a.step = 7 if rank == sh.owners[m][0] else 0
    opt = object()
    sh.gather_mice(model, opt)
    gathered = {m: (float(model.mouse_arena(m).data[0]), float(model.mouse_arena(m).exp_avg[0]), model.mouse_arena(m).step) for m in mice}
    q.put((rank, bool(torch.allclose(bucketed, single, rtol=1e-5, atol=1e-5)), float(single.abs().sum()), shared, {m: list(sh.owners[m]) for m in mice}, gathered))
    dist.barrier()
    dist.destroy_process_group()


def _run_core(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_core_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=180) for _ in range(world)]
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    sums = set()
    for rank, same, total, shared, owners, gathered in res:
        assert same, rank  # same sums; a ring all-reduce adds the ranks in an order that depends on the element's chunk: last bits only
        sums.add(round(total, 0))
        for m, v in shared.items():
            assert v == sum(float(r + 1) for r in owners[m]) * (1 + "ABCDEFG".index(m)), (rank, m, v)
        for m, (val, mom, step) in gathered.items():
            src = owners[m][0]
            assert val == 100.0 * (src + 1) and mom == float(src) and step == 7, (rank, m, val, mom, step)
    assert len(sums) == 1  # every rank holds the same reduced core gradient


def test_bucketed_reduce_and_gather_world2():
    _run_core(2)


def test_bucketed_reduce_and_gather_world4():
    _run_core(4)


def test_bucketed_reduce_and_gather_world8():
    _run_core(8)
