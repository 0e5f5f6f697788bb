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
