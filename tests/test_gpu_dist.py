"""Two ranks sharing the one GPU of the test box (gloo, which reduces device tensors through the host): the REAL trainer
path of a multi-rank step — MouseSharding deals 3 mice as 1.5 + 1.5 (mouse B cut 2 + 2), every rank runs its pieces through
the shared core in one pass (Model.forward_mice), the core gradient arena is all-reduced (SUM, train.py:97-111), the cut
mouse's arena is reduced inside its 2-rank group — against the single-rank step on the same data. (RCCL itself is only
exercised by the driver's multi-GPU bench; this checks everything around the collective on real kernels.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
MICE = ("A", "B", "C")
NEURONS = {"A": 96, "B": 50, "C": 130}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _step(rank, world, port, q, overlap=True):
    """One train_step with the optimizer replaced by a recorder of the (reduced) gradient arenas."""
    import torch.distributed as dist

    from oracle import v1t_oracle as O
    from oracle import weights as W
    from tests.helpers import build_native_model
    from v1t_amd.dist import MouseSharding
    from v1t_amd.synthetic import make_ds
    from v1t_amd.trainer import Trainer

    os.environ["V1T_DIST_OVERLAP"] = "1" if overlap else "0"  # read by Trainer.__init__
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    cfg = O.Config(num_blocks=2, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=MICE, num_neurons=NEURONS, p_dropout=0.0, t_dropout=0.0)
    sd = W.make_state_dict(cfg, 17)
    model, args = build_native_model(cfg, sd, dev)
    args.batch_size = 4
    sh = MouseSharding(list(MICE), rank=rank, world=world, batch_size=4)
    tr = Trainer(args, model, make_ds(NEURONS), sh)
    batches = {m: {k: v.to(dev) for k, v in W.make_batch(cfg, m, 4, 17).items()} for m in MICE}
    gen = torch.Generator().manual_seed(5)
    eps = {m: torch.randn(4, NEURONS[m], 2, generator=gen).to(dev) for m in MICE}
    for m, sl in sh.local_units():  # the readout's sampling noise of exactly the images this rank runs
        ro, e = model.readouts[m], (eps[m] if sl is None else eps[m][sl])
        ro.forward = (lambda inputs, sample=None, shifts=None, eps=None, _o=ro.forward, _e=e: _o(inputs, sample=sample, shifts=shifts, eps=_e))
    rec = {}
    names = {id(model.core._arena): "core", **{id(model.mouse_arena(m)): m for m in MICE}}
    tr.opt.step_arena = lambda arena, ranges, zero_grad=True: rec.__setitem__(names[id(arena)], arena.grad.detach().cpu().numpy().copy())  # numpy: pickled by value through the queue
    out = tr.train_step(batches)
    torch.cuda.synchronize()
    q.put((rank, [(m, None if sl is None else (sl.start, sl.stop)) for m, sl in sh.local_units()], rec, float(out["loss"])))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run(world, overlap=True):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_step, args=(r, world, port, q, overlap)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=300) for _ in range(world)]
    [p.join(timeout=120) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    return sorted(res, key=lambda r: r[0])


def test_two_rank_step_equals_single_rank_step():
    assert torch.cuda.is_available()
    (_, units1, ref, loss1), = _run(1)
    assert units1 == [("A", None), ("B", None), ("C", None)]
    two = _run(2)
    assert two[0][1] == [("A", None), ("B", (0, 2))] and two[1][1] == [("B", (2, 4)), ("C", None)]
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    for rank, _, rec, _ in two:
        assert rel(rec["core"], ref["core"]) < 2e-3, rank              # all-reduced core gradient = the single-rank sum over mice
        assert rel(rec["B"], ref["B"]) < 2e-3, rank                    # the cut mouse: reduced inside its 2-rank group
    assert rel(two[0][2]["A"], ref["A"]) < 2e-3 and rel(two[1][2]["C"], ref["C"]) < 2e-3  # whole mice stay local
    assert "C" not in two[0][2] and "A" not in two[1][2]
    assert abs(two[0][3] + two[1][3] - loss1) <= 1e-4 * abs(loss1)     # the local losses add up to the global one
    # the default above is the overlapped exchange (per-block buckets behind the backward's events on a communication stream,
    # concurrent group reductions); the blocking single-shot exchange gives the same reduced gradients
    blocking = _run(2, overlap=False)
    for a_, b_ in zip(two, blocking):
        assert a_[1] == b_[1]
        for k in a_[2]:
            assert rel(a_[2][k], b_[2][k]) < 2e-3, k  # float atomics inside the backward: two runs differ in the last bits


def _rccl_single_rank(port, q):
    """One rank, backend nccl (= RCCL): the mechanics of the overlapped exchange on the real library - per-block events recorded
    by v1t_vit_backward_events, the communication stream waiting on them, asynchronous all-reduces over arena slices, the
    stream-side wait - with a world of one, where an all-reduce(SUM) must return its input."""
    import torch.distributed as dist

    from oracle import v1t_oracle as O
    from oracle import weights as W
    from tests.helpers import build_native_model
    from v1t_amd.dist import MouseSharding
    from v1t_amd.losses import elu1_poisson_loss

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda:0")
    cfg = O.Config(num_blocks=3, emb_dim=64, mlp_dim=128, num_heads=4, mouse_ids=("A",), num_neurons={"A": 96}, p_dropout=0.0, t_dropout=0.0)
    model, _ = build_native_model(cfg, W.make_state_dict(cfg, 17), dev)
    model.train(False)
    core = model.core
    core.prepare()
    core._arena.attach_grads()
    core._arena.grad.zero_()
    sh = MouseSharding(["A"], rank=0, world=1, batch_size=4)
    sh.world = 2  # take the multi-rank code path; the process group itself has one rank
    sh.attach_block_events(core)
    assert core._block_events is not None and len(core._block_events) == 3
    b = {k: v.to(dev) for k, v in W.make_batch(cfg, "A", 4, 17).items()}
    u = model(inputs=b["image"], mouse_id="A", behaviors=b["behavior"], pupil_centers=b["pupil_center"], activate=False)[0]
    loss, _ = elu1_poisson_loss(u, b["response"], 4500.0, 4)
    loss.backward()
    works = sh.reduce_core_overlapped(core)
    sh.wait_all(works)
    torch.cuda.synchronize()
    after = core._arena.grad.clone()
    # reference: the same backward without events / exchange
    core._block_events = None
    core._arena.grad.zero_()
    u = model(inputs=b["image"], mouse_id="A", behaviors=b["behavior"], pupil_centers=b["pupil_center"], activate=False)[0]
    loss, _ = elu1_poisson_loss(u, b["response"], 4500.0, 4)
    loss.backward()
    torch.cuda.synchronize()
    ref = core._arena.grad.clone()
    err = float((after - ref).abs().max() / (ref.abs().max() + 1e-30))
    q.put((len(works), err, bool(all(e.query() for e in sh.__dict__.get("_unused", [])))))
    dist.destroy_process_group()


def test_overlapped_exchange_mechanics_on_rccl_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank, args=(_free_port(), q))
    p.start()
    n_works, err, _ = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert n_works >= 4  # 3 block buckets + the late ranges
    assert err < 2e-3  # float atomics inside the backward: two runs differ in the last bits
