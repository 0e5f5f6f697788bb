"""Per-mouse data parallelism: one process per GPU, mice sharded over ranks, the shared
core replicated. One training step = sum over mice of independent passes through the shared core and a
private readout/shifter (train.py:97-111), so the only exchange is ONE all-reduce(SUM — the reference
accumulates, never averages) of the core's flat gradient arena per optimizer step (9.86 MB fp32 for the
default V1T) over RCCL/xGMI. Readout / shifter parameters live only on their owner ranks and are never
communicated, except for a mouse that two ranks share (the work is dealt in half mouse-batches so that the ranks
are balanced, and config C3 "one mouse per GPU + replica"): they split that mouse's batch and all-reduce that
mouse's arena inside a 2-rank group.
"""
from __future__ import annotations

import os
import typing as t

import torch
import torch.distributed as dist


class MouseSharding:
    PIECE_COST = 0.5  # images (~0.1 ms of a rank's 0.84 ms + 0.179 ms per image, round-6 fit); see cost() below

    def __init__(self, mouse_ids: t.Sequence[str], rank: int = 0, world: int = 1, batch_size: int = 16, make_groups: bool = True):
        self.mouse_ids = list(mouse_ids)
        self.rank, self.world = rank, world
        n = len(self.mouse_ids)
        # owners[m] = ranks that process mouse m
        self.owners: t.Dict[str, t.List[int]] = {m: [] for m in self.mouse_ids}
        # slices[m][rank] = the part of mouse m's batch that rank runs (None = all of it)
        self.slices: t.Dict[str, t.Dict[int, t.Optional[slice]]] = {m: {} for m in self.mouse_ids}
        plan = None
        if world <= 2 * n:
            # balanced dealing: every mouse-batch is cut into g equal parts (g = 1, 2, 4 or 8: the smallest that gives the
            # smallest maximum load), and the n*g parts are dealt to the ranks in order, as evenly as possible
            # (7 mice x 16 images: 56 + 56 on 2 ranks instead of 4 + 3 mice; 28 images each on 4 ranks instead of
            # 32/32/32/16; 14 images each on 8 ranks). A mouse dealt to several ranks is shared: each runs its part of the
            # batch and the mouse's arena is all-reduced inside the group of its owners.
            def deal(g):
                units = n * g
                quota = [units // world + (1 if r < units % world else 0) for r in range(world)]
                plan_, r_, left = {m: {} for m in self.mouse_ids}, 0, quota[0]
                for m in self.mouse_ids:
                    for u in range(g):
                        while left == 0:
                            r_ += 1
                            left = quota[r_]
                        lo, _ = plan_[m].get(r_, (u, u))
                        plan_[m][r_] = (lo, u + 1)
                        left -= 1
                return plan_

            def cost(plan_, g):
                # step time of the most loaded rank in image-equivalents. A rank runs the shared core ONCE over all its pieces (Trainer /
                # Model.forward_mice), so its time is a line in its images whatever the cut - round 6 (tools/sim_scaling.py,
                # profiles/r06_sim_scaling.txt): 14 images 3.35 ms, 28: 5.9, 56: 11.1, 112: 20.9 = 0.84 ms + 0.179 ms per image (round 1, when this
                # model was written: 1.2 + 0.245) - and a piece adds its slice of the one-launch-per-stage tails and, for a cut mouse, one small
                # all-reduce: ~0.1 ms = PIECE_COST of about half an image (round 1: 0.25 ms = one image, a chain of ~11 launches per piece).
                per = batch_size // g
                load = [0.0] * world
                for m in self.mouse_ids:
                    for r_, (lo, hi) in plan_[m].items():
                        load[r_] += self.PIECE_COST + (hi - lo) * per
                return max(load), sum(1 for x in load if x == 0.0)

            best = None
            for g in (1, 2, 4, 8):
                if batch_size % g:
                    continue
                plan_ = deal(g)
                key = (*cost(plan_, g), g)  # cheapest most-loaded rank, then no idle rank, then fewest cuts
                if best is None or key < best[0]:
                    best = (key, g, plan_)
            _, g, plan = best
            per = batch_size // g
            for m in self.mouse_ids:
                for r, (lo, hi) in plan[m].items():
                    self.owners[m].append(r)
                    self.slices[m][r] = None if (lo, hi) == (0, g) else slice(lo * per, hi * per)
        else:
            for r in range(world):
                self.owners[self.mouse_ids[r % n]].append(r)
            for m in self.mouse_ids:
                own = self.owners[m]
                k = len(own)
                per = (batch_size + k - 1) // k
                for i, r in enumerate(own):
                    self.slices[m][r] = None if k == 1 else slice(i * per, min(batch_size, (i + 1) * per))
        self.groups: t.Dict[str, t.Any] = {}
        if make_groups and world > 1 and dist.is_initialized():
            for m in self.mouse_ids:  # every rank must create every group, in the same order
                if len(self.owners[m]) > 1:
                    self.groups[m] = dist.new_group(ranks=self.owners[m])
        self.batch_size = batch_size
        self._comm_stream = None

    def local_mice(self) -> t.List[str]:
        return [m for m in self.mouse_ids if self.rank in self.owners[m]]

    def shared_mice(self) -> t.List[str]:
        return [m for m in self.local_mice() if len(self.owners[m]) > 1]

    def local_units(self) -> t.List[t.Tuple[str, t.Optional[slice]]]:
        """(mouse, batch slice) pairs this rank runs; slice None = the whole batch."""
        return [(m, self.slices[m][self.rank]) for m in self.local_mice()]

    def images_per_step(self) -> int:
        return self.batch_size * len(self.mouse_ids)

    def reduce_core(self, arena) -> None:
        """One blocking all-reduce(SUM) over the whole core gradient arena (the reference accumulates, never averages:
        train.py:97-111)."""
        if self.world > 1:
            dist.all_reduce(arena.grad[: arena.param_floats], op=dist.ReduceOp.SUM)

    def reduce_mouse(self, mouse_id: str, arena) -> None:
        g = self.groups.get(mouse_id)
        if g is not None:
            dist.all_reduce(arena.grad, op=dist.ReduceOp.SUM, group=g)

    # ---- overlapped exchange -------------------------------------------------------------------------------------------
    def reduce_core_overlapped(self, core, bucket_mb: float = 0.0) -> t.List[t.Any]:
        """Bucketed, asynchronous all-reduce(SUM) of the core gradient arena, one bucket per transformer block (2.2 MB fp32
        for the default V1T) launched in the order the backward completes them (`ViTCore.grad_buckets`): on the GPU each
        collective is enqueued on a communication stream behind the event the HIP backward recorded for its block
        (`v1t_vit_backward_events`), so block k's exchange runs over xGMI while blocks k-1 .. 0 are still in their backward;
        the patch-embedding / BehaviorMLP ranges follow when the backward is complete. Returns the work handles:
        `wait_all()` them before the optimizer step. Same sums as `reduce_core` (up to the order in which a ring all-reduce adds
        the ranks, which depends on an element's chunk: last bits)."""
        works: t.List[t.Any] = []
        if self.world <= 1:
            return works
        arena = core._arena
        grad = arena.grad
        buckets = core.grad_buckets()
        if not grad.is_cuda:
            for _, o, n in buckets:
                works.append(dist.all_reduce(grad[o:o + n], op=dist.ReduceOp.SUM, async_op=True))
            return works
        cur = torch.cuda.current_stream()
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        comm = self._comm_stream
        done = torch.cuda.Event()
        done.record(cur)  # everything the backward launched (host side: all of it is already enqueued)
        evs = core._block_events
        with torch.cuda.stream(comm):
            for k, o, n in buckets:
                comm.wait_event(evs[k] if (k >= 0 and evs is not None) else done)
                works.append(dist.all_reduce(grad[o:o + n], op=dist.ReduceOp.SUM, async_op=True))
        return works

    def reduce_mice_overlapped(self, arenas: t.Dict[str, t.Any]) -> t.List[t.Any]:
        """The group reductions of this rank's cut mice, issued together (different groups: they do not serialise) instead
        of one blocking collective after the other. Collectives are still issued in global mouse order on every rank."""
        works = []
        for m in self.shared_mice():
            g = self.groups.get(m)
            if g is not None:
                works.append(dist.all_reduce(arenas[m].grad, op=dist.ReduceOp.SUM, group=g, async_op=True))
        return works

    @staticmethod
    def wait_all(works: t.Sequence[t.Any]) -> None:
        for w in works:
            w.wait()  # RCCL: makes the current stream wait for the collective; gloo: blocks the host

    def attach_block_events(self, core) -> None:
        """Create the per-block events the HIP backward records (GPU only; created by a first record)."""
        if self.world > 1 and core._arena.grad is not None and core._arena.grad.is_cuda and core._block_events is None:
            evs = [torch.cuda.Event() for _ in range(core.num_blocks)]
            for e in evs:
                e.record()
            core._block_events = evs

    # ---- making every rank whole again (checkpoint / validation / evaluation) -----------------------------------------
    def gather_mice(self, model, optimizer=None) -> None:
        """Per-mouse arenas (readout + shifters, and their AdamW moments / step) live only on their owner ranks while training.
        Before anything looks at the full model - a checkpoint, validation, evaluation - broadcast each mouse's arena from
        its first owner to every rank. Collective: every rank calls it."""
        if self.world <= 1:
            return
        for m in self.mouse_ids:
            src = self.owners[m][0]
            a = model.mouse_arena(m)
            dist.broadcast(a.data, src=src)
            if optimizer is not None:
                mom, var = a.moments()
                dist.broadcast(mom, src=src)
                dist.broadcast(var, src=src)
                st = torch.tensor([float(a.step)], dtype=torch.float64, device=a.data.device)
                dist.broadcast(st, src=src)
                a.step = int(st.item())


def describe_rank(sharding: "MouseSharding", device: t.Any) -> str:
    """`rank r / world -> device -> mice / slices`: one line a rank logs to stderr BEFORE its first collective, so that a hung or
    mis-placed rank of a run nobody can rehearse is identifiable from the log alone."""
    units = ", ".join(m if sl is None else f"{m}[{sl.start}:{sl.stop}]" for m, sl in sharding.local_units())
    shared = ", ".join(f"{m}<->{sharding.owners[m]}" for m in sharding.shared_mice())
    return f"rank {sharding.rank}/{sharding.world} -> {device} -> {units or 'no units'}" + (f"; shared mice {shared}" if shared else "")


def init_from_env(backend: t.Optional[str] = None, timeout_s: t.Optional[float] = None) -> t.Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's environment; initialises the process group when world > 1.

    `timeout_s` bounds the rendezvous and every collective. Default: V1T_DIST_TIMEOUT_S when set, else torch's own (10 min for RCCL, 30 min
    for gloo) - a training run may legitimately let ranks drift apart for minutes (rank-0-only validation or checkpointing, a first-step
    build). `bench.py` passes 300 s: one hung rank must not burn a benchmark budget before anything fails."""
    import datetime

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:  # V1T_DIST_BACKEND (dev): e.g. gloo to run several ranks on one GPU
            backend = os.environ.get("V1T_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        if timeout_s is None and os.environ.get("V1T_DIST_TIMEOUT_S"):
            timeout_s = float(os.environ["V1T_DIST_TIMEOUT_S"])
        kw = {} if timeout_s is None else {"timeout": datetime.timedelta(seconds=timeout_s)}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world
