"""Per-mouse data parallelism: one process per GPU, mice sharded over ranks, the shared
core replicated. One training step = sum over mice of independent passes through the shared core and a
private readout/shifter (train.py:97-111), so the only exchange is ONE all-reduce(SUM — the reference
accumulates, never averages) of the core's flat gradient arena per optimizer step (9.86 MB fp32 for the
default V1T) over RCCL/xGMI. Readout / shifter parameters live only on their owner ranks and are never
communicated, except when world > n_mice: the surplus ranks replicate mice (config C3 "one mouse per
GPU + replica"), split that mouse's batch, and all-reduce that mouse's arena inside a 2-rank group.
"""
from __future__ import annotations

import os
import typing as t

import torch
import torch.distributed as dist


class MouseSharding:
    def __init__(self, mouse_ids: t.Sequence[str], rank: int = 0, world: int = 1, batch_size: int = 16, make_groups: bool = True):
        self.mouse_ids = list(mouse_ids)
        self.rank, self.world = rank, world
        n = len(self.mouse_ids)
        # owners[m] = ranks that process mouse m
        self.owners: t.Dict[str, t.List[int]] = {m: [] for m in self.mouse_ids}
        if world <= n:
            for i, m in enumerate(self.mouse_ids):
                self.owners[m].append(i % world)
        else:
            for r in range(world):
                self.owners[self.mouse_ids[r % n]].append(r)
        self.groups: t.Dict[str, t.Any] = {}
        if make_groups and world > 1 and dist.is_initialized():
            for m in self.mouse_ids:  # every rank must create every group, in the same order
                if len(self.owners[m]) > 1:
                    self.groups[m] = dist.new_group(ranks=self.owners[m])
        self.batch_size = batch_size

    def local_mice(self) -> t.List[str]:
        return [m for m in self.mouse_ids if self.rank in self.owners[m]]

    def shared_mice(self) -> t.List[str]:
        return [m for m in self.local_mice() if len(self.owners[m]) > 1]

    def local_units(self) -> t.List[t.Tuple[str, t.Optional[slice]]]:
        """(mouse, batch slice) pairs this rank runs; slice None = the whole batch."""
        out = []
        for m in self.local_mice():
            own = self.owners[m]
            if len(own) == 1:
                out.append((m, None))
            else:
                k, i = len(own), own.index(self.rank)
                per = (self.batch_size + k - 1) // k
                out.append((m, slice(i * per, min(self.batch_size, (i + 1) * per))))
        return out

    def images_per_step(self) -> int:
        return self.batch_size * len(self.mouse_ids)

    def reduce_core(self, arena) -> None:
        if self.world > 1:
            dist.all_reduce(arena.grad[: arena.param_floats], op=dist.ReduceOp.SUM)

    def reduce_mouse(self, mouse_id: str, arena) -> None:
        g = self.groups.get(mouse_id)
        if g is not None:
            dist.all_reduce(arena.grad, op=dist.ReduceOp.SUM, group=g)


def init_from_env(backend: t.Optional[str] = None) -> t.Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's environment; initialises the process group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
