"""Training step for the native path: equivalent of the reference's train()/train_step()
(train.py:42-116) — one optimizer step = one mouse-batch per mouse, gradients SUMMED over mice
(train.py:97-111), Poisson loss scaled by sqrt(ds_size / batch) (losses.py:114-119), L1 regulariser on
all core parameters once per mouse-batch and on the visited readout's features (train.py:71,
model.py:141-149), AdamW with wd = 0 (train.py:216-223).

MI355X-first differences (same math):
  * the HIP backward accumulates directly into flat gradient arenas;
  * the L1 term is grad-independent, so its gradient n_mice * lambda * sign(p) is folded into the fused
    L1+AdamW kernel instead of a read-modify-write of every gradient per micro-batch;
  * no per-step host sync: losses stay on the device (the reference's gather() does .cpu() each step);
  * multi-GPU: mice are sharded over ranks, the shared core's gradient arena is ONE RCCL all-reduce(SUM).
"""
from __future__ import annotations

import contextlib
import os
import typing as t

import torch

from . import lib as L
from .dist import MouseSharding
from .losses import elu1_poisson_loss
from .model import Model


class FusedAdamW:
    """AdamW over flat arenas via v1t_adamw_step (one launch per (arena, range)).

    `param_groups`, `state_dict()` and `load_state_dict()` speak torch.optim.AdamW's format over
    `model.get_parameters(core_lr)` (train.py:216-223), so checkpoints written by the reference's Scheduler
    (utils/scheduler.py:84-144) load here and the ones written here load into the reference's optimizer."""

    def __init__(self, lr: float, betas=(0.9, 0.9999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.model: t.Optional[Model] = None
        self.param_groups: t.List[t.Dict[str, t.Any]] = []

    def bind(self, model: Model, core_lr: float) -> None:
        self.model = model
        self.param_groups = []
        for g in model.get_parameters(core_lr=core_lr):
            self.param_groups.append({"name": g["name"], "lr": float(g.get("lr", self.lr)), "betas": tuple(self.betas), "eps": self.eps,
                                      "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False, "params": list(g["params"])})

    def group_lr(self, name: str) -> float:
        for g in self.param_groups:
            if g["name"] == name:
                return float(g["lr"])
        raise KeyError(name)

    def _slots(self) -> t.Dict[int, t.Tuple[t.Any, t.Any]]:
        """id(parameter) -> (arena, slot) over the core arena and every mouse arena"""
        m = self.model
        m.core._arena.ensure()  # flat storage only: no kernel launch, works on a CPU model too
        arenas = [m.core._arena] + [m.mouse_arena(k) for k in m.readouts.keys()]
        return {id(s.tensor): (a, s) for a in arenas for s in a.slots}

    def state_dict(self) -> t.Dict[str, t.Any]:
        slots = self._slots()
        state, groups, i = {}, [], 0
        for g in self.param_groups:
            idx = []
            for p in g["params"]:
                a, s = slots[id(p)]
                if a.exp_avg is not None and a.step > 0:
                    view = lambda buf: s.view(buf[s.offset:s.offset + s.numel]).detach().clone().contiguous()
                    state[i] = {"step": torch.tensor(float(a.step)), "exp_avg": view(a.exp_avg), "exp_avg_sq": view(a.exp_avg_sq)}
                idx.append(i)
                i += 1
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": idx})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd: t.Dict[str, t.Any]) -> None:
        """torch.optim.AdamW-format state. Groups are matched BY NAME when the file names them (the reference does,
        model.py:112-139), by position otherwise; every moment's shape is checked against its parameter before anything
        is copied, so a file with another group order or another model fails instead of swapping moments silently."""
        saved_groups = sd["param_groups"]
        if len(saved_groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        if all("name" in g for g in saved_groups):
            by_name = {g["name"]: g for g in saved_groups}
            if len(by_name) != len(saved_groups) or set(by_name) != {g["name"] for g in self.param_groups}:
                raise ValueError(f"parameter groups {sorted(by_name)} do not match {[g['name'] for g in self.param_groups]}")
            pairs = [(g, by_name[g["name"]]) for g in self.param_groups]
        else:
            pairs = list(zip(self.param_groups, saved_groups))
        slots = self._slots()
        todo = []
        for g, saved in pairs:
            if len(saved["params"]) != len(g["params"]):
                raise ValueError(f"group {g['name']}: loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
            for p, j in zip(g["params"], saved["params"]):
                st = sd["state"].get(j, sd["state"].get(str(j)))
                if st is None:
                    continue
                for k in ("exp_avg", "exp_avg_sq"):
                    if tuple(st[k].shape) != tuple(p.shape):
                        raise ValueError(f"group {g['name']}: {k} of saved parameter {j} has shape {tuple(st[k].shape)}, the parameter {tuple(p.shape)}")
                todo.append((slots[id(p)], st))
        steps: t.Dict[int, int] = {}
        for (a, s), st in todo:
            m_, v_ = a.moments()
            s.view(m_[s.offset:s.offset + s.numel]).copy_(st["exp_avg"].to(m_.device))
            s.view(v_[s.offset:s.offset + s.numel]).copy_(st["exp_avg_sq"].to(v_.device))
            n = int(float(st["step"]))
            if steps.setdefault(id(a), n) != n:
                raise ValueError("parameters of one arena carry different step counts")
            a.step = n
        for g, saved in pairs:
            g["lr"] = float(saved["lr"])

    def step_arena(self, arena, ranges: t.Sequence[t.Tuple[int, int, float, float]], zero_grad: bool = True) -> None:
        """ranges: (start, n, l1_coeff, lr) in floats; all ranges of one arena share its step counter."""
        arena.step += 1
        m, v = arena.moments()
        lib = L.load()
        for start, n, l1, lr in ranges:
            if n <= 0:
                continue
            o = 4 * start
            L.check(lib.v1t_adamw_step(arena.data.data_ptr() + o, arena.grad.data_ptr() + o, m.data_ptr() + o, v.data_ptr() + o, n, lr,
                                       self.betas[0], self.betas[1], self.eps, self.weight_decay, arena.step, l1, int(zero_grad), L.stream()),
                    "adamw_step")


class Trainer:
    def __init__(self, args, model: Model, ds: t.Dict[str, t.Any], sharding: t.Optional[MouseSharding] = None):
        self.args, self.model = args, model
        self.mouse_ids = list(ds.keys())
        self.ds_sizes = {m: float(len(d.dataset)) for m, d in ds.items()}
        self.sharding = sharding or MouseSharding(self.mouse_ids, rank=0, world=1)
        self.lr = args.lr
        self.core_lr = args.lr if getattr(args, "core_lr", None) is None else args.core_lr
        self.opt = FusedAdamW(args.lr, betas=(args.adam_beta1, args.adam_beta2), eps=args.adam_eps)
        self.opt.bind(model, self.core_lr)
        self.batch_size = args.batch_size
        self._core_l1: t.Optional[float] = None
        # V1T_CORE_GROUP=n (dev): mouse-batches per pass of the shared core; 1 = one core pass per mouse-batch, as the
        # reference's loop does (default: all local mouse-batches in one pass)
        self.core_group = int(os.environ.get("V1T_CORE_GROUP", "7"))
        self.batch_core = self.core_group > 1
        # overlapped exchange (per-block buckets behind the backward's events) by default over RCCL; gloo moves device tensors
        # through the host and its asynchronous form is pathologically slow there (424 vs 34 ms per step, 2 ranks on one GPU),
        # so it keeps the blocking single-shot all-reduce. V1T_DIST_OVERLAP=0 / 1 forces either form (dev, tests).
        ov = os.environ.get("V1T_DIST_OVERLAP", "")
        backend = torch.distributed.get_backend() if (self.sharding.world > 1 and torch.distributed.is_initialized()) else ""
        self.overlap = (ov == "1") or (ov != "0" and backend == "nccl")
        if self.sharding.world > 1:
            model.core.fold_rank(self.sharding.rank)

    def train_step(self, batches: t.Dict[str, t.Dict[str, torch.Tensor]]) -> t.Dict[str, torch.Tensor]:
        """batches: mouse_id -> full batch (image, behavior, pupil_center, response) on the device.
        Each rank processes its share (whole mice, or a slice of a replicated mouse's batch)."""
        model = self.model
        model.train(True)
        core = model.core
        core.prepare()
        if self.sharding.world > 1 and self.overlap:
            self.sharding.attach_block_events(core)
        losses = []
        units = []
        single_pass = True  # one core backward per step: its per-block events are the step's
        for mouse_id, sl in self.sharding.local_units():
            b = batches[mouse_id]
            full = b["image"].shape[0]
            if sl is not None:
                b = {k: v[sl] for k, v in b.items()}
            model.mouse_arena(mouse_id).attach_grads()
            units.append((mouse_id, b, full))
        if self.core_group > 1 and len(units) > 1 and core.behavior_mode != 4:
            # groups of local mouse-batches through the shared core in one pass, one backward per group (gradients sum as
            # train.py:97-111)
            single_pass = len(units) <= self.core_group
            for i in range(0, len(units), self.core_group):
                grp = units[i:i + self.core_group]
                us = model.forward_mice([(m, b) for m, b, _ in grp], activate=False)
                st = getattr(model, "_last_streams", None)
                ls = []
                for i, ((m, b, full), u) in enumerate(zip(grp, us)):
                    with (torch.cuda.stream(st[i]) if st else contextlib.nullcontext()):  # the loss stays on the mouse's stream
                        ls.append(elu1_poisson_loss(u, b["response"], self.ds_sizes[m], full)[0])
                    if st:
                        ls[-1].record_stream(torch.cuda.current_stream())  # summed on the main stream below
                model.join_streams()
                (torch.stack(ls).sum() if len(ls) > 1 else ls[0]).backward()
                model.join_streams()  # the readout backward kernels wrote into the mouse arenas on their side streams
                losses += [l_.detach() for l_ in ls]
        else:
            single_pass = len(units) <= 1
            for mouse_id, b, full in units:
                u, _, _ = model(inputs=b["image"], mouse_id=mouse_id, behaviors=b["behavior"], pupil_centers=b["pupil_center"], activate=False)
                loss, _ = elu1_poisson_loss(u, b["response"], self.ds_sizes[mouse_id], full)
                loss.backward()
                losses.append(loss.detach())
        # data-parallel exchange: the core buckets were enqueued behind the per-block events of the backward (they overlap its
        # tail), the cut mice's group reductions run concurrently; everything is awaited (stream-side) before the optimizer
        sh = self.sharding
        if sh.world > 1:
            if self.overlap and single_pass:
                works = sh.reduce_core_overlapped(core)
                works += sh.reduce_mice_overlapped({m: model.mouse_arena(m) for m in sh.shared_mice()})
                sh.wait_all(works)
            else:
                sh.reduce_core(core._arena)
                for mouse_id in sh.shared_mice():
                    sh.reduce_mouse(mouse_id, model.mouse_arena(mouse_id))
        # optimizer: core (L1 once per mouse-batch of the global step), then the local mice's arenas
        if not core.frozen:
            ca = core._arena
            if self._core_l1 is None:  # the reg_scale buffer lives on the device: one read, not one sync per step
                self._core_l1 = float(core.reg_scale) * len(self.mouse_ids)
            self.opt.step_arena(ca, [(0, ca.param_floats, self._core_l1, self.opt.group_lr("core"))])
            core.mark_updated()
        for mouse_id in self.sharding.local_mice():
            a = model.mouse_arena(mouse_id)
            # one launch per (L1 coefficient, optimizer group) run: readouts / image_cropper / core_shifter keep their own lr
            self.opt.step_arena(a, [(o, n, c, self.opt.group_lr(g)) for o, n, c, g in model.mouse_step_ranges(mouse_id)])
        return {"loss": torch.stack(losses).sum() if losses else torch.zeros((), device=core._arena.data.device)}

    @torch.no_grad()
    def predict(self, batch: t.Dict[str, torch.Tensor], mouse_id: str) -> torch.Tensor:
        self.model.train(False)
        y, _, _ = self.model(inputs=batch["image"], mouse_id=mouse_id, behaviors=batch["behavior"], pupil_centers=batch["pupil_center"])
        return y
