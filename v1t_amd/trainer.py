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
import ctypes as C
import math
import os
import typing as t
import warnings

import torch

from . import lib as L
from .dist import MouseSharding
from .losses import elu1_poisson_loss
from .model import Model, mouse_arena, mouse_step_ranges


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

    # ---- torch.optim.Optimizer surface: an OPT-IN replacement for `torch.optim.AdamW(params=model.get_parameters(core_lr), ...)` in the
    # reference's own loop (train.py:216-223; `scaler.step(optimizer)` with the GradScaler disabled calls `optimizer.step()`, train.py:76-79)
    @classmethod
    def for_model(cls, model: Model, lr: float, core_lr: t.Optional[float] = None, betas=(0.9, 0.9999), eps: float = 1e-8, weight_decay: float = 0.0) -> "FusedAdamW":
        opt = cls(lr, betas=betas, eps=eps, weight_decay=weight_decay)
        opt.bind(model, lr if core_lr is None else core_lr)
        return opt

    @staticmethod
    def _has_grads(arena) -> bool:
        first = next((s for s in arena.slots if s.is_param and s.tensor.requires_grad), None)
        return first is not None and first.tensor.grad is not None

    def _adopt_grads(self, arena) -> None:
        """Gradients that autograd accumulated OUTSIDE the gradient arena (a parameter whose .grad was None when its backward ran gets a
        fresh tensor from AccumulateGrad) are added into the arena and the .grad views re-attached; no-op when they already are views."""
        stray = [(s, s.tensor.grad) for s in arena.slots if s.is_param and s.tensor.grad is not None
                 and s.tensor.grad.data_ptr() != arena.grad.data_ptr() + 4 * s.offset]
        if stray:
            with torch.no_grad():
                for s, g in stray:
                    s.view(arena.grad[s.offset:s.offset + s.numel]).add_(g)
                    s.tensor.grad = s.view(arena.grad[s.offset:s.offset + s.numel])

    @torch.no_grad()
    def step(self, closure=None):
        """One AdamW step over the core arena and the arena of every mouse whose readout ran a forward in grad mode since the last step
        (torch.optim skips parameters whose .grad is None: the mice not visited since zero_grad(set_to_none=True); here the gradient
        views stay attached and the readout's `_visited` mark says the same). The L1 coefficients are 0: in the reference's loop the L1
        gradient arrives through autograd (`model.regularizer`, train.py:71) - unlike `Trainer`, which folds it into this kernel. The
        kernel zeroes the gradients it consumed (the `optimizer.zero_grad()` that follows, train.py:79, finds them clean)."""
        model = self.model
        core = model.core
        if not core.frozen and self._has_grads(core._arena):
            ca = core._arena
            self._adopt_grads(ca)
            self.step_arena(ca, [(0, ca.param_floats, 0.0, self.group_lr("core"))], zero_grad=True)
            core.mark_updated()
        items = []
        for m in model.readouts.keys():
            ro = model.readouts[m]
            a = mouse_arena(model, m)
            if not (getattr(ro, "_visited", False) or any(s.tensor.grad is not None and s.tensor.grad.data_ptr() != a.grad.data_ptr() + 4 * s.offset
                                                           for s in a.slots if s.is_param)):
                continue
            ro._visited = False
            a.attach_grads()
            self._adopt_grads(a)
            items.append((a, [(o, n, 0.0, self.group_lr(g)) for o, n, _, g in mouse_step_ranges(model, m)]))
        self.step_arenas(items, zero_grad=True)

    def zero_grad(self, set_to_none: bool = True) -> None:
        """torch.optim.Optimizer.zero_grad over the arenas. Every parameter's .grad stays (or becomes) the view of its gradient arena - the
        backward kernels accumulate straight into it - and every arena is zero-filled, ALSO when `step` has just zeroed it in its kernel:
        a backward that ran between `step()` and this call (an auxiliary loss, a discarded backward after a skipped GradScaler step) must
        not leak into the next step (torch.optim semantics; ADVICE r05 - eight fills of 5-10 MB, ~0.2 % of a step).
        `set_to_none` is accepted for signature compatibility; a mouse that is not visited is skipped by `step` through the readout's mark,
        which is what `None` gradients achieve in torch.optim."""
        model = self.model
        arenas = [model.core._arena] + [mouse_arena(model, m) for m in model.readouts.keys()]
        for a in arenas:
            a.ensure()
            a.grad.zero_()
            first = next((s for s in a.slots if s.is_param and s.tensor.requires_grad), None)
            if first is not None and (first.tensor.grad is None or first.tensor.grad.data_ptr() != a.grad.data_ptr() + 4 * first.offset):
                a.attach_grads(force=True)

    def group_lr(self, name: str) -> float:
        for g in self.param_groups:
            if g["name"] == name:
                return float(g["lr"])
        raise KeyError(name)

    def _slots(self) -> t.Dict[int, t.Tuple[t.Any, t.Any]]:
        """id(parameter) -> (arena, slot) over the core arena and every mouse arena"""
        m = self.model
        m.core._arena.ensure()  # flat storage only: no kernel launch, works on a CPU model too
        arenas = [m.core._arena] + [mouse_arena(m, k) for k in m.readouts.keys()]
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
        steps: t.Dict[int, int] = {}
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
                a_, s_ = slots[id(p)]
                n = int(float(st["step"]))
                if steps.setdefault(id(a_), n) != n:  # one fused launch per arena shares one bias-correction step
                    raise ValueError(f"group {g['name']}: parameters of one arena carry different step counts ({steps[id(a_)]} and {n})")
                todo.append(((a_, s_), st))
        # nothing has been written up to here: a rejected file leaves the optimizer untouched
        for (a, s), st in todo:
            m_, v_ = a.moments()
            s.view(m_[s.offset:s.offset + s.numel]).copy_(st["exp_avg"].to(m_.device))
            s.view(v_[s.offset:s.offset + s.numel]).copy_(st["exp_avg_sq"].to(v_.device))
            a.step = steps[id(a)]
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


    def step_arenas(self, items: t.Sequence[t.Tuple[t.Any, t.Sequence[t.Tuple[int, int, float, float]]]], zero_grad: bool = True) -> None:
        """`step_arena` for several arenas in ONE launch (`v1t_adamw_multi`: the seven mice of a step are 21 small ranges). An instance that
        overrides `step_arena` (tests intercept the gradients there) gets one call per arena instead."""
        if "step_arena" in self.__dict__:
            for arena, ranges in items:
                self.step_arena(arena, ranges, zero_grad)
            return
        rs = []
        for arena, ranges in items:
            arena.step += 1
            m, v = arena.moments()
            for start, n, l1, lr in ranges:
                if n > 0:
                    o = 4 * start
                    rs.append(L.AdamRange(arena.data.data_ptr() + o, arena.grad.data_ptr() + o, m.data_ptr() + o, v.data_ptr() + o, n, lr, l1, arena.step, 0))
        if rs:
            arr = (L.AdamRange * len(rs))(*rs)
            L.check(L.load().v1t_adamw_multi(arr, len(rs), self.betas[0], self.betas[1], self.eps, self.weight_decay, int(zero_grad), L.stream()), "adamw_multi")


class _NativeStep:
    """One optimizer step's forward + backward for a fixed list of (mouse, n images) units WITHOUT torch autograd or ATen
    kernels on the way: every buffer is allocated once, the C-ABI entry points are called directly in the order autograd would
    run them, gradients land in the flat arenas. Same math as `Model.forward_mice` + `elu1_poisson_loss` + `.backward()` (what the
    reference's loop over mice computes, train.py:42-116); what disappears is ~130 small launches per step (zero-fills, copies,
    cat / stack / add kernels, torch's Philox normal_()) and the autograd bookkeeping on the host - the fixed cost that limits
    the data-parallel speed-up when a rank only has 14 images.

    Falls back (returns None from `build`) when the configuration needs something only the module path does: per-mouse
    BehaviorMLPs (behavior_mode 4), a learned image shifter / centre crop / behaviour-as-channels in the cropper, stochastic depth,
    another readout type, or a readout whose `forward` was overridden on the instance (tests inject eps that way)."""

    @staticmethod
    def unsupported(trainer: "Trainer", units: t.Sequence[t.Tuple[str, t.Dict[str, torch.Tensor], int]]) -> t.Optional[str]:
        """Why this configuration cannot take the native step (None: it can)."""
        from .readout import Gaussian2DReadout

        model = trainer.model
        core, crop = model.core, model.image_cropper
        if not units:
            return "no local units"
        if core.behavior_mode == 4:
            return "behavior_mode 4 (one BehaviorMLP per mouse: the shared core cannot run all mice in one pass)"
        if core.frozen:
            return "frozen core"
        if core.drop_path_rate > 0:
            return "stochastic depth (drop_path > 0)"
        if crop.image_shifter is not None:
            return "learned image shifter (shift_mode 1 / 3 / 4)"
        if crop.crop_scale < 1:
            return "center_crop < 1"
        if crop.behavior_mode == 1:
            return "behavior_mode 1 (behaviour as image channels)"
        gh, gw = core.output_shape[1:]
        if gh * gw > 4096:
            # the split (sort / dz / parameter) readout backward needs the sorted form: LDS histogram of <= 4096 cells (readout.hip)
            return f"latent grid {gh} x {gw} > 4096 cells"
        for m, b_, _ in units:
            ro = model.readouts[m]
            if type(ro) is not Gaussian2DReadout:
                return f"readout of mouse {m} is {type(ro).__name__}, not Gaussian2DReadout"
            if "forward" in ro.__dict__:
                return f"readout.forward of mouse {m} was overridden on the instance"
            if int(L.load().v1t_gaussian2d_backward_ws_bytes(int(b_["image"].shape[0]), gh, gw, ro.num_neurons)) <= 0:
                return f"readout backward workspace unavailable for mouse {m}"
            if model.core_shifter is not None and len(model.core_shifter[m].mlp) != 6:
                return f"core shifter of mouse {m} is not the 3-layer MLP"
        return None

    @staticmethod
    def build(trainer: "Trainer", units: t.Sequence[t.Tuple[str, t.Dict[str, torch.Tensor], int]]) -> t.Optional["_NativeStep"]:
        why = _NativeStep.unsupported(trainer, units)
        if why is not None:
            # a ~25 % slower step must not be silent: warn ONCE per trainer and reason
            if why not in trainer._fallback_warned:
                trainer._fallback_warned.add(why)
                warnings.warn(f"v1t_amd.Trainer: the native training step does not cover this configuration ({why}); "
                              f"falling back to the {trainer.fallback_path(len(units))} autograd path", RuntimeWarning, stacklevel=3)
            return None
        return _NativeStep(trainer, units)

    def __init__(self, trainer: "Trainer", units):
        model = trainer.model
        core = model.core
        lib = L.load()
        dev = core._arena.data.device
        self.sig = tuple((m, int(b["image"].shape[0])) for m, b, _ in units)
        self.B = sum(n for _, n in self.sig)
        B, T, DP = self.B, core.num_tokens, core.padded_dim
        c, h, w = core.input_shape
        f32 = dict(dtype=torch.float32, device=dev)
        self.img = torch.empty((B, c, h, w), **f32)
        self.nbeh = {0: 0, 2: 3, 3: 5}[core.behavior_mode if core.behavior_mode in (2, 3) else 0]
        self.beh = torch.empty((B, max(self.nbeh, 1)), **f32) if self.nbeh else None
        self.tokens = torch.empty((B, T, DP), **f32)
        self.gout = torch.empty((B, T, DP), **f32)
        self.ws_bytes = int(lib.v1t_vit_workspace_bytes(core._plan, B, 1))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.sb = int(lib.v1t_vit_scratch_bytes(core._plan, B))
        self.scratch = torch.empty(self.sb, dtype=torch.uint8, device=dev)
        # zeroed once per step: the per-unit loss scalars and the d shift accumulators
        nz = len(units) + 1 + sum(2 * n for _, n in self.sig)  # + the step's total loss (summed in the loss kernel, not on the host)
        self.zeros = torch.zeros(nz, **f32)
        self.tails = []
        zo = len(units) + 1
        self.loss_total = None  # the step's total loss: a FRESH one-element tensor per step (the caller may keep it; `run`)
        self.units_c = None  # ctypes table of v1t_tail_unit, rebuilt with the pointers
        self.mice_stepped = False
        C_, gh, gw = core.output_shape
        for i, (m, n) in enumerate(self.sig):
            ro = model.readouts[m]
            N = ro.num_neurons
            t_ = dict(m=m, n=n, N=N, ro=ro, loss=self.zeros[i:i + 1], dshift=self.zeros[zo:zo + 2 * n].view(n, 2), off=sum(k for _, k in self.sig[:i]))
            zo += 2 * n
            t_["shift"] = torch.empty((n, 2), **f32) if model.core_shifter is not None else None
            t_["eps"] = torch.empty((n, N, 2), **f32)
            t_["grid"] = torch.empty((n, N, 2), **f32)
            t_["dgrid"] = torch.empty((n, N, 2), **f32)
            t_["u"] = torch.empty((n, N), **f32)
            t_["yhat"] = torch.empty((n, N), **f32)
            t_["du"] = torch.empty((n, N), **f32)
            t_["rws"] = torch.empty(max(int(lib.v1t_gaussian2d_backward_ws_bytes(n, gh, gw, N)), 1), dtype=torch.uint8, device=dev)
            t_["gws"] = torch.empty(max(int(lib.v1t_readout_grid_backward_ws_bytes(n, N)), 1), dtype=torch.uint8, device=dev)
            self.tails.append(t_)
        self.geom = (C_, gh, gw, T, DP)
        self.generation = None  # arena generations the cached pointers belong to

    def _pointers(self, trainer: "Trainer") -> None:
        """Data / gradient pointers of every per-mouse parameter (views of the mouse arenas; re-derived when an arena was rebuilt)."""
        model = trainer.model
        gen = tuple(model.mouse_arena(t_["m"]).generation for t_ in self.tails)
        if gen == self.generation:
            return
        for t_ in self.tails:
            a = model.mouse_arena(t_["m"])
            a.attach_grads()
            slot = {id(s.tensor): s for s in a.slots}
            gptr = lambda p: a.grad.data_ptr() + 4 * slot[id(p)].offset  # noqa: E731
            ro = t_["ro"]
            ro.feature_storage()
            t_["feat"], t_["dfeat"], t_["FS"] = a.data.data_ptr() + 4 * slot[id(ro.features)].offset, gptr(ro.features), ro.feat_stride
            t_["bias"], t_["dbias"] = (ro.bias.data_ptr(), gptr(ro.bias)) if ro.bias is not None else (None, None)
            t_["sigma"], t_["dsigma"] = ro.sigma.data_ptr(), gptr(ro.sigma)
            if ro._predicted_grid:
                l0, l2 = ro.mu_transform[0], ro.mu_transform[2]
                t_["gd"], t_["src"] = int(ro.source_grid.shape[1]), ro.source_grid.data_ptr()
                t_["gp"] = [l0.weight.data_ptr(), l0.bias.data_ptr(), l2.weight.data_ptr(), l2.bias.data_ptr()]
                t_["dgp"] = [gptr(l0.weight), gptr(l0.bias), gptr(l2.weight), gptr(l2.bias)]
                t_["mu"], t_["dmu"] = None, None
            else:
                t_["gd"], t_["src"], t_["gp"], t_["dgp"] = 0, None, [None] * 4, [None] * 4
                t_["mu"], t_["dmu"] = ro._mu.data_ptr(), gptr(ro._mu)
            if model.core_shifter is not None:
                ml = model.core_shifter[t_["m"]].mlp
                ps = [ml[0].weight, ml[0].bias, ml[2].weight, ml[2].bias, ml[4].weight, ml[4].bias]
                t_["sp"], t_["dsp"] = [p.data_ptr() for p in ps], [gptr(p) for p in ps]
        self.generation = gen
        P = lambda x: C.c_void_p(x) if x else C.c_void_p(None)  # noqa: E731
        us = []
        for t_ in self.tails:
            u = L.TailUnit()
            u.n_images, u.n_neurons, u.image_offset, u.grid_dim = t_["n"], t_["N"], t_["off"], t_["gd"]
            u.eps_stream = ((1 + trainer.mouse_ids.index(t_["m"])) << 16) | (trainer.sharding.rank & 0xFFFF)
            u.fill_eps, u.loss_scale, u.feat_stride = 1, 1.0, t_["FS"]
            if t_["shift"] is not None:
                for k in range(6):
                    u.sp[k], u.dsp[k] = t_["sp"][k], t_["dsp"][k]
                u.shift, u.dshift = t_["shift"].data_ptr(), t_["dshift"].data_ptr()
            u.src = P(t_["src"])
            for k in range(4):
                u.gp[k], u.dgp[k] = P(t_["gp"][k]), P(t_["dgp"][k])
            # (d sigma / d mu are OVERWRITTEN by the sample-position backward, every other gradient is accumulated: one unit per mouse and step,
            # and the optimizer zeroes the arena behind it - tests that sum several calls take the gradients per call)
            u.mu, u.dmu, u.sigma, u.dsigma = P(t_["mu"]), P(t_["dmu"]), t_["sigma"], t_["dsigma"]
            u.feat, u.dfeat, u.bias, u.dbias = t_["feat"], t_["dfeat"], P(t_["bias"]), P(t_["dbias"])
            u.eps, u.grid, u.dgrid = t_["eps"].data_ptr(), t_["grid"].data_ptr(), t_["dgrid"].data_ptr()
            u.u, u.yhat, u.du, u.loss = t_["u"].data_ptr(), t_["yhat"].data_ptr(), t_["du"].data_ptr(), t_["loss"].data_ptr()
            u.rws, u.rws_bytes, u.gws, u.gws_bytes = t_["rws"].data_ptr(), t_["rws"].numel(), t_["gws"].data_ptr(), t_["gws"].numel()
            us.append(u)
        self.units_c = (L.TailUnit * len(us))(*us)

    def run(self, trainer: "Trainer", units) -> torch.Tensor:
        """One step's forward + backward. The per-mouse tails run as ONE launch per stage over all units (`v1t_tails_*`, csrc/tails.hip;
        rounds 2-3 ran a chain of ~11 small launches per mouse over three side streams - 260 launches per step, 57 of them ATen / runtime
        fills and copies):
          side stream, before the core forward is enqueued: zero fills of the step's accumulators and of the token-gradient buffer,
              `v1t_tails_prepare` (shifter forward, position noise, sample positions, counting sort of the taps: nothing of it needs the core);
          main: resize / concat -> v1t_vit_forward -> `v1t_tails_forward` (readout, ELU1 + Poisson with the step's total loss, dz gather)
              -> v1t_vit_backward;
          side stream, beside the core backward: `v1t_tails_backward` (readout / position / shifter parameter gradients) and, on one GPU,
              the mice's AdamW as one launch."""
        model = trainer.model
        core, crop = model.core, model.image_cropper
        lib = L.load()
        self._pointers(trainer)
        C_, gh, gw, T, DP = self.geom
        nu = len(units)
        main = torch.cuda.current_stream()
        st = main.cuda_stream
        side = model._side_streams(1)[0] if (model.readout_streams and nu > 1) else None
        # dtype / layout conversions of the per-mouse inputs run HERE, on the main stream in front of `start`: the side stream reads them
        # behind that event (no-ops for fp32 contiguous batches; an fp64 or strided batch would otherwise race with the shifter forward)
        pups = [b["pupil_center"].to(torch.float32).contiguous() for _, b, _ in units]
        ys = [b["response"].to(torch.float32).contiguous() for _, b, _ in units]
        trainer._eps_state = (trainer._eps_state * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        # the step's total loss gets its own storage every step: the caller may keep per-step losses on the device and read them
        # later (the reference's update_dict / log_metrics do), so it must not alias a buffer the next step zeroes and re-accumulates.
        # Allocated on the main stream; zeroed on the side stream in front of `prepared`, which the main stream waits for before the
        # loss kernel adds into it.
        self.loss_total = torch.empty(1, dtype=torch.float32, device=self.zeros.device)
        for i, ((m, b, full), t_) in enumerate(zip(units, self.tails)):
            u = self.units_c[i]
            u.pupil, u.response = pups[i].data_ptr(), ys[i].data_ptr()
            u.loss_scale = math.sqrt(trainer.ds_sizes[m] / full)
            u.fill_eps = 0 if (trainer.eps_override and trainer.eps_override.get(m) is not None) else 1
        on_side = (lambda: torch.cuda.stream(side)) if side is not None else contextlib.nullcontext
        if side is not None:
            start = torch.cuda.Event()
            start.record(main)
            side.wait_event(start)  # the previous step's readers of the shared buffers, this step's inputs
        with on_side():
            s_ = torch.cuda.current_stream().cuda_stream
            L.check(lib.v1t_fill_zero(self.zeros.data_ptr(), 4 * self.zeros.numel(), s_), "fill_zero")  # per-unit losses, d shift
            L.check(lib.v1t_fill_zero(self.loss_total.data_ptr(), 4, s_), "fill_zero")
            L.check(lib.v1t_fill_zero(self.gout.data_ptr(), 4 * self.gout.numel(), s_), "fill_zero")
            for (m, _, _), t_ in zip(units, self.tails):
                ov = trainer.eps_override.get(m) if trainer.eps_override else None
                if ov is not None:
                    t_["eps"].copy_(ov.reshape(t_["n"], t_["N"], 2))  # (tests replay the reference's draws)
                if t_["mu"] is not None:
                    with torch.no_grad():
                        t_["ro"]._mu.clamp_(min=-1, max=1)  # gaussian2d.py:212-215 (acts on the free parameter only)
            L.check(lib.v1t_tails_prepare(self.units_c, nu, trainer._eps_state, gh, gw, s_), "tails_prepare")
            if side is not None:
                prepared = torch.cuda.Event()
                prepared.record(side)
        # ---- inputs straight into the shared batch buffers: every unit's resize (or copy) and behaviour rows in one launch
        srcs = [b["image"] if (b["image"].dtype == torch.float32 and b["image"].is_contiguous()) else b["image"].to(torch.float32).contiguous() for _, b, _ in units]
        VP = C.c_void_p * nu
        behs = [b["behavior"].to(torch.float32).contiguous() for _, b, _ in units] if self.nbeh else None
        for (m, _, _), x, (_, n) in zip(units, srcs, self.sig):  # one launch reads every unit with the first unit's geometry
            if tuple(x.shape[1:]) != tuple(srcs[0].shape[1:]) or int(x.shape[0]) != n:
                raise RuntimeError(f"native step: images of mouse {m} have shape {tuple(x.shape)}, expected ({n}, {', '.join(map(str, srcs[0].shape[1:]))})")
        ih, iw = srcs[0].shape[2], srcs[0].shape[3]
        oh, ow = crop.resize if crop.resize is not None else (ih, iw)
        L.check(lib.v1t_inputs_multi(VP(*[x.data_ptr() for x in srcs]), VP(*[x.data_ptr() for x in behs]) if behs else None,
                                     VP(*[x.data_ptr() for x in pups]) if self.nbeh == 5 else None, (C.c_int * nu)(*[n for _, n in self.sig]), nu,
                                     srcs[0].shape[1], ih, iw, self.img.data_ptr(), oh, ow, L.ptr(self.beh) if self.nbeh else None,
                                     3 if self.nbeh else 0, 2 if self.nbeh == 5 else 0, st), "inputs_multi")
        # ---- shared core, one pass over all units
        seed = core._next_seed()
        L.check(lib.v1t_vit_forward(core._plan, core._arena.data.data_ptr(), core._shadow.data_ptr(), self.img.data_ptr(), L.ptr(self.beh), 0, self.B,
                                    self.ws.data_ptr(), self.ws_bytes, 1, 1, seed, None, self.tokens.data_ptr(), st), "vit_forward")
        core._last_ws = (self.ws, self.B, True)
        # ---- every unit's readout, loss and dz, one launch each (class-token row skipped)
        if side is not None:
            main.wait_event(prepared)
        zptr = self.tokens.data_ptr() + 4 * core.cls_tokens * DP
        gptr = self.gout.data_ptr() + 4 * core.cls_tokens * DP
        L.check(lib.v1t_tails_forward(self.units_c, nu, zptr, gptr, T * DP, DP, C_, gh, gw, self.loss_total.data_ptr(), st), "tails_forward")
        if side is not None:
            dz_done = torch.cuda.Event()
            dz_done.record(main)
        # ---- shared core backward (gradients accumulate into the core arena; per-block events for the data-parallel exchange)
        core._arena.attach_grads()
        evs = core._block_events
        ev_arr = (C.c_void_p * len(evs))(*[e.cuda_event for e in evs]) if evs is not None else None
        L.check(lib.v1t_vit_backward_events(core._plan, core._arena.data.data_ptr(), core._shadow.data_ptr(), self.img.data_ptr(), L.ptr(self.beh), 0, self.B,
                                            self.ws.data_ptr(), self.scratch.data_ptr(), self.sb, 1, seed, None, self.gout.data_ptr(), core._arena.grad.data_ptr(),
                                            ev_arr, st), "vit_backward")
        # ---- what the core's backward does not wait for: the tails' parameter gradients (and, on one GPU, the mice's optimizer step)
        own_opt = trainer.sharding.world == 1 and side is not None
        if side is not None:
            # (round 6, experiment 2: releasing this HBM-bound side work at the backward's first attention kernel instead of here, beside the
            # dGELU / dX GEMMs: 20.79-20.83 against 20.75-20.76 ms per step - no gain, profiles/r06_experiments.txt)
            side.wait_event(dz_done)
        with on_side():
            s_ = torch.cuda.current_stream().cuda_stream
            L.check(lib.v1t_tails_backward(self.units_c, nu, zptr, T * DP, DP, C_, gh, gw, s_), "tails_backward")
            if own_opt:
                trainer.step_mice([m for m, _, _ in units])
        if side is not None:
            main.wait_stream(side)
        self.mice_stepped = own_opt
        return self.loss_total.reshape(())


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
        # V1T_NATIVE_STEP=0 (dev): forward / backward through the nn.Module + autograd path instead of the direct C-ABI sequence
        self.native = os.environ.get("V1T_NATIVE_STEP", "1") != "0"
        self._fallback_warned: t.Set[str] = set()
        self.last_step_path: t.Optional[str] = None  # "native" | "batched-autograd" | "per-mouse": what the last train_step ran
        self._native_cache: t.Dict[t.Any, t.Optional[_NativeStep]] = {}
        self._eps_state = (int(getattr(args, "seed", 1234)) * 2654435761 + 97) & 0xFFFFFFFFFFFFFFFF
        self.eps_override: t.Optional[t.Dict[str, torch.Tensor]] = None  # tests: mouse -> (n, N, 2) position noise to replay

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
        native = None
        if self.native and (len(units) <= self.core_group or len(units) == 1) and units and units[0][1]["image"].is_cuda:
            key = tuple((m, int(b["image"].shape[0])) for m, b, _ in units)
            if key not in self._native_cache:
                self._native_cache[key] = _NativeStep.build(self, units)
            native = self._native_cache[key]
            if native is not None and any("forward" in model.readouts[m].__dict__ for m, _, _ in units):
                native = None  # a test overrode a readout's forward on the instance
        self.last_step_path = "native" if native is not None else self.fallback_path(len(units))
        if native is not None:
            losses = [native.run(self, units)]
        elif self.core_group > 1 and len(units) > 1 and core.behavior_mode != 4:
            # groups of local mouse-batches through the shared core in one pass, one backward per group (gradients sum as
            # train.py:97-111)
            single_pass = len(units) <= self.core_group
            for i in range(0, len(units), self.core_group):
                grp = units[i:i + self.core_group]
                us = model.forward_mice([(m, b) for m, b, _ in grp], activate=False, join=False)
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
        mice_left = [] if (native is not None and native.mice_stepped) else list(self.sharding.local_mice())  # (the native single-GPU step ran them beside the core's backward)
        items = [(self.model.mouse_arena(m), [(o, n, c, self.opt.group_lr(g)) for o, n, c, g in self.model.mouse_step_ranges(m)]) for m in mice_left]
        if not core.frozen:
            ca = core._arena
            if self._core_l1 is None:  # the reg_scale buffer lives on the device: one read, not one sync per step
                self._core_l1 = float(core.reg_scale) * len(self.mouse_ids)
            core_item = (ca, [(0, ca.param_floats, self._core_l1, self.opt.group_lr("core"))])
            if items and "step_arena" not in self.opt.__dict__:
                items.insert(0, core_item)  # a rank of a multi-GPU step: the core and its mice in ONE launch behind the exchange
            else:
                self.opt.step_arena(*core_item)
            core.mark_updated()
        if items:
            self.opt.step_arenas(items)
        if native is not None:
            return {"loss": losses[0]}  # the step's total, summed by the loss kernel
        return {"loss": torch.stack(losses).sum() if losses else torch.zeros((), device=core._arena.data.device)}

    def fallback_path(self, n_units: int) -> str:
        """The autograd path a step takes when the native step does not cover the configuration: all local mouse-batches through the
        shared core in one pass ("batched-autograd"), or the reference's loop over mice ("per-mouse", train.py:97-111)."""
        return "batched-autograd" if (self.core_group > 1 and n_units > 1 and self.model.core.behavior_mode != 4) else "per-mouse"

    def step_mice(self, mouse_ids: t.Sequence[str]) -> None:
        """`step_mouse` for several mice in one launch (each arena keeps its own step counter and its ranges' learning rates)."""
        self.opt.step_arenas([(self.model.mouse_arena(m), [(o, n, c, self.opt.group_lr(g)) for o, n, c, g in self.model.mouse_step_ranges(m)]) for m in mouse_ids])

    def step_mouse(self, mouse_id: str) -> None:
        """AdamW (+ L1) over one mouse's arena on the CURRENT stream: one launch per (L1 coefficient, optimizer group) run -
        readouts / image_cropper / core_shifter keep their own learning rates."""
        a = self.model.mouse_arena(mouse_id)
        self.opt.step_arena(a, [(o, n, c, self.opt.group_lr(g)) for o, n, c, g in self.model.mouse_step_ranges(mouse_id)])

    @torch.no_grad()
    def predict(self, batch: t.Dict[str, torch.Tensor], mouse_id: str) -> torch.Tensor:
        self.model.train(False)
        y, _, _ = self.model(inputs=batch["image"], mouse_id=mouse_id, behaviors=batch["behavior"], pupil_centers=batch["pupil_center"])
        return y
