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
    def build(trainer: "Trainer", units: t.Sequence[t.Tuple[str, t.Dict[str, torch.Tensor], int]]) -> t.Optional["_NativeStep"]:
        from .readout import Gaussian2DReadout

        model = trainer.model
        core, crop = model.core, model.image_cropper
        if core.behavior_mode == 4 or core.frozen or core.drop_path_rate > 0 or not units:
            return None
        if crop.image_shifter is not None or crop.crop_scale < 1 or crop.behavior_mode == 1:
            return None
        gh, gw = core.output_shape[1:]
        if gh * gw > 4096:
            return None  # the split (sort / dz / parameter) readout backward needs the sorted form: LDS histogram of <= 4096 cells (readout.hip)
        for m, b_, _ in units:
            ro = model.readouts[m]
            if type(ro) is not Gaussian2DReadout or "forward" in ro.__dict__:
                return None
            if int(L.load().v1t_gaussian2d_backward_ws_bytes(int(b_["image"].shape[0]), gh, gw, ro.num_neurons)) <= 0:
                return None
            if model.core_shifter is not None and len(model.core_shifter[m].mlp) != 6:
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
        nz = len(units) + sum(2 * n for _, n in self.sig)
        self.zeros = torch.zeros(nz, **f32)
        self.tails = []
        zo = len(units)
        self._dz_events = None
        self.mice_stepped = False
        C_, gh, gw = core.output_shape
        for i, (m, n) in enumerate(self.sig):
            ro = model.readouts[m]
            N = ro.num_neurons
            t_ = dict(m=m, n=n, N=N, ro=ro, loss=self.zeros[i:i + 1], dshift=self.zeros[zo:zo + 2 * n].view(n, 2))
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

    def run(self, trainer: "Trainer", units) -> torch.Tensor:
        """One step's forward + backward. Stream plan (kernel trace of round 2: the per-mouse tails were a 0.8 ms hole between the
        core's forward and backward - chains of ~11 small dependent kernels, four chains at a time (hardware queues)):
          side stream of mouse m, BEFORE the core forward is enqueued: everything that does not need the core's output - shifter
              forward, position noise, sample positions, and the counting sort of the taps the dz gather reads;
          main: resize / concat -> v1t_vit_forward;
          side stream, behind the forward: readout forward -> ELU1 + Poisson -> dz gather -> [event: dz of mouse m complete] ->
              parameter gradients of the readout, sample positions and shifter -> (single-GPU: this mouse's AdamW);
          main: waits for the seven dz events only -> v1t_vit_backward -> ... The rest of the tails overlaps the core backward."""
        model = trainer.model
        core, crop = model.core, model.image_cropper
        lib = L.load()
        self._pointers(trainer)
        C_, gh, gw, T, DP = self.geom
        main = torch.cuda.current_stream()
        st = main.cuda_stream
        streams = None
        if model.readout_streams and len(units) > 1:
            # V1T_TAIL_STREAMS side streams, the mice dealt over them round-robin (default 3: with the main stream that is the four
            # hardware queues HIP multiplexes streams onto - a fifth stream would share the main stream's in-order queue)
            pool = model._side_streams(max(1, min(len(units), trainer.tail_streams)))
            streams = [pool[i % len(pool)] for i in range(len(units))]
        if self._dz_events is None:
            self._dz_events = [torch.cuda.Event() for _ in units]
        # the shared buffers of this step (previous step's readers are behind us on this stream or were awaited at its end)
        self.zeros.zero_()  # per-unit loss scalars, d shift accumulators
        self.gout.zero_()
        # dtype / layout conversions of the per-mouse inputs run HERE, on the main stream in front of `start`: the side streams read them
        # behind that event (no-ops for fp32 contiguous batches; an fp64 or strided batch would otherwise race with the shifter forward)
        pups = [b["pupil_center"].to(torch.float32).contiguous() for _, b, _ in units]
        ys = [b["response"].to(torch.float32).contiguous() for _, b, _ in units]
        start = torch.cuda.Event()
        start.record(main)
        trainer._eps_state = (trainer._eps_state * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        side = lambda i: (torch.cuda.stream(streams[i]) if streams else contextlib.nullcontext())  # noqa: E731
        # ---- per mouse, core-independent part
        for i, ((m, b, full), t_) in enumerate(zip(units, self.tails)):
            n, N, ro = t_["n"], t_["N"], t_["ro"]
            pup = pups[i]
            if streams:
                streams[i].wait_event(start)  # parameters of the previous optimizer step, inputs
            with side(i):
                s_ = torch.cuda.current_stream().cuda_stream
                if t_["shift"] is not None:
                    L.check(lib.v1t_core_shifter_forward(n, pup.data_ptr(), *t_["sp"], t_["shift"].data_ptr(), s_), "core_shifter_forward")
                ov = trainer.eps_override.get(m) if trainer.eps_override else None
                if ov is not None:
                    t_["eps"].copy_(ov.reshape(n, N, 2))
                else:
                    L.check(lib.v1t_normal_fill(t_["eps"].data_ptr(), n * N * 2, trainer._eps_state, ((1 + trainer.mouse_ids.index(m)) << 16) | (trainer.sharding.rank & 0xFFFF), s_),
                            "normal_fill")
                if t_["mu"] is not None:
                    with torch.no_grad():
                        ro._mu.clamp_(min=-1, max=1)  # gaussian2d.py:212-215 (acts on the free parameter only)
                L.check(lib.v1t_readout_grid_forward(n, N, t_["gd"], t_["src"], *t_["gp"], t_["mu"], t_["sigma"], t_["eps"].data_ptr(), L.ptr(t_["shift"]),
                                                     t_["grid"].data_ptr(), s_), "readout_grid_forward")
                L.check(lib.v1t_gaussian2d_backward_parts(None, T * DP, DP, n, C_, gh, gw, N, t_["grid"].data_ptr(), None, t_["FS"], None, None, T * DP, DP,
                                                          None, None, None, t_["rws"].data_ptr(), t_["rws"].numel(), 1, s_), "gaussian2d_sort")
        # ---- inputs straight into the shared batch buffers
        off = 0
        for (m, b, _), (_, n) in zip(units, self.sig):
            src = b["image"]
            if src.dtype != torch.float32 or not src.is_contiguous():
                src = src.to(torch.float32).contiguous()
            dst = self.img[off:off + n]
            if crop.resize is not None:
                L.check(lib.v1t_resize_bilinear(src.data_ptr(), n * src.shape[1], src.shape[2], src.shape[3], dst.data_ptr(), crop.resize[0], crop.resize[1], st),
                        "resize_bilinear")
            else:
                dst.copy_(src)
            if self.nbeh:
                beh, pup = b["behavior"].to(torch.float32).contiguous(), b["pupil_center"].to(torch.float32).contiguous()
                L.check(lib.v1t_concat2(beh.data_ptr(), 3, pup.data_ptr() if self.nbeh == 5 else None, 2 if self.nbeh == 5 else 0, n,
                                        self.beh[off:off + n].data_ptr(), self.nbeh, st), "concat2")
            off += n
        # ---- shared core, one pass over all units
        seed = core._next_seed()
        L.check(lib.v1t_vit_forward(core._plan, core._arena.data.data_ptr(), core._shadow.data_ptr(), self.img.data_ptr(), L.ptr(self.beh), 0, self.B,
                                    self.ws.data_ptr(), self.ws_bytes, 1, 1, seed, None, self.tokens.data_ptr(), st), "vit_forward")
        core._last_ws = (self.ws, self.B, True)
        fwd_done = torch.cuda.Event()
        fwd_done.record(main)
        # ---- per mouse, behind the core: readout, loss, dz (-> event), then the parameter gradients (and this mouse's optimizer)
        own_opt = trainer.sharding.world == 1 and streams is not None
        off = 0
        for i, ((m, b, full), t_) in enumerate(zip(units, self.tails)):
            n, N = t_["n"], t_["N"]
            if streams:
                streams[i].wait_event(fwd_done)
            with side(i):
                s_ = torch.cuda.current_stream().cuda_stream
                zptr = self.tokens.data_ptr() + 4 * (off * T * DP + core.cls_tokens * DP)  # this unit's images, class-token row skipped
                gptr = self.gout.data_ptr() + 4 * (off * T * DP + core.cls_tokens * DP)
                L.check(lib.v1t_gaussian2d_forward(zptr, T * DP, DP, n, C_, gh, gw, N, t_["grid"].data_ptr(), t_["feat"], t_["FS"], t_["bias"], t_["u"].data_ptr(), s_),
                        "gaussian2d_forward")
                scale = math.sqrt(trainer.ds_sizes[m] / full)
                L.check(lib.v1t_elu1_poisson(t_["u"].data_ptr(), ys[i].data_ptr(), n * N, scale, 1.0, t_["yhat"].data_ptr(), t_["du"].data_ptr(), t_["loss"].data_ptr(), s_),
                        "elu1_poisson")
                bw = lambda parts: L.check(lib.v1t_gaussian2d_backward_parts(  # noqa: E731
                    zptr, T * DP, DP, n, C_, gh, gw, N, t_["grid"].data_ptr(), t_["feat"], t_["FS"], t_["du"].data_ptr(), gptr, T * DP, DP, t_["dgrid"].data_ptr(),
                    t_["dfeat"], t_["dbias"], t_["rws"].data_ptr(), t_["rws"].numel(), parts, s_), "gaussian2d_backward")
                bw(4)  # dz of this mouse's images, from the taps sorted before the forward
                self._dz_events[i].record(torch.cuda.current_stream())
            off += n
        if streams:
            for ev in self._dz_events:
                main.wait_event(ev)
        # ---- shared core backward (gradients accumulate into the core arena; per-block events for the data-parallel exchange)
        core._arena.attach_grads()
        evs = core._block_events
        ev_arr = (C.c_void_p * len(evs))(*[e.cuda_event for e in evs]) if evs is not None else None
        L.check(lib.v1t_vit_backward_events(core._plan, core._arena.data.data_ptr(), core._shadow.data_ptr(), self.img.data_ptr(), L.ptr(self.beh), 0, self.B,
                                            self.ws.data_ptr(), self.scratch.data_ptr(), self.sb, 1, seed, None, self.gout.data_ptr(), core._arena.grad.data_ptr(),
                                            ev_arr, st), "vit_backward")
        # second pass over the mice: what the core's backward does not wait for. Issued AFTER every mouse's dz chain and after
        # the core's backward because the streams share four in-order hardware queues: work enqueued
        # earlier would sit in front of another mouse's dz chain, or of the backward itself
        off = 0
        for i, ((m, b, full), t_) in enumerate(zip(units, self.tails)):
            n, N = t_["n"], t_["N"]
            with side(i):
                s_ = torch.cuda.current_stream().cuda_stream
                zptr = self.tokens.data_ptr() + 4 * (off * T * DP + core.cls_tokens * DP)
                gptr = self.gout.data_ptr() + 4 * (off * T * DP + core.cls_tokens * DP)
                L.check(lib.v1t_gaussian2d_backward_parts(zptr, T * DP, DP, n, C_, gh, gw, N, t_["grid"].data_ptr(), t_["feat"], t_["FS"], t_["du"].data_ptr(), gptr, T * DP, DP,
                                                          t_["dgrid"].data_ptr(), t_["dfeat"], t_["dbias"], t_["rws"].data_ptr(), t_["rws"].numel(), 2, s_),
                        "gaussian2d_backward")  # d grid, d features, d bias
                L.check(lib.v1t_readout_grid_backward_ws(n, N, t_["gd"], t_["src"], *t_["gp"], t_["mu"], t_["sigma"], t_["eps"].data_ptr(), t_["dgrid"].data_ptr(),
                                                         *t_["dgp"], t_["dmu"], t_["dsigma"], t_["dshift"].data_ptr() if t_["shift"] is not None else None,
                                                         t_["gws"].data_ptr(), t_["gws"].numel(), s_), "readout_grid_backward")
                if t_["shift"] is not None:
                    L.check(lib.v1t_core_shifter_backward(n, pups[i].data_ptr(), *t_["sp"], t_["dshift"].data_ptr(), *t_["dsp"], s_), "core_shifter_backward")
                if own_opt:
                    trainer.step_mouse(m)  # this mouse's arena is complete: its AdamW runs here, beside the core's backward
            off += n
        if streams:
            for s_i in dict.fromkeys(streams):
                main.wait_stream(s_i)  # the tails' parameter gradients (and optimizer steps); long done by now
        self.mice_stepped = own_opt
        return self.zeros[:len(units)]


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
        self.tail_streams = int(os.environ.get("V1T_TAIL_STREAMS", "3"))
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
        if not core.frozen:
            ca = core._arena
            if self._core_l1 is None:  # the reg_scale buffer lives on the device: one read, not one sync per step
                self._core_l1 = float(core.reg_scale) * len(self.mouse_ids)
            self.opt.step_arena(ca, [(0, ca.param_floats, self._core_l1, self.opt.group_lr("core"))])
            core.mark_updated()
        if not (native is not None and native.mice_stepped):  # (the native single-GPU step ran them on the mice's own streams)
            for mouse_id in self.sharding.local_mice():
                self.step_mouse(mouse_id)
        if native is not None:
            return {"loss": losses[0].sum()}
        return {"loss": torch.stack(losses).sum() if losses else torch.zeros((), device=core._arena.data.device)}

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
