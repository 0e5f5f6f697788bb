"""Core plugin registry + the MI355X-native "vit" core.

Host-side mirror of the reference's core interface for this hot path:
  - `register(name)` / `get_core(args)` / `Core`   <- src/v1t/models/core/core.py:8-16, 62-65, 19-59
  - `ViTCore(args, input_shape)` registered as "vit" <- src/v1t/models/core/vit.py:365-436
Same names, constructor/forward signatures, attributes (`input_shape`, `output_shape`, `frozen`,
`regularizer()`), state-dict keys and error behaviour, so that `get_core(args)(args, input_shape=...)`
and checkpoints of the reference work unchanged. The compute is NOT torch: `forward` is one
autograd node around `v1t_vit_forward` / `v1t_vit_backward` of libv1t_amd.so (hand-written gfx950
kernels). Without the library, or on a CPU tensor, it raises RuntimeError — there is no fallback.

If the reference package `v1t` is importable, `install_into_reference()` overwrites its registry
entries with these classes.
"""
from __future__ import annotations

import ctypes as C
import math
import typing as t

import torch
from torch import nn

from . import lib as L
from .flat import FlatArena, Slot

_CORES: t.Dict[str, t.Any] = dict()


def register(name: str):
    """Decorator filling the core registry (reference core/core.py:8-16)."""

    def add_to_dict(fn):
        _CORES[name] = fn
        return fn

    return add_to_dict


def get_core(args):
    """reference core/core.py:62-65"""
    if args.core not in _CORES.keys():
        raise NotImplementedError(f"Core {args.core} has not been implemented.")
    return _CORES[args.core]


class Core(nn.Module):
    """Base class (reference core/core.py:19-59)."""

    def __init__(self, args: t.Any, input_shape: t.Tuple[int, int, int], name: str = "Core"):
        super().__init__()
        self.input_shape = input_shape
        self.name = name
        self.behavior_mode = args.behavior_mode
        if args.core != "vit":
            assert self.behavior_mode != 2
        self.frozen = False
        self.verbose = getattr(args, "verbose", 0)

    def freeze(self):
        for param in self.parameters():
            param.requires_grad_(False)
        self.frozen = True

    def unfreeze(self):
        for param in self.parameters():
            param.requires_grad_(True)
        self.frozen = False

    def regularizer(self):
        raise NotImplementedError("regularizer function has not been implemented")

    def forward(self, inputs, mouse_id, behaviors, pupil_centers):
        raise NotImplementedError("forward function has not been implemented")


def find_shape(num_patches: int) -> t.Tuple[int, int]:
    """reference vit.py:411-417"""
    dim1 = math.ceil(math.sqrt(num_patches))
    while num_patches % dim1 != 0 and dim1 > 0:
        dim1 -= 1
    return dim1, num_patches // dim1


class _VitFn(torch.autograd.Function):
    """One autograd node for the whole core. `anchor` is a dummy leaf that requires grad iff the core
    is trainable; parameter gradients are accumulated in place into the core's gradient arena
    (their .grad views), so the node returns no gradients."""

    @staticmethod
    def forward(ctx, core: "ViTCore", images, behaviors, mouse_idx: int, anchor, need_bwd: int):
        # need_bwd (the workspace mode 0 / 1 / 2) is decided by the caller: grad mode is always off inside Function.forward
        B = images.shape[0]
        training = core.training
        seed = core._next_seed() if training else 0
        lib = L.load()
        # stochastic depth (DropPath, models/utils.py:121-141): one U[0,1) per sample and residual branch, as the reference
        # draws them; the kernels only apply the factor floor(keep + U) / keep
        ps = None
        if training and core.drop_path_rate > 0.0:
            ps = core._path_scale_override
            if ps is None:
                keep = 1.0 - core.drop_path_rates.to(images.device)[:, None, None]  # (blocks, 1, 1): one rate per block
                ps = torch.floor(keep + torch.rand((core.num_blocks, 2, B), dtype=torch.float32, device=images.device)) / keep
            ps = ps.to(device=images.device, dtype=torch.float32).contiguous()
            assert ps.shape == (core.num_blocks, 2, B)
        ws_bytes = lib.v1t_vit_workspace_bytes(core._plan, B, int(need_bwd))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=images.device)
        out = torch.empty((B, core.num_tokens, core.padded_dim), dtype=torch.float32, device=images.device)
        L.check(
            lib.v1t_vit_forward(core._plan, core._arena.data.data_ptr(), core._shadow.data_ptr(), images.data_ptr(),
                                L.ptr(behaviors), mouse_idx, B, ws.data_ptr(), ws_bytes, int(need_bwd), int(training), seed,
                                L.ptr(ps), out.data_ptr(), L.stream()),
            "vit_forward",
        )
        ctx.core, ctx.ws, ctx.images, ctx.behaviors = core, ws, images, behaviors
        ctx.mouse_idx, ctx.seed, ctx.training, ctx.B = mouse_idx, seed, training, B
        ctx.want_dx = bool(images.requires_grad)
        ctx.path_scale = ps
        core._last_ws = (ws, B, bool(need_bwd))
        return out

    @staticmethod
    def backward(ctx, gout):
        core = ctx.core
        lib = L.load()
        gout = gout.contiguous()
        core._arena.attach_grads()
        want_dx = ctx.want_dx and ctx.needs_input_grad[1]
        sb = (lib.v1t_vit_scratch_bytes_input if want_dx else lib.v1t_vit_scratch_bytes)(core._plan, ctx.B)
        scratch = torch.empty(sb, dtype=torch.uint8, device=gout.device)
        dx = torch.empty_like(ctx.images) if want_dx else None  # d / d(core input): vit.py:66-72, 122-129 under the reference's autograd
        evs = core._block_events  # data-parallel trainer: one event per block, recorded when its gradients are complete
        ev_arr = None
        if evs is not None:
            ev_arr = (C.c_void_p * len(evs))(*[e.cuda_event for e in evs])
        L.check(
            lib.v1t_vit_backward_input(core._plan, core._arena.data.data_ptr(), core._shadow.data_ptr(), ctx.images.data_ptr(),
                                       L.ptr(ctx.behaviors), ctx.mouse_idx, ctx.B, ctx.ws.data_ptr(), scratch.data_ptr(), sb,
                                       int(ctx.training), ctx.seed, L.ptr(ctx.path_scale), gout.data_ptr(), core._arena.grad.data_ptr(),
                                       ev_arr, L.ptr(dx), L.stream()),
            "vit_backward",
        )
        ctx.ws = None
        return None, dx, None, None, None, None


class _SplitFn(torch.autograd.Function):
    """Batch slices of the concatenated core output as separate autograd outputs; the backward gathers the slices'
    gradients into one buffer so that the core runs ONE backward for all of them."""

    @staticmethod
    def forward(ctx, tokens, *sizes):
        ctx.sizes, ctx.meta = sizes, (tokens.shape, tokens.dtype, tokens.device)
        return tuple(tokens.split(list(sizes), dim=0))

    @staticmethod
    def backward(ctx, *grads):
        shape, dtype, device = ctx.meta
        g = torch.empty(shape, dtype=dtype, device=device)
        o = 0
        for n, gi in zip(ctx.sizes, grads):
            if gi is None:
                g[o:o + n].zero_()
            else:
                g[o:o + n].copy_(gi)
            o += n
        return (g,) + (None,) * len(ctx.sizes)


class _L1Fn(torch.autograd.Function):
    """scale * sum|p| over a flat arena; backward adds scale * sign(p) into the gradient arena."""

    @staticmethod
    def forward(ctx, arena: FlatArena, start: int, n: int, scale: float, anchor):
        out = torch.zeros((), dtype=torch.float32, device=arena.data.device)
        L.check(L.load().v1t_l1_sum(arena.data.data_ptr() + 4 * start, n, scale, out.data_ptr(), L.stream()), "l1_sum")
        ctx.arena, ctx.start, ctx.n, ctx.scale = arena, start, n, scale
        return out

    @staticmethod
    def backward(ctx, g):
        a = ctx.arena
        a.attach_grads()
        # g is the 0-dim upstream gradient (1, or B_micro / B_full: train.py:71); the kernel reads it on the device - `float(g)` here was a
        # host sync in the middle of every micro-batch's backward on the reference's own loop
        g = g.to(torch.float32).contiguous()
        L.check(L.load().v1t_l1_grad_dev(a.data.data_ptr() + 4 * ctx.start, a.grad.data_ptr() + 4 * ctx.start, ctx.n, ctx.scale, g.data_ptr(), L.stream()), "l1_grad")
        return None, None, None, None, None


def cached_scalar(buf: torch.Tensor) -> float:
    """float(buf) for a 0-dim device buffer (the modules' `reg_scale`), read from the device ONCE per value: `float(tensor)` is a host
    sync, and one in `regularizer()` stalls the reference's loop behind every mouse's forward (round 5: the per-mouse path then enqueued
    its loss / regulariser kernels into an empty queue, 1.3 ms of exposed launch latency per mouse). Keyed on the buffer's storage and
    version counter, so `load_state_dict`, `.to()` or an in-place change re-reads it."""
    key = (buf.data_ptr(), buf._version, buf.device)
    c = buf.__dict__.get("_v1t_scalar")
    if c is None or c[0] != key:
        c = (key, float(buf))
        buf.__dict__["_v1t_scalar"] = c
    return c[1]


def _runs(tensors: t.Sequence[torch.Tensor]) -> t.List[t.Tuple[int, int, t.List[int]]]:
    """(first data_ptr, floats, indices) of maximal runs of fp32 tensors that sit back to back in memory (the flat arenas)."""
    order = sorted(range(len(tensors)), key=lambda i: tensors[i].data_ptr())
    runs: t.List[t.Tuple[int, int, t.List[int]]] = []
    for i in order:
        x = tensors[i]
        if runs and runs[-1][0] + 4 * runs[-1][1] == x.data_ptr():
            runs[-1] = (runs[-1][0], runs[-1][1] + x.numel(), runs[-1][2] + [i])
        else:
            runs.append((x.data_ptr(), x.numel(), [i]))
    return runs


class _L1ParamsFn(torch.autograd.Function):
    """scale * sum_p sum|p| over a list of dense fp32 GPU parameters (core_shifter.py:36-37, image_cropper.py:38-39): one `v1t_l1_sum`
    launch per run of parameters that are contiguous in memory (one, when they live in a mouse arena) instead of abs / sum / add per
    parameter; the backward adds scale * g * sign(p) straight into the attached gradient views (`v1t_l1_grad_dev`, g read on the device)
    or returns the gradients where a parameter has no such view."""

    @staticmethod
    def forward(ctx, scale: float, *params):
        out = torch.zeros((), dtype=torch.float32, device=params[0].device)
        runs = _runs(params)
        lib = L.load()
        for ptr, n, _ in runs:
            L.check(lib.v1t_l1_sum(ptr, n, scale, out.data_ptr(), L.stream()), "l1_sum")
        ctx.params, ctx.scale, ctx.runs = params, scale, runs
        return out

    @staticmethod
    def backward(ctx, g):
        params, lib = ctx.params, L.load()
        g = g.to(torch.float32).contiguous()
        grads: t.List[t.Optional[torch.Tensor]] = [None] * len(params)
        for ptr, n, idx in ctx.runs:
            sinks = [L.grad_sink(params[i]) for i in idx]
            if all(r is None for _, r in sinks) and [x[0] for x in _runs([s_ for s_, _ in sinks])] == [sinks[0][0].data_ptr()]:
                L.check(lib.v1t_l1_grad_dev(ptr, sinks[0][0].data_ptr(), n, ctx.scale, g.data_ptr(), L.stream()), "l1_grad")  # the whole run in one launch
                continue
            for i, (sink, ret) in zip(idx, sinks):
                L.check(lib.v1t_l1_grad_dev(params[i].data_ptr(), sink.data_ptr(), params[i].numel(), ctx.scale, g.data_ptr(), L.stream()), "l1_grad")
                grads[i] = ret
        return (None, *grads)


def l1_of_parameters(module: nn.Module, reg_scale: torch.Tensor):
    """`reg_scale * sum(p.abs().sum() for p in module.parameters())` (the reference's shifter regularisers). GPU: fused (above), and
    exactly 0.0 without a launch when the scale is zero (the reference's default `shifter_reg_scale` / `cropper_reg_scale`: the term
    and its gradient are 0 * ...)."""
    ps = [p for p in module.parameters()]
    if not ps or not ps[0].is_cuda or any(p.dtype != torch.float32 or not p.is_contiguous() for p in ps):
        total = 0
        for p in ps:
            total = total + p.abs().sum()
        return reg_scale * total
    scale = cached_scalar(reg_scale)
    if scale == 0.0:
        return 0.0
    if not any(p.requires_grad for p in ps) or not torch.is_grad_enabled():
        with torch.no_grad():
            return _L1ParamsFn.apply(scale, *ps)
    return _L1ParamsFn.apply(scale, *ps)


class _AttendTap(nn.Module):
    """Stands where the reference's `Attention.attend` (an nn.Softmax, vit.py:229) stands: the reference's `Recorder` registers a forward
    hook on it (utils/attention_rollout.py:28-36). The fused attention never runs a softmax module; when a hook is registered, the core
    recomputes the block's per-head probabilities after the forward (`v1t_attention_probs`) and passes them through this identity."""

    def forward(self, probabilities: torch.Tensor) -> torch.Tensor:
        return probabilities


def _seq(*mods: nn.Module) -> nn.Sequential:
    return nn.Sequential(*mods)


class _ParamBag(nn.Module):
    """Container whose only job is to hold parameters / sub-containers under the reference's names."""


@register("vit")
class ViTCore(Core):
    """MI355X-native drop-in for the reference ViTCore (vit.py:365-436).

    Reads the same `args` fields (vit.py:374-405): core_reg_scale, behavior_mode, patch_mode,
    patch_size, patch_stride, emb_dim, p_dropout, num_blocks, num_heads, mlp_dim, t_dropout, use_lsa,
    drop_path, disable_bias, output_shapes (mouse ids for behavior_mode 4). `grad_checkpointing` is
    accepted and ignored: the fused attention never materialises (B,H,T,T), so there is nothing to
    recompute (vit.py:277-284)."""

    def __init__(self, args, input_shape: t.Tuple[int, int, int], name: str = "ViTCore"):
        super().__init__(args, input_shape=input_shape, name=name)
        self.register_buffer("reg_scale", torch.tensor(float(args.core_reg_scale)))
        if not hasattr(args, "grad_checkpointing"):
            args.grad_checkpointing = False
        if args.patch_mode not in (0, 1, 2, 3):
            raise NotImplementedError(f"--patch_mode {args.patch_mode} not implemented.")
        if args.patch_mode == 2 and input_shape[0] != 1:
            raise NotImplementedError("--patch_mode 2 (Shifted Patch Tokenization) is defined for single-channel input only (vit.py:84).")
        self.drop_path_rate = float(getattr(args, "drop_path", 0.0))
        assert 0.0 <= self.drop_path_rate < 1.0
        assert 1 <= args.patch_stride <= args.patch_size
        c, h, w = input_shape
        self.mouse_ids = list(args.output_shapes.keys())
        cfg = L.VitConfig(
            in_channels=c, in_h=h, in_w=w, patch_size=args.patch_size, patch_stride=args.patch_stride,
            patch_mode=args.patch_mode, emb_dim=args.emb_dim, num_heads=args.num_heads, mlp_dim=int(args.mlp_dim),
            num_blocks=args.num_blocks, behavior_mode=self.behavior_mode if self.behavior_mode in (2, 3, 4) else 0,
            num_mice=len(self.mouse_ids), use_lsa=int(bool(args.use_lsa)), use_bias=int(not args.disable_bias),
            p_dropout=float(args.p_dropout), t_dropout=float(args.t_dropout), ln_eps=1e-5,
        )
        self._finish_init(args, cfg, c)

    def _finish_init(self, args, cfg, c: int) -> None:
        """Plan, module tree under the reference's names, flat arena, bookkeeping (shared with the CCT core, cct.py)."""
        self._cfg = cfg
        lib = L.load()
        plan = C.c_void_p()
        L.check(lib.v1t_vit_create(C.byref(cfg), C.byref(plan)), "vit_create")
        self._plan = plan
        self.num_tokens = lib.v1t_vit_tokens(plan)
        self.cls_tokens = lib.v1t_vit_cls_tokens(plan)  # 1: token 0 is the class token (ViT); 0: patches only (CCT)
        self.padded_dim = lib.v1t_vit_padded_dim(plan)
        self.emb_dim = args.emb_dim
        gh, gw = lib.v1t_vit_grid_h(plan), lib.v1t_vit_grid_w(plan)
        assert (gh, gw) == find_shape(self.num_tokens - self.cls_tokens)
        self.output_shape = (args.emb_dim, gh, gw)

        self._build_modules(args, c)
        self._arena = self._make_arena()
        self._shadow: t.Optional[torch.Tensor] = None
        self._packed_key = None
        self._anchor = torch.zeros((), requires_grad=True)
        self._seed_state = int(getattr(args, "seed", 1234)) * 1000003 + 12345
        self._last_ws = None
        self._block_events = None  # set by the data-parallel trainer (dist.MouseSharding.reduce_core_overlapped)
        self._path_scale_override = None  # tests: (num_blocks, 2, B) factors to replay instead of drawing them
        self.num_blocks = int(args.num_blocks)
        self.attach_recorder_taps()  # no-op unless v1t_amd.install_into_reference() has named the reference's Attention class
        if not hasattr(self, "drop_path_rates"):  # ViT: one DropPath module for every block (vit.py:333)
            self.drop_path_rates = torch.full((self.num_blocks,), self.drop_path_rate, dtype=torch.float32)
        # the attention-probability dropout (vit.py:263) runs at round(65536 p) / 65536 (include/v1t_amd.h): say so when that is not p within 2 %
        # (rates below 2^-17 run as 0)
        self.attention_dropout_rate = float(lib.v1t_attention_dropout_rate(float(args.t_dropout))) if float(args.t_dropout) > 0 else 0.0
        if float(args.t_dropout) > 0 and abs(self.attention_dropout_rate - float(args.t_dropout)) > 0.02 * float(args.t_dropout):
            import warnings

            warnings.warn(f"v1t_amd: attention-probability dropout runs at {self.attention_dropout_rate:.5f} (t_dropout = {float(args.t_dropout):.5f} "
                          f"quantised to 1/65536); the proj / MLP dropouts use the exact rate", stacklevel=2)

    def __del__(self):
        try:
            if getattr(self, "_plan", None):
                L.load().v1t_vit_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def _build_modules(self, args, c: int) -> None:
        D, H, M, P = args.emb_dim, args.num_heads, int(args.mlp_dim), args.patch_size
        bias = not args.disable_bias
        T = self.num_tokens
        pe = _ParamBag()
        if args.patch_mode == 0:
            pe.projection = _seq(nn.Identity(), nn.Identity(), nn.Linear(c * P * P, D))  # key projection.2.* (vit.py:68-72)
        elif args.patch_mode == 2:  # PatchShifting, Unfold, Rearrange, LayerNorm, Linear: keys projection.3.* / .4.* (vit.py:83-91)
            pd = (c + 4) * P * P
            pe.projection = _seq(nn.Identity(), nn.Identity(), nn.Identity(), nn.LayerNorm(pd), nn.Linear(pd, D))
        elif args.patch_mode == 3:  # Unfold, Rearrange, LayerNorm, Linear, LayerNorm: keys projection.2.* / .3.* / .4.* (vit.py:92-100)
            pd = c * P * P
            pe.projection = _seq(nn.Identity(), nn.Identity(), nn.LayerNorm(pd), nn.Linear(pd, D), nn.LayerNorm(D))
        else:
            conv = nn.Conv2d(c, D, kernel_size=P, stride=args.patch_stride)  # key projection.0.* (vit.py:74-82)
            nn.init.kaiming_normal_(conv.weight)
            pe.projection = _seq(conv, nn.Identity())
        pe.cls_token = nn.Parameter(torch.randn(1, 1, D))
        pe.pos_embedding = nn.Parameter(torch.randn(T, D))
        self.patch_embedding = pe

        def lin(i, o, b=True):
            m = nn.Linear(i, o, bias=b)
            nn.init.trunc_normal_(m.weight, std=0.02)  # Transformer.init_weight vit.py:338-346
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
            return m

        tr = _ParamBag()
        tr.blocks = nn.ModuleList()
        in_dim = 3 if self.behavior_mode == 2 else 5
        for _ in range(args.num_blocks):
            mha = _ParamBag()
            mha.layer_norm = nn.LayerNorm(D)
            mha.to_qkv = lin(D, 3 * H * D, False)
            mha.projection = _seq(lin(H * D, D, bias), nn.Identity())
            scale = D ** -0.5
            if args.use_lsa:
                mha.scale = nn.Parameter(torch.full((H,), scale))
                mha.register_buffer("mask", torch.nonzero(torch.eye(T, T) == 1, as_tuple=False))
                mha.register_buffer("max_value", torch.tensor(torch.finfo(torch.get_default_dtype()).max))
            else:
                mha.register_buffer("scale", torch.tensor(scale))
            mlp = _ParamBag()
            mlp.model = _seq(nn.LayerNorm(D), lin(D, M, bias), nn.Identity(), nn.Identity(), lin(M, D, bias), nn.Identity())
            block = nn.ModuleDict({"mha": mha, "mlp": mlp})
            if self.behavior_mode in (2, 3, 4):
                bm = _ParamBag()
                ids = self.mouse_ids if self.behavior_mode == 4 else ["share"]
                bm.models = nn.ModuleDict(
                    {m: _seq(lin(in_dim, D // 2, bias), nn.Identity(), nn.Identity(), lin(D // 2, D, bias), nn.Identity()) for m in ids}
                )
                block["b-mlp"] = bm
            tr.blocks.append(block)
        dp = _ParamBag()
        dp.register_buffer("keep_prop", torch.tensor(1.0 - float(getattr(args, "drop_path", 0.0))))
        tr.drop_path = dp
        self.transformer = tr

    def _make_arena(self) -> FlatArena:
        lib = L.load()
        n = lib.v1t_vit_num_tensors(self._plan)
        named = dict(self.named_parameters())
        named.update(dict(self.named_buffers()))
        slots = []
        name_buf = C.create_string_buffer(256)
        off, nd, isp = C.c_longlong(), C.c_int(), C.c_int()
        shape = (C.c_longlong * 4)()
        for i in range(n):
            L.check(lib.v1t_vit_tensor_info(self._plan, i, name_buf, 256, C.byref(off), C.byref(nd), shape, C.byref(isp)), "tensor_info")
            key = name_buf.value.decode()
            if "@" in key:  # per-mouse B-MLP: "models.@<i>" -> mouse id
                pre, rest = key.split("@", 1)
                idx, rest = rest.split(".", 1)
                key = f"{pre}{self.mouse_ids[int(idx)]}.{rest}"
            tns = named[key]
            shp = tuple(shape[j] for j in range(nd.value))
            assert tuple(tns.shape) == shp, (key, tuple(tns.shape), shp)
            slots.append(Slot(tns, off.value, tns.numel(), (lambda st, s=shp: st.view(s)), bool(isp.value)))
        assert {id(s.tensor) for s in slots if s.is_param} == {id(p) for p in self.parameters()}
        return FlatArena(slots, lib.v1t_vit_arena_floats(self._plan), lib.v1t_vit_param_floats(self._plan))

    def _next_seed(self) -> int:
        self._seed_state = (self._seed_state * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return self._seed_state

    def prepare(self) -> None:
        """Make sure parameters live in the flat arena on their current device and the bf16 weight
        shadow matches their current values (after .to(), load_state_dict, or an optimizer step)."""
        a = self._arena
        a.ensure()
        if self._anchor.device != a.data.device:
            self._anchor = torch.zeros((), device=a.data.device, requires_grad=True)
        key = (a.generation, a.version())
        if self._shadow is None or self._shadow.device != a.data.device:
            self._shadow = torch.empty(L.load().v1t_vit_shadow_bytes(self._plan), dtype=torch.uint8, device=a.data.device)
            self._packed_key = None
        if key != self._packed_key:
            L.check(L.load().v1t_vit_pack(self._plan, a.data.data_ptr(), self._shadow.data_ptr(), L.stream()), "vit_pack")
            self._packed_key = key

    def grad_buckets(self) -> t.List[t.Tuple[int, int, int]]:
        """(block, start, n) float ranges of the gradient arena in the order the backward completes them: blocks last to
        first (their attention + MLP parameters, contiguous), then (-1, ...) ranges complete only when the whole backward
        is: the patch embedding in front of block 0 and every block's BehaviorMLP tail."""
        a = self._arena
        names = {id(p): k for k, p in self.named_parameters()}
        blocks: t.Dict[int, t.List[t.Tuple[int, int]]] = {}
        late: t.List[t.Tuple[int, int]] = []
        for sl in sorted((x for x in a.slots if x.is_param), key=lambda x: x.offset):
            key = names[id(sl.tensor)]
            if key.startswith("transformer.blocks.") and ".b-mlp." not in key and ".b_mlp." not in key:
                blocks.setdefault(int(key.split(".")[2]), []).append((sl.offset, sl.numel))
            else:
                late.append((sl.offset, sl.numel))

        def runs(rs):
            out = []
            for o, n in rs:
                if out and out[-1][0] + out[-1][1] == o:
                    out[-1][1] += n
                else:
                    out.append([o, n])
            return [(int(o), int(n)) for o, n in out]

        res = []
        for k in sorted(blocks, reverse=True):
            r = runs(blocks[k])
            assert len(r) == 1, "a block's attention + MLP parameters are contiguous in the arena"
            res.append((k, r[0][0], r[0][1]))
        res += [(-1, o, n) for o, n in runs(late)]
        assert sum(n for _, _, n in res) == a.param_floats
        return res

    def fold_rank(self, rank: int) -> None:
        """Distinct dropout streams per data-parallel rank (parameters stay identical): the counter-based masks are keyed by
        (seed, site, local row), so without this every rank would drop the same positions of its own images."""
        self._seed_state = (self._seed_state ^ (0x9E3779B97F4A7C15 * (rank + 1))) & 0xFFFFFFFFFFFFFFFF if rank else self._seed_state

    def mark_updated(self) -> None:
        """Call after parameters were changed through raw pointers (fused optimizer)."""
        self._packed_key = None

    # ------------------------------------------------------------------ reference interface
    def regularizer(self):
        """L1 over ALL core parameters (vit.py:419-421), computed by a HIP reduction over the arena."""
        self.prepare()
        a = self._arena
        return _L1Fn.apply(a, 0, a.param_floats, cached_scalar(self.reg_scale), self._anchor)

    def forward_tokens(self, inputs: torch.Tensor, mouse_id: str, behaviors: torch.Tensor, pupil_centers: torch.Tensor,
                       keep_workspace: bool = False) -> torch.Tensor:
        """Token-major residual stream (B, T, DP) fp32 (CLS at t=0, columns >= emb_dim are zero).
        keep_workspace: keep every block's activations (qkv, lse2, ...) even under no_grad (attention rollout)."""
        L.require_cuda(inputs, "ViTCore.forward")
        self.prepare()
        inputs = inputs.to(torch.float32).contiguous()  # differentiable: an input that requires grad gets its gradient from v1t_vit_backward_input
        if tuple(inputs.shape[1:]) != tuple(self.input_shape):
            raise RuntimeError(f"ViTCore: expected input (B, {self.input_shape}), got {tuple(inputs.shape)}")
        beh = None
        if self.behavior_mode in (3, 4):
            beh = torch.cat((behaviors, pupil_centers), dim=-1).to(torch.float32).contiguous()
        elif self.behavior_mode == 2:
            beh = behaviors.to(torch.float32).contiguous()
        midx = self.mouse_ids.index(mouse_id) if self.behavior_mode == 4 else 0
        self._anchor.requires_grad_(next(self.parameters()).requires_grad and not self.frozen)  # freeze() flips every parameter
        # workspace mode of v1t_vit_forward: 1 = everything the backward reads; 2 = inference that keeps every block's q / k / log-sum-exp
        # (rollout, attention probabilities); 0 = inference. Modes 0 and 2 skip the planes only the backward reads (include/v1t_amd.h)
        need_bwd = 1 if (torch.is_grad_enabled() and (self._anchor.requires_grad or inputs.requires_grad)) else (2 if keep_workspace else 0)
        return _VitFn.apply(self, inputs, beh, midx, self._anchor, need_bwd)

    # ------------------------------------------------------------------ the reference's own Recorder (utils/attention_rollout.py:15-77)
    _reference_attention_cls: t.Optional[type] = None  # set by v1t_amd.install_into_reference(): v1t.models.core.vit.Attention

    def attach_recorder_taps(self, attention_cls: t.Optional[type] = None) -> None:
        """Give every block a parameter-free sub-module that IS an instance of `attention_cls` (the reference's `vit.Attention`) and has an
        `attend` child, so that the reference's `Recorder` - which collects `isinstance(m, Attention)` modules under `core.transformer` and
        hooks their `.attend` (attention_rollout.py:24-36) - finds what it looks for on the native core. The state-dict is unchanged (no
        parameters, no buffers). While such a hook is registered, `forward` keeps q / k / log-sum-exp and emits each block's (B, H, T, T)
        probabilities through the tap, in block order: `misc/extract_attention_maps.py` then runs unchanged. Memory and time are the
        reference's (175 MB of probabilities per image and default model): meant for the handful of images that script processes."""
        cls = attention_cls or type(self)._reference_attention_cls
        if cls is None or not self.cls_tokens:
            return
        for blk in self.transformer.blocks:
            mha = blk["mha"]
            tap = getattr(mha, "recorder_tap", None)
            if tap is None or not isinstance(tap, cls):
                tap = cls.__new__(cls)   # no __init__: the reference's constructor would allocate a second set of attention parameters
                nn.Module.__init__(tap)
                tap.attend = _AttendTap()
                mha.recorder_tap = tap

    def _hooked_taps(self) -> t.List[nn.Module]:
        taps = [getattr(blk["mha"], "recorder_tap", None) for blk in self.transformer.blocks]
        return [t_ for t_ in taps if t_ is not None] if any(t_ is not None and len(t_.attend._forward_hooks) > 0 for t_ in taps) else []

    @torch.no_grad()
    def _emit_attention_probabilities(self, B: int) -> None:
        lib = L.load()
        cfg, T = self._cfg, self.num_tokens
        H, DP = cfg.num_heads, self.padded_dim
        TP = (T + 3) // 4 * 4
        nqkv, nlse = B * T * 3 * H * DP * 2, B * H * T * 4
        for k, blk in enumerate(self.transformer.blocks):
            tap = getattr(blk["mha"], "recorder_tap", None)
            if tap is None:
                continue
            qkv = self.workspace_tensor("qkv", k)[:nqkv]
            lse2 = self.workspace_tensor("lse2", k)[:nlse]
            P = torch.empty((B, H, T, TP), dtype=torch.float32, device=qkv.device)
            L.check(lib.v1t_attention_probs(qkv.data_ptr(), lse2.data_ptr(), B, H, T, DP, blk["mha"].scale.data_ptr(), int(cfg.use_lsa), int(cfg.use_lsa),
                                            P.data_ptr(), TP, L.stream()), "attention_probs")
            tap.attend(P[..., :T])  # nn.Module.__call__: fires the registered forward hooks with outputs = the probabilities

    def tokens_to_output(self, tokens: torch.Tensor) -> torch.Tensor:
        c, h, w = self.output_shape
        # (B, C', h, w) exactly like vit.py:434-435, as a zero-copy strided view of the token-major buffer
        out = tokens[:, self.cls_tokens:, :c].unflatten(1, (h, w)).permute(0, 3, 1, 2)
        out._v1t_tokens = tokens  # lets the native readout skip the view chain (and its backward kernels)
        return out

    def forward(self, inputs: torch.Tensor, mouse_id: str, behaviors: torch.Tensor, pupil_centers: torch.Tensor):
        if self._hooked_taps():  # the reference's Recorder is listening
            tokens = self.forward_tokens(inputs, mouse_id, behaviors, pupil_centers, keep_workspace=True)
            self._emit_attention_probabilities(int(tokens.shape[0]))
            return self.tokens_to_output(tokens)
        return self.tokens_to_output(self.forward_tokens(inputs, mouse_id, behaviors, pupil_centers))

    def forward_many(self, inputs: t.Sequence[torch.Tensor], mouse_ids: t.Sequence[str], behaviors: t.Sequence[torch.Tensor],
                     pupil_centers: t.Sequence[torch.Tensor]) -> t.List[torch.Tensor]:
        """The shared core over several mouse-batches in ONE pass (the core does not depend on the mouse unless
        behavior_mode == 4): same outputs as one `forward` per batch, but every kernel sees the concatenated batch, which
        is what fills 256 CUs evenly (one 16-image batch is 3.25 rounds of attention-backward workgroups, i.e. 4)."""
        if self.behavior_mode == 4 and len(set(mouse_ids)) > 1:
            raise NotImplementedError("behavior_mode 4 has one BehaviorMLP per mouse: run the mice one by one")
        sizes = [int(x.shape[0]) for x in inputs]
        tokens = self.forward_tokens(torch.cat(list(inputs)), mouse_ids[0], torch.cat(list(behaviors)), torch.cat(list(pupil_centers)))
        parts = _SplitFn.apply(tokens, *sizes) if len(sizes) > 1 else (tokens,)
        return [self.tokens_to_output(p_) for p_ in parts]

    def workspace_tensor(self, name: str, block: int = 0) -> torch.Tensor:
        """Debug / test access to an intermediate of the LAST forward (see v1t_vit_workspace_offset)."""
        ws, B, save = self._last_ws
        off = L.load().v1t_vit_workspace_offset(self._plan, B, int(save), name.encode(), block)
        if off < 0:
            raise KeyError(name)
        return ws[off:]
