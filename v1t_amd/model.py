"""Model assembly around the native core / readout: ImageCropper -> Core -> CoreShifter -> Readouts -> ELU1.

Host-side mirror of src/v1t/models/model.py:50-177 (`Model`), core_shifter.py:7-69, image_cropper.py:50-140
and models/utils.py:109-118 (`ELU1`) — the thin callers on either side of the hot path (SURVEY.md §8f
rank 1). Same constructor arguments, attribute names, `forward` / `regularizer` / `get_parameters`
contracts and state-dict keys. The shifter MLPs (2->5->5->2) and the cropper are O(B) / O(image)
tensor plumbing and run as torch ops; everything between core input and readout output is HIP.
"""
from __future__ import annotations

import contextlib
import os
import typing as t

import torch
from torch import nn

from . import lib as L
from .core import get_core, l1_of_parameters
from .flat import FlatArena
from .readout import Gaussian2DReadout, Readouts


class ELU1(nn.Module):
    """ELU + 1 (reference models/utils.py:109-118). On GPU tensors uses the fused HIP elementwise kernel."""

    def __init__(self):
        super().__init__()
        self.register_buffer("one", torch.tensor(1.0))

    def forward(self, inputs: torch.Tensor):
        return _Elu1Fn.apply(inputs)


class _Elu1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u):
        L.require_cuda(u, "ELU1")
        u = u.contiguous()
        y = torch.empty_like(u)
        L.check(L.load().v1t_elu1_poisson(u.data_ptr(), None, u.numel(), 1.0, 1.0, y.data_ptr(), None, None, L.stream()), "elu1")
        ctx.save_for_backward(u, y)
        return y

    @staticmethod
    def backward(ctx, g):
        u, y = ctx.saved_tensors  # d/du (elu(u) + 1) = 1 | exp(u) = y (u <= 0)
        g = g.contiguous()
        du = torch.empty_like(u)
        L.check(L.load().v1t_elu1_backward(u.data_ptr(), y.data_ptr(), g.data_ptr(), u.numel(), du.data_ptr(), L.stream()), "elu1_backward")
        return du


class _ShifterFn(torch.autograd.Function):
    """2 -> 5 -> 5 -> 2 tanh MLP (core_shifter.py:24-40) as one HIP kernel forward / one backward."""

    @staticmethod
    def forward(ctx, pupil, W0, b0, W2, b2, W4, b4):
        B = pupil.shape[0]
        out = torch.empty((B, 2), dtype=torch.float32, device=pupil.device)
        L.check(L.load().v1t_core_shifter_forward(B, pupil.data_ptr(), W0.data_ptr(), b0.data_ptr(), W2.data_ptr(), b2.data_ptr(), W4.data_ptr(),
                                                  b4.data_ptr(), out.data_ptr(), L.stream()), "core_shifter_forward")
        ctx.save_for_backward(pupil, W0, b0, W2, b2, W4, b4)
        return out

    @staticmethod
    def backward(ctx, dshift):
        pupil, W0, b0, W2, b2, W4, b4 = ctx.saved_tensors
        sinks = [L.grad_sink(t_) for t_ in (W0, b0, W2, b2, W4, b4)]  # the kernel accumulates (+=)
        dshift = dshift.contiguous()
        L.check(L.load().v1t_core_shifter_backward(pupil.shape[0], pupil.data_ptr(), W0.data_ptr(), b0.data_ptr(), W2.data_ptr(), b2.data_ptr(),
                                                   W4.data_ptr(), b4.data_ptr(), dshift.data_ptr(), *[x[0].data_ptr() for x in sinks], L.stream()),
                "core_shifter_backward")
        return (None, *[x[1] for x in sinks])


def _tanh_mlp(widths: t.Sequence[int]) -> nn.Sequential:
    """Linear -> Tanh chain over `widths` (modules 0, 2, 4, ... are the Linears: the state-dict keys `mlp.0/2/4.*` of
    SURVEY.md Appendix C for widths (2, 5, 5, 2) / (2|5, 10, 10, 2))."""
    mods: t.List[nn.Module] = []
    for fan_in, fan_out in zip(widths[:-1], widths[1:]):
        mods += [nn.Linear(fan_in, fan_out), nn.Tanh()]
    return nn.Sequential(*mods)


def _l1(module: nn.Module, scale: torch.Tensor):
    return l1_of_parameters(module, scale)  # reg_scale * sum_p |p|.sum(), fused on the GPU (core.py)


class CoreShifter(nn.Module):
    """Pupil centre -> (dx, dy) added to the readout positions (reference core_shifter.py:7-40): tanh MLP
    in_features -> hidden x (num_layers - 1) -> 2, L1-regularised with `shifter_reg_scale`."""

    def __init__(self, args, in_features: int, hidden_features: int, num_layers: int, name: str = "CoreShifter"):
        super().__init__()
        self.name = name
        self.register_buffer("reg_scale", torch.tensor(float(getattr(args, "shifter_reg_scale", 0.0))))
        self.mlp = _tanh_mlp([in_features] + [hidden_features] * (num_layers - 1) + [2])

    def regularizer(self):
        return _l1(self, self.reg_scale)

    def forward(self, pupil_center: torch.Tensor):
        if pupil_center.is_cuda and len(self.mlp) == 6:
            m = self.mlp
            return _ShifterFn.apply(pupil_center.to(torch.float32).contiguous(), m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias)
        return self.mlp(pupil_center)


class CoreShifters(nn.ModuleDict):
    """reference core_shifter.py:43-69"""

    def __init__(self, args, mouse_ids: t.List[str], input_channels: int, hidden_features: int, num_layers: int):
        super().__init__()
        for mouse_id in mouse_ids:
            self.add_module(mouse_id, CoreShifter(args, input_channels, hidden_features, num_layers, name=f"Mouse{mouse_id}CoreShifter"))

    def regularizer(self, mouse_id: str):
        return self[mouse_id].regularizer()

    def forward(self, pupil_centers: torch.Tensor, mouse_id: str):
        return self[mouse_id](pupil_centers)


class ImageShifter(nn.Module):
    """(pupil | behaviour + pupil) -> crop-window offset in [-max_shift, max_shift]^2 (reference image_cropper.py:10-47):
    the same tanh MLP shape as the core shifter with 10 hidden units, scaled by `max_shift`. O(B) work: torch ops."""

    def __init__(self, args, max_shift: float, in_features: int, hidden_features: int, num_layers: int, name: str = "ImageShifter"):
        super().__init__()
        self.name = name
        self.register_buffer("max_shift", torch.tensor(max_shift))
        self.register_buffer("reg_scale", torch.tensor(float(args.cropper_reg_scale)))
        self.mlp = _tanh_mlp([in_features] + [hidden_features] * (num_layers - 1) + [2])

    def regularizer(self):
        return _l1(self, self.reg_scale)

    def forward(self, behaviors: torch.Tensor, pupil_centers: torch.Tensor):
        return self.max_shift * self.mlp(torch.cat((behaviors, pupil_centers), dim=-1))


def _crop_nearest(inputs: torch.Tensor, grid: torch.Tensor, shifts: t.Optional[torch.Tensor]) -> torch.Tensor:
    """Nearest-neighbour crop (F.grid_sample(mode="nearest", align_corners=True), image_cropper.py:126-133) as one
    gather kernel. Nearest sampling has no gradient w.r.t. the grid (the reference's shifter receives exact zeros
    through it), so the result is a constant of the graph: the ImageShifter learns from its L1 term only."""
    L.require_cuda(inputs, "ImageCropper")
    if inputs.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("ImageCropper: the gradient with respect to the raw image through the nearest-neighbour crop has no gfx950 kernel "
                                  "(center_crop < 1 or shift_mode 1 / 3 / 4); differentiate with respect to the core input instead")
    src = inputs.detach().to(torch.float32).contiguous()
    b, c, ih, iw = src.shape
    oh, ow = grid.shape[1], grid.shape[2]
    out = torch.empty((b, c, oh, ow), dtype=torch.float32, device=src.device)
    sh = None if shifts is None else shifts.detach().to(torch.float32).contiguous()
    L.check(L.load().v1t_crop_nearest(src.data_ptr(), b, c, ih, iw, grid.data_ptr(), None if sh is None else sh.data_ptr(), out.data_ptr(), oh, ow,
                                      L.stream()), "crop_nearest")
    return out


class _ResizeFn(torch.autograd.Function):
    """torchvision Resize(antialias=False) of the cropper (image_cropper.py:96-99, 134-135) as one kernel each way; the backward is the
    adjoint the reference gets from autograd of F.interpolate (d response / d raw image: MEIs, saliency)."""

    @staticmethod
    def forward(ctx, src: torch.Tensor, oh: int, ow: int):
        b, c, ih, iw = src.shape
        out = torch.empty((b, c, oh, ow), dtype=torch.float32, device=src.device)
        L.check(L.load().v1t_resize_bilinear(src.data_ptr(), b * c, ih, iw, out.data_ptr(), oh, ow, L.stream()), "resize_bilinear")
        ctx.shape = (b, c, ih, iw, oh, ow)
        return out

    @staticmethod
    def backward(ctx, g):
        b, c, ih, iw, oh, ow = ctx.shape
        g = g.to(torch.float32).contiguous()
        din = torch.empty((b, c, ih, iw), dtype=torch.float32, device=g.device)
        L.check(L.load().v1t_resize_bilinear_backward(g.data_ptr(), b * c, ih, iw, din.data_ptr(), oh, ow, L.stream()), "resize_bilinear_backward")
        return din, None, None


class ImageCropper(nn.Module):
    """reference image_cropper.py:50-140: center crop (nearest grid_sample over a [-crop, crop] grid, moved per image
    by the learned ImageShifter for shift_mode 1/3/4), bilinear 144x256 -> 36x64 (torchvision Resize(antialias=False)
    == F.interpolate(bilinear, align_corners=False)), optional behaviour-as-channels (behavior_mode 1)."""

    RESIZED = (36, 64)  # what the cropper hands to the core for the Sensorium recordings (image_cropper.py:96-99)

    def __init__(self, args, ds: t.Dict[str, t.Any]):
        super().__init__()
        self.shift_mode, self.behavior_mode = args.shift_mode, args.behavior_mode
        self.input_shape = args.input_shape
        channels, height, width = args.input_shape
        scale = float(args.center_crop)
        self.crop_scale = args.center_crop
        # window of the crop: the full frame, or int(scale * size) pixels spanning [-scale, scale] of the normalised frame
        self.crop_h = height if scale >= 1 else int(height * scale)
        self.crop_w = width if scale >= 1 else int(width * scale)
        ys = torch.linspace(-scale, scale, self.crop_h)
        xs = torch.linspace(-scale, scale, self.crop_w)
        # grid[0, i, j] = (x_j, y_i): the (1, h, w, 2) sampling grid F.grid_sample expects; state-dict key `image_cropper.grid`
        self.register_buffer("grid", torch.stack((xs[None, :].expand(self.crop_h, -1), ys[:, None].expand(-1, self.crop_w)), dim=-1)[None].contiguous())
        # learned per-mouse window shift: shift_mode 1 / 3 see the pupil centre (2), shift_mode 4 behaviours + pupil centre (5)
        self.image_shifter = None
        if self.shift_mode in (1, 3, 4):
            n_in = 5 if self.shift_mode == 4 else 2
            self.image_shifter = nn.ModuleDict()
            for mouse_id in ds.keys():
                self.image_shifter[mouse_id] = ImageShifter(args, max_shift=1 - scale, in_features=n_in, hidden_features=10, num_layers=3,
                                                            name=f"Mouse{mouse_id}ImageShifter")
        self.resize = self.RESIZED if (args.resize_image == 1 and args.ds_name != "franke2022") else None
        out_h, out_w = self.resize if self.resize is not None else (self.crop_h, self.crop_w)
        self.output_shape = (channels + (3 if self.behavior_mode == 1 else 0), out_h, out_w)

    def regularizer(self, mouse_id: str):
        return 0 if self.image_shifter is None else self.image_shifter[mouse_id].regularizer()

    def forward(self, inputs: torch.Tensor, mouse_id: str, behaviors: torch.Tensor, pupil_centers: torch.Tensor):
        grid = self.grid.expand(inputs.size(0), -1, -1, -1)
        outputs = inputs
        shifts = None
        if self.image_shifter is not None:
            b = behaviors if self.shift_mode == 4 else behaviors[:, :0]
            shifts = self.image_shifter[mouse_id](b, pupil_centers)
            grid = grid + shifts[:, None, None, :]
        if shifts is not None or self.crop_scale < 1:  # at crop 1 without shifts the sampled grid is the identity
            outputs = _crop_nearest(inputs, self.grid, shifts)
        if self.resize is not None:
            src = outputs.to(torch.float32).contiguous()
            L.require_cuda(src, "ImageCropper")
            outputs = _ResizeFn.apply(src, int(self.resize[0]), int(self.resize[1]))
        if self.behavior_mode == 1:
            h, w = outputs.size(2), outputs.size(3)
            outputs = torch.concat((outputs, behaviors[:, :, None, None].expand(-1, -1, h, w)), dim=1)
        return outputs, grid


class Model(nn.Module):
    """reference model.py:50-177"""

    def __init__(self, args: t.Any, ds: t.Dict[str, t.Any], name: str = "Model"):
        super().__init__()
        assert isinstance(args.output_shapes, dict), "output_shapes must be a dictionary of mouse_id and output_shape"
        self.name = name
        self.input_shape = args.input_shape
        self.output_shapes = args.output_shapes
        self.shift_mode = args.shift_mode
        self.add_module("image_cropper", ImageCropper(args, ds=ds))
        self.add_module("core", get_core(args)(args, input_shape=self.image_cropper.output_shape))
        if self.shift_mode in (2, 3, 4):
            self.add_module("core_shifter", CoreShifters(args, mouse_ids=list(ds.keys()), input_channels=2, hidden_features=5, num_layers=3))
        else:
            self.core_shifter = None
        self.add_module("readouts", Readouts(args, model=args.readout, input_shape=self.core.output_shape, output_shapes=self.output_shapes, ds=ds))
        self.elu1 = ELU1()
        self._mouse_arenas: t.Dict[str, FlatArena] = {}
        self._mouse_l1: t.Dict[str, t.Tuple[int, t.List[t.Tuple[int, int, float, str]]]] = {}
        self._streams: t.List[t.Any] = []
        self.readout_streams = os.environ.get("V1T_READOUT_STREAMS", "1") != "0"
        if self.readout_streams and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            # parameters used on a mouse's side stream accumulate their gradient there: intended (see forward_mice)
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    def get_parameters(self, core_lr: float):
        """reference model.py:112-139"""
        params = []
        if not self.core.frozen:
            params.append({"params": self.core.parameters(), "lr": core_lr, "name": "core"})
        params.append({"params": self.readouts.parameters(), "name": "readouts"})
        # group ORDER matters: torch optimizer state dicts are positional (golden G12 pins it for shift_mode 2 and 4)
        if self.image_cropper.image_shifter is not None:
            params.append({"params": self.image_cropper.parameters(), "name": "image_cropper"})
        if self.core_shifter is not None:
            params.append({"params": self.core_shifter.parameters(), "name": "core_shifter"})
        return params

    def regularizer(self, mouse_id: str):
        """reference model.py:141-149"""
        terms = []
        if not self.core.frozen:
            terms.append(self.core.regularizer())
        terms.append(self.readouts.regularizer(mouse_id=mouse_id))
        terms.append(self.image_cropper.regularizer(mouse_id=mouse_id))
        if self.core_shifter is not None:
            terms.append(self.core_shifter.regularizer(mouse_id=mouse_id))
        # the reference's `reg = 0; reg += term` chain, without an add kernel for the terms that are exactly the number 0 (no image
        # shifter; a shifter whose reg_scale is 0)
        reg = 0
        for x in terms:
            if isinstance(x, (int, float)) and x == 0:
                continue
            reg = x if (isinstance(reg, (int, float)) and reg == 0) else reg + x
        return reg

    def forward(self, inputs: torch.Tensor, mouse_id: str, behaviors: torch.Tensor, pupil_centers: torch.Tensor, activate: bool = True):
        images, image_grids = self.image_cropper(inputs, mouse_id=mouse_id, behaviors=behaviors, pupil_centers=pupil_centers)
        outputs = self.core(images, mouse_id=mouse_id, behaviors=behaviors, pupil_centers=pupil_centers)
        shifts = None
        if self.core_shifter is not None:
            shifts = self.core_shifter(pupil_centers, mouse_id=mouse_id)
        outputs = self.readouts(outputs, mouse_id=mouse_id, shifts=shifts)
        if activate:
            outputs = self.elu1(outputs)
        return outputs, images, image_grids

    def forward_mice(self, batches: t.Sequence[t.Tuple[str, t.Dict[str, torch.Tensor]]], activate: bool = True, join: bool = True) -> t.List[torch.Tensor]:
        """`forward` for several (mouse_id, batch) pairs with ONE pass through the shared core (ViTCore.forward_many);
        cropper, shifter and readout stay per mouse. Returns the per-pair outputs in order. The per-mouse tails run on side
        streams; with `join` (default) the current stream waits for them before the outputs are returned, so they can be used
        like any other tensor. `join=False` (the trainer: the loss and its backward stay on the mouse's stream) leaves that to
        the caller (`join_streams`)."""
        images = [self.image_cropper(b["image"], mouse_id=m, behaviors=b["behavior"], pupil_centers=b["pupil_center"])[0] for m, b in batches]
        zs = self.core.forward_many(images, [m for m, _ in batches], [b["behavior"] for _, b in batches], [b["pupil_center"] for _, b in batches])
        outs = []
        # the per-mouse tails (shifter, grid, readout, later the loss and their backward) are chains of small latency-bound
        # kernels that do not depend on each other across mice: one side stream per mouse lets them overlap. Autograd
        # runs each node's backward on the stream its forward ran on and orders streams at the graph edges.
        streams = self._side_streams(len(batches)) if (self.readout_streams and len(batches) > 1 and zs[0].is_cuda) else None
        main = torch.cuda.current_stream() if streams else None
        for i, ((m, b), z) in enumerate(zip(batches, zs)):
            if streams:
                streams[i].wait_stream(main)
                ctx = torch.cuda.stream(streams[i])
            else:
                ctx = contextlib.nullcontext()
            with ctx:
                shifts = self.core_shifter(b["pupil_center"], mouse_id=m) if self.core_shifter is not None else None
                y = self.readouts(z, mouse_id=m, shifts=shifts)
                outs.append(self.elu1(y) if activate else y)
        self._last_streams = streams
        if streams and join:
            for s_, y in zip(streams, outs):
                main.wait_stream(s_)
                y.record_stream(main)  # allocated on the side stream, used on this one
        return outs

    def _side_streams(self, n: int):
        while len(self._streams) < n:
            self._streams.append(torch.cuda.Stream())
        return self._streams[:n]

    def join_streams(self) -> None:
        """Make the current stream wait for the side streams of the last forward_mice (call before using its outputs
        on the current stream)."""
        if getattr(self, "_last_streams", None):
            cur = torch.cuda.current_stream()
            for s_ in self._last_streams:
                cur.wait_stream(s_)

    # ------------------------------------------------------------------ flat per-mouse arenas (fused optimizer / DDP)
    def mouse_arena(self, mouse_id: str) -> FlatArena:
        return mouse_arena(self, mouse_id)

    def mouse_step_ranges(self, mouse_id: str) -> t.List[t.Tuple[int, int, float, str]]:
        return mouse_step_ranges(self, mouse_id)

    def mouse_l1_ranges(self, mouse_id: str) -> t.List[t.Tuple[int, int, float]]:
        """(start, n, L1 coefficient) runs of `mouse_step_ranges`."""
        return [(o, n, c) for o, n, c, _ in self.mouse_step_ranges(mouse_id)]


# ---------------------------------------------------------------------- flat per-mouse arenas: module-level, so that they also serve the
# REFERENCE's own `Model` (model.py:50-177) built over the native classes by install_into_reference() - it has the same attributes
# (`readouts`, `core_shifter`, `image_cropper.image_shifter`) but none of this class's methods; `FusedAdamW.for_model` goes through these
def mouse_arena(model: nn.Module, mouse_id: str) -> FlatArena:
    """All per-mouse parameters (readout + core shifter + image shifter) in one flat arena; `features` first, in
    neuron-major storage, so the L1 term and the feature kernel see one contiguous [N][FS] block."""
    arenas = model.__dict__.setdefault("_mouse_arenas", {})
    a = arenas.get(mouse_id)
    if a is None:
        ro = model.readouts[mouse_id]
        params = [ro.features] + [p for p in ro.parameters() if p is not ro.features]
        if getattr(model, "core_shifter", None) is not None:
            params += list(model.core_shifter[mouse_id].parameters())
        if model.image_cropper.image_shifter is not None:
            params += list(model.image_cropper.image_shifter[mouse_id].parameters())
        special = {}
        if isinstance(ro, Gaussian2DReadout):
            special[id(ro.features)] = (ro.feature_storage_numel(), ro._feature_view)
        a = FlatArena.from_params(params, special)
        arenas[mouse_id] = a
    a.ensure()
    return a


def mouse_step_ranges(model: nn.Module, mouse_id: str) -> t.List[t.Tuple[int, int, float, str]]:
    """(start, n, L1 coefficient, optimizer group) runs over the mouse arena for the fused L1 + AdamW step. The
    coefficients are the terms Model.regularizer adds for this mouse (readout features gaussian2d.py:99-100, core
    shifter core_shifter.py:21-22, image shifter image_cropper.py:38-39; 0 elsewhere); the group name selects the
    learning rate (model.py:112-139: readouts / image_cropper / core_shifter are separate optimizer groups)."""
    a = mouse_arena(model, mouse_id)
    cache = model.__dict__.setdefault("_mouse_l1", {})
    cached = cache.get(mouse_id)
    if cached is not None and cached[0] == a.generation:
        return cached[1]
    ro = model.readouts[mouse_id]
    tag = {id(p): (0.0, "readouts") for p in ro.parameters()}
    tag[id(ro.features)] = (float(ro.reg_scale), "readouts")  # reg_scale buffers live on the device: read once, not per step
    if getattr(model, "core_shifter", None) is not None:
        cs = model.core_shifter[mouse_id]
        c = float(cs.reg_scale)
        tag.update({id(p): (c, "core_shifter") for p in cs.parameters()})
    if model.image_cropper.image_shifter is not None:
        sh = model.image_cropper.image_shifter[mouse_id]
        c = float(sh.reg_scale)
        tag.update({id(p): (c, "image_cropper") for p in sh.parameters()})
    runs: t.List[list] = []
    for s in a.slots:
        c, grp = tag[id(s.tensor)]
        if runs and runs[-1][2] == c and runs[-1][3] == grp and runs[-1][0] + runs[-1][1] == s.offset:
            runs[-1][1] += s.numel
        else:
            runs.append([s.offset, s.numel, c, grp])
    out = [(int(o), int(n), float(c), str(g)) for o, n, c, g in runs]
    cache[mouse_id] = (a.generation, out)
    return out
