"""Readout plugin registry + the MI355X-native "gaussian2d" readout.

Host-side mirror of the reference's readout interface for this hot path:
  - `register(name)` / `Readout` / `Readouts`  <- src/v1t/models/readout/readout.py:10-18, 21-49, 52-85
  - `Gaussian2DReadout(args, input_shape, output_shape, ds, name)` registered as "gaussian2d"
                                              <- src/v1t/models/readout/gaussian2d.py:13-278
The per-neuron bilinear sample + feature dot + bias (gaussian2d.py:270-276) and its backward are the
HIP kernels `v1t_gaussian2d_forward/backward`; the O(N) grid bookkeeping (mu MLP on the cortical
coordinates, sigma * eps, clamp, + shifts: gaussian2d.py:188-235, 265-268) is a handful of tiny tensor
ops left to torch autograd (it returns d grid to them).

Memory layout: the `features` parameter keeps the reference's shape (1, C, 1, N) and state-dict key
but is STORED neuron-major ([N][FS], FS = C rounded up to 32) — the parameter is a strided view of
that storage — so that a wave reading one neuron's C weights, or adding its C gradient values, touches
one contiguous row.
"""
from __future__ import annotations

import typing as t

import numpy as np
import torch
from torch import nn

from . import lib as L

_READOUTS: t.Dict[str, t.Any] = dict()


def register(name: str):
    """reference readout/readout.py:10-18"""

    def add_to_dict(fn):
        _READOUTS[name] = fn
        return fn

    return add_to_dict


class Readout(nn.Module):
    """Basic readout module for a single animal (reference readout/readout.py:21-49)."""

    def __init__(self, args: t.Any, input_shape: tuple, output_shape: tuple, ds: t.Any, name: str = None):
        super().__init__()
        self.name = "Readout" if name is None else name
        self.input_shape = input_shape
        self.output_shape = output_shape
        self.neuron_coordinates = ds.dataset.coordinates
        self.register_buffer("reg_scale", torch.tensor(float(args.readout_reg_scale)))

    @property
    def num_neurons(self):
        return self.output_shape[-1]

    def initialize(self, *args: t.Any, **kwargs: t.Any):
        pass

    def regularizer(self, reduction: str):
        return self.reg_scale * sum(p.abs().sum() for p in self.parameters())


class Readouts(nn.ModuleDict):
    """Mouse ID -> Readout module (reference readout/readout.py:52-85)."""

    def __init__(self, args: t.Any, model: str, input_shape: t.Tuple[int], output_shapes: t.Dict[str, tuple], ds: t.Dict[str, t.Any]):
        super().__init__()
        if model not in _READOUTS.keys():
            raise NotImplementedError(f"Readout {model} has not been implemented.")
        self.input_shape = input_shape
        self.output_shapes = output_shapes
        readout_model = _READOUTS[model]
        for mouse_id, output_shape in self.output_shapes.items():
            self.add_module(
                name=mouse_id,
                module=readout_model(args, input_shape=input_shape, output_shape=output_shape, ds=ds[mouse_id], name=f"Mouse{mouse_id}Readout"),
            )

    def regularizer(self, mouse_id: str, reduction: str = "sum"):
        return self[str(mouse_id)].regularizer(reduction=reduction)

    def forward(self, inputs: torch.Tensor, mouse_id: str, shifts: torch.Tensor = None):
        return self[mouse_id](inputs, shifts=shifts)


class _Gaussian2dFn(torch.autograd.Function):
    """u[b,n] = sum_c F[n,c] * bilinear(z[b,:,:,c], grid[b,n]) + bias[n]  via the HIP kernels.
    z is addressed as z[b*zsb + cell*zsc + c] starting `zoff` floats into `zbuf` (token-major core
    output: zoff skips the CLS row). feat_st is the neuron-major [N][FS] storage."""

    @staticmethod
    def forward(ctx, zbuf, grid, feat_st, bias, geom, feat_param):
        zoff, zsb, zsc, B, Cc, H, W, N, FS = geom
        out = torch.empty((B, N), dtype=torch.float32, device=zbuf.device)
        grid = grid.contiguous()
        L.check(
            L.load().v1t_gaussian2d_forward(zbuf.data_ptr() + 4 * zoff, zsb, zsc, B, Cc, H, W, N, grid.data_ptr(), feat_st.data_ptr(),
                                            FS, L.ptr(bias), out.data_ptr(), L.stream()),
            "gaussian2d_forward",
        )
        ctx.save_for_backward(zbuf, grid, feat_st, bias)
        ctx.geom = geom
        ctx.feat_param = feat_param
        return out

    @staticmethod
    def backward(ctx, gout):
        zbuf, grid, feat_st, bias = ctx.saved_tensors
        zoff, zsb, zsc, B, Cc, H, W, N, FS = ctx.geom
        gout = gout.contiguous()
        need_z, need_grid, _, need_bias = ctx.needs_input_grad[0], ctx.needs_input_grad[1], None, ctx.needs_input_grad[3]
        dz = torch.zeros_like(zbuf) if need_z else None
        dgrid = torch.empty_like(grid) if need_grid else None
        need_feat = ctx.needs_input_grad[2]
        # d features / d bias are accumulated (+=) by the kernel: straight into the gradient arena when the parameter's
        # .grad is the attached arena view (same neuron-major layout as the storage), else into fresh zeroed buffers
        dfeat = dfeat_ret = None
        if need_feat:
            fp, FSn = ctx.feat_param, feat_st.shape[1]
            fg = getattr(fp, "grad", None)
            if fg is not None and fg.dtype == torch.float32 and fg.shape == fp.shape and fg.stride(1) == 1 and fg.stride(3) == FSn and fg.device == feat_st.device:
                dfeat = fg  # (1, C, 1, N) view of an [N][FS] gradient storage: element (0,0,0,0) is its first float
            else:
                dfeat = dfeat_ret = torch.zeros_like(feat_st)
        dbias = dbias_ret = None
        if bias is not None and need_bias:
            dbias, dbias_ret = L.grad_sink(bias)
        lib = L.load()
        ws_bytes = lib.v1t_gaussian2d_backward_ws_bytes(B, H, W, N) if need_z else 0
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=zbuf.device) if ws_bytes else None  # inverted tap index (csrc/readout.hip)
        L.check(
            lib.v1t_gaussian2d_backward_ws(zbuf.data_ptr() + 4 * zoff, zsb, zsc, B, Cc, H, W, N, grid.data_ptr(), feat_st.data_ptr(), FS,
                                           gout.data_ptr(), (dz.data_ptr() + 4 * zoff) if need_z else None, zsb, zsc,
                                           L.ptr(dgrid), L.ptr(dfeat), L.ptr(dbias), L.ptr(ws), ws_bytes, L.stream()),
            "gaussian2d_backward",
        )
        # d features (when not accumulated in place) is returned through the storage-shaped tensor; the parameter is a view of it
        return dz, dgrid, dfeat_ret, dbias_ret, None, None


class _GridFn(torch.autograd.Function):
    """grid[b][n] = clamp(sigma_n . eps[b][n] + mu_n, -1, 1) + shift[b] with mu from the grid predictor (or the
    free parameter) — one HIP kernel forward, one backward (v1t_readout_grid_forward/backward)."""

    @staticmethod
    def forward(ctx, B, src, W0, b0, W2, b2, mu_free, sigma, eps, shift):
        N = sigma.shape[1]
        gd = 0 if src is None else src.shape[1]
        grid = torch.empty((B, N, 2), dtype=torch.float32, device=sigma.device)
        eps = None if eps is None else eps.contiguous()
        shift = None if shift is None else shift.contiguous()
        L.check(L.load().v1t_readout_grid_forward(B, N, gd, L.ptr(src), L.ptr(W0), L.ptr(b0), L.ptr(W2), L.ptr(b2), L.ptr(mu_free), sigma.data_ptr(),
                                                  L.ptr(eps), L.ptr(shift), grid.data_ptr(), L.stream()), "readout_grid_forward")
        ctx.save_for_backward(src, W0, b0, W2, b2, mu_free, sigma, eps, shift)
        ctx.B = B
        return grid

    @staticmethod
    def backward(ctx, dgrid):
        src, W0, b0, W2, b2, mu_free, sigma, eps, shift = ctx.saved_tensors
        B, N = ctx.B, sigma.shape[1]
        gd = 0 if src is None else src.shape[1]
        dgrid = dgrid.contiguous()
        # the predictor gradients are accumulated (+=) by the kernel: straight into the arena views when attached
        sinks = [(None, None) if t_ is None else L.grad_sink(t_) for t_ in (W0, b0, W2, b2)]
        dmu = None if mu_free is None else torch.empty_like(mu_free)
        dsigma = torch.empty_like(sigma) if eps is not None else torch.zeros_like(sigma)
        dshift = None if shift is None else torch.zeros_like(shift)
        lib = L.load()
        ws = torch.empty(int(lib.v1t_readout_grid_backward_ws_bytes(B, N)), dtype=torch.uint8, device=dgrid.device)
        L.check(lib.v1t_readout_grid_backward_ws(B, N, gd, L.ptr(src), L.ptr(W0), L.ptr(b0), L.ptr(W2), L.ptr(b2), L.ptr(mu_free), sigma.data_ptr(),
                                                 L.ptr(eps), dgrid.data_ptr(), *[L.ptr(x[0]) for x in sinks], L.ptr(dmu),
                                                 dsigma.data_ptr() if eps is not None else None, L.ptr(dshift), ws.data_ptr(), ws.numel(), L.stream()),
                "readout_grid_backward")
        return (None, None, *[x[1] for x in sinks], dmu, dsigma, None, dshift)


@register("gaussian2d")
class Gaussian2DReadout(Readout):
    """MI355X-native drop-in for the reference Gaussian2DReadout (gaussian2d.py:13-278), full gaussian."""

    def __init__(self, args, input_shape: tuple, output_shape: tuple, ds: t.Any, use_bias: bool = True,
                 init_mu_range: float = 0.3, init_sigma: float = 0.1, gaussian_type: str = "full", name: str = "Gaussian2DReadout"):
        super().__init__(args, input_shape=input_shape, output_shape=output_shape, ds=ds, name=name)
        if init_mu_range > 1.0 or init_mu_range <= 0.0 or init_sigma <= 0.0:
            raise ValueError("either init_mu_range doesn't belong to [0.0, 1.0] or init_sigma_range is non-positive")
        if gaussian_type != "full":
            if gaussian_type in ("uncorrelated", "isotropic"):
                raise NotImplementedError(f"gaussian_type {gaussian_type} has no gfx950 path (reference always constructs 'full').")
            raise ValueError(f"Unknown Gaussian type {gaussian_type}.")
        self.init_mu_range = init_mu_range
        self.init_sigma = init_sigma
        self.gaussian_type = gaussian_type
        n = self.num_neurons
        c = input_shape[0]
        self.grid_shape = (1, n, 1, 2)
        self._predicted_grid = not args.disable_grid_predictor
        if args.disable_grid_predictor:
            self._mu = nn.Parameter(torch.empty(*self.grid_shape).uniform_(-init_mu_range, init_mu_range))
        else:
            dim = args.grid_predictor_dim
            src = np.asarray(self.neuron_coordinates)[:, :dim].astype(np.float32)
            self.mu_transform = nn.Sequential(nn.Linear(dim, 30), nn.ELU(), nn.Linear(30, 2), nn.Tanh())  # gaussian2d.py:113-131
            src = src - src.mean(axis=0, keepdims=True)
            src = src / np.abs(src).max()
            self.register_buffer("source_grid", torch.from_numpy(src))
        self.sigma = nn.Parameter(torch.empty(1, n, 2, 2).uniform_(-init_sigma, init_sigma))  # gaussian2d.py:182
        # features: neuron-major storage [N][FS], exposed with the reference's (1, C, 1, N) shape
        self.feat_stride = (c + 31) // 32 * 32
        st = torch.zeros(n, self.feat_stride)
        st[:, :c] = 1.0 / c  # gaussian2d.py:183
        self._feat_storage = st
        self.features = nn.Parameter(self._feature_view(st))
        self.use_bias = use_bias
        self.bias_mode = args.bias_mode
        stats = ds.dataset.response_stats
        if use_bias:  # gaussian2d.py:153-169
            if self.bias_mode == 0:
                b = torch.zeros(len(stats["mean"]))
            elif self.bias_mode == 1:
                b = torch.from_numpy(np.asarray(stats["mean"], dtype=np.float32))
            elif self.bias_mode == 2:
                b = torch.from_numpy(np.asarray(stats["mean"] / stats["std"], dtype=np.float32))
            else:
                raise NotImplementedError(f"Gaussian2dReadout: bias mode {self.bias_mode} has not been implemented.")
            self.bias = nn.Parameter(b.to(torch.float32))
        else:
            self.bias = None

    def _feature_view(self, storage: torch.Tensor) -> torch.Tensor:
        c = self.input_shape[0]
        return storage.view(self.num_neurons, self.feat_stride)[:, :c].t()[None, :, None, :]

    # storage hooks used by flat.FlatArena (model-level per-mouse arena)
    def feature_storage_numel(self) -> int:
        return self.num_neurons * self.feat_stride

    def feature_storage(self) -> torch.Tensor:
        """Neuron-major [N][FS] tensor backing `features` (rebuilt if the parameter was re-pointed or moved)."""
        f = self.features
        n, fs, c = self.num_neurons, self.feat_stride, self.input_shape[0]
        ok = tuple(f.shape) == (1, c, 1, n) and (c == 1 or f.stride(1) == 1) and (n == 1 or f.stride(3) == fs)
        st = self._feat_storage
        if (not ok) or st.device != f.device or st.data_ptr() != f.data_ptr():
            if ok:
                # parameter still has the neuron-major layout but lives elsewhere (arena / .to()): alias it
                st = torch.as_strided(f.data, (n, fs), (fs, 1))
            else:
                st = torch.zeros(n, fs, device=f.device, dtype=torch.float32)
                st[:, :c] = f.data.reshape(c, n).t()
                f.data = self._feature_view(st)
            self._feat_storage = st
        return st

    def feature_l1(self, reduction="sum"):
        l1 = self.features.abs()
        if reduction == "sum":
            l1 = l1.sum()
        elif reduction == "mean":
            l1 = l1.mean()
        return l1

    def regularizer(self, reduction="sum"):
        """gaussian2d.py:99-100. GPU, reduction "sum": one `v1t_l1_sum` launch over the neuron-major feature storage (its pad columns are
        0), and in the backward one `v1t_l1_grad_dev` into the attached gradient storage - instead of abs / sum / mul over the strided
        (1, C, 1, N) view and their three backward kernels."""
        f = self.features
        if reduction == "sum" and f.is_cuda and f.dtype == torch.float32:
            from .core import cached_scalar

            return _FeatL1Fn.apply(f, self.feature_storage(), cached_scalar(self.reg_scale))
        return self.reg_scale * self.feature_l1(reduction=reduction)

    @property
    def mu(self):
        """gaussian2d.py:188-193"""
        if self._predicted_grid:
            return self.mu_transform(self.source_grid.squeeze()).view(*self.grid_shape)
        return self._mu

    def sample_grid(self, batch_size: int, sample: t.Optional[bool] = None, eps: t.Optional[torch.Tensor] = None):
        """gaussian2d.py:195-235 (full gaussian). `eps` (B,N,2) optionally injects the normal draws."""
        mu = self.mu
        if not self._predicted_grid:
            with torch.no_grad():
                self._mu.clamp_(min=-1, max=1)
        sample = self.training if sample is None else sample
        if eps is not None:
            norm = eps.reshape(batch_size, self.num_neurons, 1, 2).to(mu.dtype)
        elif sample:
            norm = mu.new_empty(batch_size, self.num_neurons, 1, 2).normal_()
        else:
            return mu.clamp(min=-1, max=1).expand(batch_size, -1, -1, -1)
        return torch.clamp(torch.einsum("ancd,bnid->bnic", self.sigma, norm) + mu, min=-1, max=1)

    def _grid(self, B: int, sample: t.Optional[bool], eps: t.Optional[torch.Tensor], shifts: t.Optional[torch.Tensor]) -> torch.Tensor:
        """(B, N, 2) sample positions incl. shifts, through the fused HIP grid kernel."""
        sample = self.training if sample is None else sample
        if eps is not None:
            eps = eps.reshape(B, self.num_neurons, 2).to(torch.float32)
        elif sample:
            eps = torch.empty(B, self.num_neurons, 2, dtype=torch.float32, device=self.sigma.device).normal_()
        if self._predicted_grid:
            l0, l2 = self.mu_transform[0], self.mu_transform[2]
            return _GridFn.apply(B, self.source_grid, l0.weight, l0.bias, l2.weight, l2.bias, None, self.sigma, eps, shifts)
        with torch.no_grad():
            self._mu.clamp_(min=-1, max=1)  # gaussian2d.py:212-215 (acts on the free parameter only)
        return _GridFn.apply(B, None, None, None, None, None, self._mu, self.sigma, eps, shifts)

    def forward(self, inputs: torch.Tensor, sample: t.Optional[bool] = None, shifts: t.Optional[torch.Tensor] = None, eps: t.Optional[torch.Tensor] = None):
        L.require_cuda(inputs, "Gaussian2DReadout.forward")
        B, c, h, w = inputs.shape
        n = self.num_neurons
        if torch.is_grad_enabled():
            self._visited = True  # FusedAdamW.step (opt-in optimizer of the reference's own loop) steps the mice seen since its last step
        grid = self._grid(B, sample, eps, shifts)
        tokens = getattr(inputs, "_v1t_tokens", None)
        if tokens is not None and tokens.shape[0] == B:
            T, DP = tokens.shape[1], tokens.shape[2]
            zbuf, geom = tokens, ((T - h * w) * DP, T * DP, DP, B, c, h, w, n, self.feat_stride)  # skip the class-token row (ViT; the CCT core has none)
        else:
            zbuf = inputs.permute(0, 2, 3, 1).contiguous().to(torch.float32)  # (B, h, w, C) channel-last
            geom = (0, h * w * c, c, B, c, h, w, n, self.feat_stride)
        st = self.feature_storage()
        feat_st = _FeatStorageFn.apply(self.features, st) if self.features.requires_grad else st
        return _Gaussian2dFn.apply(zbuf, grid, feat_st, self.bias, geom, self.features)


class _FeatL1Fn(torch.autograd.Function):
    """scale * sum|features| over the [N][FS] storage (pad columns are exactly 0, so they add nothing and receive sign(0) = 0)."""

    @staticmethod
    def forward(ctx, feat_param, storage, scale: float):
        out = torch.zeros((), dtype=torch.float32, device=storage.device)
        L.check(L.load().v1t_l1_sum(storage.data_ptr(), storage.numel(), scale, out.data_ptr(), L.stream()), "l1_sum")
        ctx.feat_param, ctx.storage, ctx.scale = feat_param, storage, scale
        return out

    @staticmethod
    def backward(ctx, g):
        fp, st = ctx.feat_param, ctx.storage
        FS = st.shape[1]
        g = g.to(torch.float32).contiguous()
        fg = getattr(fp, "grad", None)
        ret = None
        if fg is not None and fg.dtype == torch.float32 and fg.shape == fp.shape and fg.stride(1) == 1 and fg.stride(3) == FS and fg.device == st.device:
            sink = fg  # (1, C, 1, N) view of an [N][FS] gradient storage (the arena): element (0,0,0,0) is its first float
        else:
            sink = torch.zeros_like(st)
            c = fp.shape[1]
            ret = sink[:, :c].t()[None, :, None, :]
        L.check(L.load().v1t_l1_grad_dev(st.data_ptr(), sink.data_ptr(), st.numel(), ctx.scale, g.data_ptr(), L.stream()), "l1_grad")
        return ret, None, None


class _FeatStorageFn(torch.autograd.Function):
    """Identity bridge from the (1,C,1,N) parameter to its neuron-major [N][FS] storage, so that the
    gradient produced in storage layout lands in the parameter's .grad (a view of the same layout)."""

    @staticmethod
    def forward(ctx, feat_param, storage):
        ctx.shape = feat_param.shape
        return storage.view_as(storage)

    @staticmethod
    def backward(ctx, g):
        # g: [N][FS] -> (1, C, 1, N) strided view, no copy
        c = ctx.shape[1]
        return g[:, :c].t()[None, :, None, :], None
