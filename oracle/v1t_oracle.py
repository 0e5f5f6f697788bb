"""
ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (torch CPU tensor math, fp32 or fp64) of the reference hot path
`train.py --core vit --readout gaussian2d` of bryanlimy/V1T. Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the
product package `v1t_amd` never does (its ops raise when the HIP library is missing).

Parity pinning: the reference ships no tests or golden vectors, so this oracle
is pinned against outputs of the reference itself, imported in the build container by
`oracle/gen_golden.py`, and against the fixtures that script commits under `tests/golden/`
(`tests/test_oracle_golden.py`, `tests/test_oracle_vs_reference.py`).

Every function cites the reference file:line (relative to /root/reference) it restates. The
math is written out from elementary tensor ops (index gathers, explicit layer-norm / softmax /
erf-GELU / bilinear taps) instead of calling the fused `torch.nn` modules the reference uses, so
that it is an independent statement of the algorithm; gradients come from autograd over these
elementary ops.

Weights are passed as a flat dict with the reference's state-dict keys.
Stochastic pieces (dropout masks, readout eps) are explicit inputs so that a HIP kernel's own
counter-based masks can be replayed here bit for bit.
"""

from __future__ import annotations

import math
import typing as t
from dataclasses import dataclass, field

import torch

Tensor = torch.Tensor
SD = t.Dict[str, Tensor]


@dataclass
class Config:
    """Fields mirror the argparse names the reference constructors read (vit.py:374-405,
    gaussian2d.py:53-79, model.py:60-105)."""

    input_shape: t.Tuple[int, int, int] = (1, 36, 64)  # core input (C,H,W), post-cropper
    patch_size: int = 8
    patch_stride: int = 1
    patch_mode: int = 0
    num_blocks: int = 4
    num_heads: int = 4
    emb_dim: int = 155
    mlp_dim: int = 488
    behavior_mode: int = 3
    use_lsa: bool = False
    disable_bias: bool = False
    p_dropout: float = 0.0229
    t_dropout: float = 0.2544
    drop_path: float = 0.0
    core_reg_scale: float = 0.5379
    readout_reg_scale: float = 0.0076
    shifter_reg_scale: float = 0.0
    cropper_reg_scale: float = 0.0
    center_crop: float = 1.0  # < 1: input_shape above is the CROPPED shape (int(h * crop), int(w * crop)), raw_input_shape the image's
    raw_input_shape: t.Optional[t.Tuple[int, int, int]] = None
    shift_mode: int = 2
    disable_grid_predictor: bool = False
    grid_predictor_dim: int = 2
    bias_mode: int = 0
    mouse_ids: t.Tuple[str, ...] = ("A",)
    num_neurons: t.Dict[str, int] = field(default_factory=lambda: {"A": 256})
    core: str = "vit"      # "vit" (core/vit.py) or "cct" (core/cct.py: conv tokenizer, no class token, head dim emb_dim / heads^2)
    pos_emb: str = "sine"  # cct only (train.py:603-605): "sine" | "none"

    @property
    def grid_hw(self) -> t.Tuple[int, int]:
        c, h, w = self.input_shape
        if self.core == "cct":  # Conv2d(kernel patch_size, stride, padding 3) then MaxPool2d(3, 2, 1): cct.py:46-56
            ch = (h + 6 - self.patch_size) // self.patch_stride + 1
            cw = (w + 6 - self.patch_size) // self.patch_stride + 1
            return (ch + 2 - 3) // 2 + 1, (cw + 2 - 3) // 2 + 1
        nh = (h - self.patch_size) // self.patch_stride + 1
        nw = (w - self.patch_size) // self.patch_stride + 1
        return nh, nw

    @property
    def num_patches(self) -> int:
        nh, nw = self.grid_hw
        return nh * nw

    @property
    def latent_hw(self) -> t.Tuple[int, int]:
        return find_shape(self.num_patches)


def find_shape(num_patches: int) -> t.Tuple[int, int]:
    """vit.py:411-417 — largest divisor <= ceil(sqrt(L)) is dim1, L // dim1 is dim2."""
    d1 = math.ceil(math.sqrt(num_patches))
    while d1 > 0 and num_patches % d1 != 0:
        d1 -= 1
    return d1, num_patches // d1


# --------------------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm over the last dim (vit.py:220,145): biased variance, eps inside sqrt."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() default = exact erf form (vit.py:147)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def softmax_lastdim(x: Tensor) -> Tensor:
    """nn.Softmax(dim=-1) (vit.py:226)."""
    m = x.max(dim=-1, keepdim=True).values
    e = torch.exp(x - m)
    return e / e.sum(dim=-1, keepdim=True)


def linear(x: Tensor, w: Tensor, b: t.Optional[Tensor]) -> Tensor:
    y = x @ w.transpose(-1, -2)
    return y if b is None else y + b


# None = the reference's arithmetic. A number = ALSO emulate the one documented saturation of the product under test (the
# attention output and the GELU output are stored as fp16 planes, written saturated at +-65504): used by ONE test that drives activations
# past fp16's range and checks that the product degrades as documented (finite, clamped) instead of overflowing.
F16_PLANE_MAX: t.Optional[float] = None


def _plane(x: Tensor) -> Tensor:
    return x if F16_PLANE_MAX is None else x.clamp(-F16_PLANE_MAX, F16_PLANE_MAX)


def apply_mask(x: Tensor, mask: t.Optional[Tensor], p: float) -> Tensor:
    """nn.Dropout in train mode with an explicit keep-mask (1 = keep): x * mask / (1 - p)."""
    if mask is None:
        return x
    return x * mask.to(x.dtype) * (1.0 / (1.0 - p))


def unfold_patches(x: Tensor, patch: int, stride: int) -> Tensor:
    """nn.Unfold(kernel_size=P, stride=s) followed by `b c l -> b l c` (vit.py:69-70).
    Feature order inside a patch is (c, kh, kw); patches are row-major over (r, q)."""
    b, c, h, w = x.shape
    nh = (h - patch) // stride + 1
    nw = (w - patch) // stride + 1
    rr = (torch.arange(nh) * stride)[:, None] + torch.arange(patch)[None, :]  # (nh,P)
    cc = (torch.arange(nw) * stride)[:, None] + torch.arange(patch)[None, :]  # (nw,P)
    p = x[:, :, rr[:, None, :, None], cc[None, :, None, :]]  # (B,C,nh,nw,P,P)
    return p.permute(0, 2, 3, 1, 4, 5).reshape(b, nh * nw, c * patch * patch)


def patch_shifting(x: Tensor, patch: int) -> Tensor:
    """PatchShifting vit.py:15-38: concat input with its 4 diagonal half-patch shifts."""
    s = patch // 2
    b, c, h, w = x.shape
    pad = x.new_zeros(b, c, h + 2 * s, w + 2 * s)
    pad[:, :, s : s + h, s : s + w] = x
    lu = pad[..., : h, : w]
    ru = pad[..., : h, 2 * s : 2 * s + w]
    lb = pad[..., 2 * s : 2 * s + h, : w]
    rb = pad[..., 2 * s : 2 * s + h, 2 * s : 2 * s + w]
    return torch.cat([x, lu, ru, lb, rb], dim=1)


# --------------------------------------------------------------------------------------
# ViT core (vit.py)
# --------------------------------------------------------------------------------------
def patch_embed(cfg: Config, sd: SD, x: Tensor, mask: t.Optional[Tensor] = None, pfx: str = "core.") -> Tensor:
    """Image2Patches.forward vit.py:122-129 with modes vit.py:65-102."""
    p = pfx + "patch_embedding."
    P, s = cfg.patch_size, cfg.patch_stride
    mode = cfg.patch_mode
    if mode == 0:
        e = linear(unfold_patches(x, P, s), sd[p + "projection.2.weight"], sd[p + "projection.2.bias"])
    elif mode == 1:
        w = sd[p + "projection.0.weight"]  # (D,C,P,P) conv weight
        e = linear(unfold_patches(x, P, s), w.reshape(w.shape[0], -1), sd[p + "projection.0.bias"])
    elif mode == 2:
        u = unfold_patches(patch_shifting(x, P), P, s)
        u = layer_norm(u, sd[p + "projection.3.weight"], sd[p + "projection.3.bias"])
        e = linear(u, sd[p + "projection.4.weight"], sd[p + "projection.4.bias"])
    elif mode == 3:
        u = unfold_patches(x, P, s)
        u = layer_norm(u, sd[p + "projection.2.weight"], sd[p + "projection.2.bias"])
        e = linear(u, sd[p + "projection.3.weight"], sd[p + "projection.3.bias"])
        e = layer_norm(e, sd[p + "projection.4.weight"], sd[p + "projection.4.bias"])
    else:
        raise NotImplementedError(f"--patch_mode {mode} not implemented.")
    b = x.shape[0]
    cls = sd[p + "cls_token"].expand(b, 1, -1)
    out = torch.cat([cls, e], dim=1) + sd[p + "pos_embedding"]
    return apply_mask(out, mask, cfg.p_dropout)


def behavior_mlp(cfg: Config, sd: SD, k: int, v: Tensor, mouse_id: str, pfx: str = "core.") -> Tensor:
    """BehaviorMLP.forward vit.py:157-202: Linear -> tanh -> (dropout p=0) -> Linear -> tanh."""
    key = mouse_id if cfg.behavior_mode == 4 else "share"
    p = f"{pfx}transformer.blocks.{k}.b-mlp.models.{key}."
    h = torch.tanh(linear(v, sd[p + "0.weight"], sd.get(p + "0.bias")))
    return torch.tanh(linear(h, sd[p + "3.weight"], sd.get(p + "3.bias")))


def attention(
    cfg: Config,
    sd: SD,
    k: int,
    x: Tensor,
    masks: t.Optional[t.Dict[str, Tensor]] = None,
    record: t.Optional[t.List[Tensor]] = None,
    pfx: str = "core.",
) -> Tensor:
    """Attention.mha vit.py:267-275 + scaled_dot_product_attention vit.py:253-265.
    head dim = emb_dim, inner = emb_dim*heads (vit.py:218); `(h d)` split has h slow (vit.py:225)."""
    p = f"{pfx}transformer.blocks.{k}.mha."
    masks = masks or {}
    b, n, d = x.shape
    h = cfg.num_heads
    z = layer_norm(x, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"])
    qkv = linear(z, sd[p + "to_qkv.weight"], None)
    q, kk, v = qkv.split(h * d, dim=-1)
    q = q.reshape(b, n, h, d).permute(0, 2, 1, 3)
    kk = kk.reshape(b, n, h, d).permute(0, 2, 1, 3)
    v = v.reshape(b, n, h, d).permute(0, 2, 1, 3)
    scale = sd[p + "scale"]
    dots = q @ kk.transpose(-1, -2)
    if cfg.use_lsa:
        dots = dots * scale.reshape(1, h, 1, 1)
        eye = torch.eye(n, dtype=torch.bool)
        dots = dots.masked_fill(eye, -torch.finfo(torch.float32).max)
    else:
        dots = dots * scale
    attn = softmax_lastdim(dots)
    if record is not None:
        record.append(attn.detach().clone())  # Recorder hook: P before dropout (attention_rollout.py:28-36)
    # "attn_p": effective rate of a replayed mask (the HIP kernels run the attention-P dropout at round(65536 p) / 65536)
    attn = apply_mask(attn, masks.get(f"attn{k}"), masks.get("attn_p", cfg.t_dropout))
    o = _plane((attn @ v).permute(0, 2, 1, 3).reshape(b, n, h * d))
    o = linear(o, sd[p + "projection.0.weight"], sd.get(p + "projection.0.bias"))
    return apply_mask(o, masks.get(f"proj{k}"), cfg.t_dropout)


def mlp(cfg: Config, sd: SD, k: int, x: Tensor, masks: t.Optional[t.Dict[str, Tensor]] = None, pfx: str = "core.") -> Tensor:
    """MLP.forward vit.py:132-154: LN -> Linear -> GELU -> Dropout -> Linear -> Dropout."""
    p = f"{pfx}transformer.blocks.{k}.mlp.model."
    masks = masks or {}
    z = layer_norm(x, sd[p + "0.weight"], sd[p + "0.bias"])
    hdn = gelu_erf(linear(z, sd[p + "1.weight"], sd.get(p + "1.bias")))
    hdn = _plane(apply_mask(hdn, masks.get(f"fc1{k}"), cfg.t_dropout))
    y = linear(hdn, sd[p + "4.weight"], sd.get(p + "4.bias"))
    return apply_mask(y, masks.get(f"fc2{k}"), cfg.t_dropout)


def vit_tokens(
    cfg: Config,
    sd: SD,
    x: Tensor,
    mouse_id: str,
    behaviors: Tensor,
    pupil_centers: Tensor,
    masks: t.Optional[t.Dict[str, Tensor]] = None,
    record: t.Optional[t.List[Tensor]] = None,
    taps: t.Optional[t.Dict[str, Tensor]] = None,
    pfx: str = "core.",
) -> Tensor:
    """ViTCore.forward vit.py:423-433 + Transformer.forward vit.py:348-362, token-major (B,T,D).
    DropPath (models/utils.py:121-141) is the identity in eval mode or at drop_path=0; in train mode the branch output
    is `(y / keep) * floor(keep + U[0,1))` with one draw per sample (:136-140). The draws are not made here: pass
    masks["drop_path"] = {(block, "mha" | "mlp"): (B,) tensor of 0/1} (absent = identity, i.e. eval)."""
    masks = masks or {}
    dpm = masks.get("drop_path") or {}
    keep = 1.0 - cfg.drop_path

    def drop_path(y: Tensor, key) -> Tensor:
        m = dpm.get(key)
        return y if m is None else (y / keep) * m.to(y.dtype)[:, None, None]

    out = patch_embed(cfg, sd, x, masks.get("patch"), pfx=pfx)
    if taps is not None:
        taps["patch_embed"] = out
    if cfg.behavior_mode in (3, 4):
        v = torch.cat([behaviors, pupil_centers], dim=-1)
    else:
        v = behaviors
    for k in range(cfg.num_blocks):
        if cfg.behavior_mode in (2, 3, 4):
            out = out + behavior_mlp(cfg, sd, k, v, mouse_id, pfx=pfx)[:, None, :]
        out = drop_path(attention(cfg, sd, k, out, masks, record, pfx=pfx), (k, "mha")) + out
        if taps is not None:
            taps[f"mha{k}"] = out
        out = drop_path(mlp(cfg, sd, k, out, masks, pfx=pfx), (k, "mlp")) + out
        if taps is not None:
            taps[f"mlp{k}"] = out
    return out


def vit_core(cfg: Config, sd: SD, x: Tensor, mouse_id: str, behaviors: Tensor, pupil_centers: Tensor, **kw) -> Tensor:
    """vit.py:434-435: drop CLS, `b (h w) c -> b c h w`."""
    tok = vit_tokens(cfg, sd, x, mouse_id, behaviors, pupil_centers, **kw)
    h, w = cfg.latent_hw
    b, _, d = tok.shape
    return tok[:, 1:, :].reshape(b, h, w, d).permute(0, 3, 1, 2)


# --------------------------------------------------------------------------------------
# CCT core (core/cct.py)
# --------------------------------------------------------------------------------------
def cct_tokenizer(cfg: Config, sd: SD, x: Tensor, mask: t.Optional[Tensor] = None, pfx: str = "core.") -> Tensor:
    """Tokenizer.forward cct.py:87-103: conv (no bias, padding 3) -> ReLU -> MaxPool2d(3, 2, 1) -> `b c h w -> b (h w) c`
    -> + pos_embedding (a buffer for "sine") -> Dropout(p_dropout)."""
    y = torch.nn.functional.conv2d(x, sd[pfx + "tokenizer.conv2d.weight"], None, stride=cfg.patch_stride, padding=3)
    y = torch.nn.functional.max_pool2d(torch.relu(y), kernel_size=3, stride=2, padding=1)
    b, d, h, w = y.shape
    y = y.permute(0, 2, 3, 1).reshape(b, h * w, d)
    if cfg.pos_emb != "none":
        y = y + sd[pfx + "tokenizer.pos_embedding"].to(y.dtype)
    return apply_mask(y, mask, cfg.p_dropout)


def cct_attention(cfg: Config, sd: SD, k: int, x: Tensor, masks: t.Optional[t.Dict[str, Tensor]] = None, pfx: str = "core.") -> Tensor:
    """Attention.forward cct.py:128-143: inner = emb_dim // heads is the width of each of q, k, v; `b n (h d) -> b h n d` then cuts
    it into `heads` heads of inner // heads columns; q * scale (the buffer inner^-0.5) before the product."""
    p = f"{pfx}transformer.blocks.{k}.mha."
    masks = masks or {}
    b, n, _ = x.shape
    h = cfg.num_heads
    z = layer_norm(x, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"])
    q, kk, v = linear(z, sd[p + "qkv.weight"], None).chunk(3, dim=-1)
    d = q.shape[-1] // h
    q = q.reshape(b, n, h, d).permute(0, 2, 1, 3) * sd[p + "scale"]
    kk = kk.reshape(b, n, h, d).permute(0, 2, 1, 3)
    v = v.reshape(b, n, h, d).permute(0, 2, 1, 3)
    attn = softmax_lastdim(q @ kk.transpose(-1, -2))
    attn = apply_mask(attn, masks.get(f"attn{k}"), masks.get("attn_p", cfg.t_dropout))
    o = (attn @ v).permute(0, 2, 1, 3).reshape(b, n, h * d)
    o = linear(o, sd[p + "projection.0.weight"], sd.get(p + "projection.0.bias"))
    return apply_mask(o, masks.get(f"proj{k}"), cfg.t_dropout)


def cct_tokens(cfg: Config, sd: SD, x: Tensor, mouse_id: str, behaviors: Tensor, pupil_centers: Tensor,
               masks: t.Optional[t.Dict[str, Tensor]] = None, taps: t.Optional[t.Dict[str, Tensor]] = None, pfx: str = "core.") -> Tensor:
    """CCTCore.forward cct.py:305-316 + TransformerBlock.forward cct.py:189-197. DropPath: block k has its own rate
    linspace(0, drop_path, blocks)[k] (cct.py:219); draws are passed as masks["drop_path"] like vit_tokens."""
    masks = masks or {}
    dpm = masks.get("drop_path") or {}
    rates = [cfg.drop_path * k / max(cfg.num_blocks - 1, 1) for k in range(cfg.num_blocks)] if cfg.num_blocks > 1 else [0.0]

    def drop_path(y: Tensor, key) -> Tensor:
        m = dpm.get(key)
        return y if m is None else (y / (1.0 - rates[key[0]])) * m.to(y.dtype)[:, None, None]

    out = cct_tokenizer(cfg, sd, x, masks.get("patch"), pfx=pfx)
    if taps is not None:
        taps["patch_embed"] = out
    v = torch.cat([behaviors, pupil_centers], dim=-1) if cfg.behavior_mode in (3, 4) else behaviors
    for k in range(cfg.num_blocks):
        if cfg.behavior_mode in (3, 4):
            key = mouse_id if cfg.behavior_mode == 4 else "share"
            p = f"{pfx}transformer.blocks.{k}.b_mlp.models.{key}."
            hdn = torch.tanh(linear(v, sd[p + "0.weight"], sd.get(p + "0.bias")))
            out = out + torch.tanh(linear(hdn, sd[p + "3.weight"], sd.get(p + "3.bias")))[:, None, :]
        out = drop_path(cct_attention(cfg, sd, k, out, masks, pfx=pfx), (k, "mha")) + out
        if taps is not None:
            taps[f"mha{k}"] = out
        p = f"{pfx}transformer.blocks.{k}.mlp."
        z = layer_norm(out, sd[p + "0.weight"], sd[p + "0.bias"])
        hdn = apply_mask(gelu_erf(linear(z, sd[p + "1.weight"], sd.get(p + "1.bias"))), masks.get(f"fc1{k}"), cfg.t_dropout)
        y = apply_mask(linear(hdn, sd[p + "4.weight"], sd.get(p + "4.bias")), masks.get(f"fc2{k}"), cfg.t_dropout)
        out = drop_path(y, (k, "mlp")) + out
        if taps is not None:
            taps[f"mlp{k}"] = out
    return out


def cct_core(cfg: Config, sd: SD, x: Tensor, mouse_id: str, behaviors: Tensor, pupil_centers: Tensor, **kw) -> Tensor:
    """cct.py:316: every token, `b (h w) c -> b c h w`."""
    tok = cct_tokens(cfg, sd, x, mouse_id, behaviors, pupil_centers, **kw)
    h, w = cfg.latent_hw
    b, _, d = tok.shape
    return tok.reshape(b, h, w, d).permute(0, 3, 1, 2)


def core_regularizer(cfg: Config, sd: SD, param_keys: t.Iterable[str]) -> Tensor:
    """ViTCore.regularizer vit.py:419-421: reg_scale * sum |p| over all core parameters."""
    return cfg.core_reg_scale * sum(sd[k].abs().sum() for k in param_keys)


# --------------------------------------------------------------------------------------
# shifter + readout (core_shifter.py, gaussian2d.py)
# --------------------------------------------------------------------------------------
def core_shifter(sd: SD, mouse_id: str, pupil_centers: Tensor) -> Tensor:
    """CoreShifter.forward core_shifter.py:24-40 with num_layers=3 (model.py:86-92): 2->5->5->2, tanh each."""
    p = f"core_shifter.{mouse_id}.mlp."
    h = torch.tanh(linear(pupil_centers, sd[p + "0.weight"], sd[p + "0.bias"]))
    h = torch.tanh(linear(h, sd[p + "2.weight"], sd[p + "2.bias"]))
    return torch.tanh(linear(h, sd[p + "4.weight"], sd[p + "4.bias"]))


def readout_mu(cfg: Config, sd: SD, mouse_id: str) -> Tensor:
    """Gaussian2DReadout.mu gaussian2d.py:188-193 (+ init_grid_predictor :113-131): (N,2)."""
    p = f"readouts.{mouse_id}."
    if cfg.disable_grid_predictor:
        return sd[p + "_mu"].reshape(-1, 2)
    g = sd[p + "source_grid"]
    h = linear(g, sd[p + "mu_transform.0.weight"], sd[p + "mu_transform.0.bias"])
    h = torch.where(h > 0, h, torch.expm1(h))  # nn.ELU
    return torch.tanh(linear(h, sd[p + "mu_transform.2.weight"], sd[p + "mu_transform.2.bias"]))


def readout_grid(cfg: Config, sd: SD, mouse_id: str, batch: int, eps: t.Optional[Tensor], shifts: t.Optional[Tensor]) -> Tensor:
    """sample_grid gaussian2d.py:195-235 (full gaussian: einsum 'ancd,bnid->bnic') then
    `+ shifts` AFTER the clamp (gaussian2d.py:267-268). eps None = eval (grid = mu). -> (B,N,2)."""
    mu = readout_mu(cfg, sd, mouse_id)  # (N,2)
    sigma = sd[f"readouts.{mouse_id}.sigma"].reshape(-1, 2, 2)  # (N,c,d)
    if eps is None:
        g = mu[None].expand(batch, -1, -1)
    else:
        g = torch.einsum("ncd,bnd->bnc", sigma, eps) + mu[None]
    g = g.clamp(-1.0, 1.0)
    if shifts is not None:
        g = g + shifts[:, None, :]
    return g


def bilinear_sample(z: Tensor, grid: Tensor) -> Tensor:
    """F.grid_sample(bilinear, zeros padding, align_corners=True) at one point per neuron
    (gaussian2d.py:270). z (B,C,H,W), grid (B,N,2) with (x,y) -> (B,C,N). SURVEY Appendix A.1 step 7."""
    b, c, hh, ww = z.shape
    px = (grid[..., 0] + 1.0) * 0.5 * (ww - 1)
    py = (grid[..., 1] + 1.0) * 0.5 * (hh - 1)
    x0 = torch.floor(px)
    y0 = torch.floor(py)
    ax = px - x0
    ay = py - y0
    x0 = x0.long()
    y0 = y0.long()
    zf = z.reshape(b, c, hh * ww)
    out = z.new_zeros(b, c, grid.shape[1])
    for dx, dy, wgt in ((0, 0, (1 - ax) * (1 - ay)), (1, 0, ax * (1 - ay)), (0, 1, (1 - ax) * ay), (1, 1, ax * ay)):
        xi = x0 + dx
        yi = y0 + dy
        ok = (xi >= 0) & (xi <= ww - 1) & (yi >= 0) & (yi <= hh - 1)
        idx = (yi.clamp(0, hh - 1) * ww + xi.clamp(0, ww - 1))[:, None, :].expand(b, c, -1)
        out = out + torch.gather(zf, 2, idx) * (wgt * ok.to(z.dtype))[:, None, :]
    return out


def gaussian2d_readout(
    cfg: Config, sd: SD, mouse_id: str, z: Tensor, eps: t.Optional[Tensor] = None, shifts: t.Optional[Tensor] = None
) -> Tensor:
    """Gaussian2DReadout.forward gaussian2d.py:237-278 -> (B,N) pre-activation."""
    p = f"readouts.{mouse_id}."
    grid = readout_grid(cfg, sd, mouse_id, z.shape[0], eps, shifts)
    s = bilinear_sample(z, grid)  # (B,C,N)
    feat = sd[p + "features"].reshape(1, z.shape[1], -1)
    out = (s * feat).sum(dim=1)
    if (p + "bias") in sd:
        out = out + sd[p + "bias"]
    return out


class _Elu(torch.autograd.Function):
    """nn.ELU (alpha = 1) exactly as ATen evaluates it: forward expm1(u) for u <= 0 (so elu(u) + 1 is
    quantised to multiples of 2^-24 near 0), backward exp(u) computed from the INPUT
    (elu_backward with is_result = False), which stays non-zero where expm1(u) + 1 has rounded to 0."""

    @staticmethod
    def forward(ctx, u):
        ctx.save_for_backward(u)
        return torch.where(u > 0, u, torch.expm1(u))

    @staticmethod
    def backward(ctx, g):
        (u,) = ctx.saved_tensors
        return g * torch.where(u > 0, torch.ones_like(u), torch.exp(u))


def elu1(u: Tensor) -> Tensor:
    """ELU1 models/utils.py:109-118: nn.ELU()(u) + 1."""
    return _Elu.apply(u) + 1.0


EPS32 = float(torch.finfo(torch.float32).eps)


def poisson_loss(y_true: Tensor, y_pred: Tensor, ds_size: float, batch_size: int) -> Tensor:
    """PoissonLoss.forward losses.py:153-166 + scale_ds losses.py:114-119."""
    yt, yp = y_true + EPS32, y_pred + EPS32
    loss = (yp - yt * torch.log(yp)).sum()
    return math.sqrt(ds_size / batch_size) * loss


def correlation(y1: Tensor, y2: Tensor, dim: int = 0, eps: float = 1e-8) -> Tensor:
    """losses.py:43-58."""
    a = (y1 - y1.mean(dim=dim, keepdim=True)) / (y1.std(dim=dim, unbiased=False, keepdim=True) + eps)
    b = (y2 - y2.mean(dim=dim, keepdim=True)) / (y2.std(dim=dim, unbiased=False, keepdim=True) + eps)
    return (a * b).mean(dim=dim)


# --------------------------------------------------------------------------------------
# pre-core stage (image_cropper.py) — "next" row §8(f)1
# --------------------------------------------------------------------------------------
def resize_bilinear(x: Tensor, out_hw: t.Tuple[int, int]) -> Tensor:
    """torchvision Resize(antialias=False) on a tensor = bilinear, align_corners=False
    (image_cropper.py:96-99). Explicit half-pixel taps with edge clamping."""
    b, c, h, w = x.shape
    oh, ow = out_hw

    def taps(n_in: int, n_out: int):
        src = (torch.arange(n_out, dtype=x.dtype) + 0.5) * (n_in / n_out) - 0.5
        src = src.clamp(min=0.0)
        i0 = torch.floor(src).long().clamp(max=n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        return i0, i1, src - i0.to(x.dtype)

    r0, r1, fr = taps(h, oh)
    c0, c1, fc = taps(w, ow)
    top = x[:, :, r0][:, :, :, c0] * (1 - fc) + x[:, :, r0][:, :, :, c1] * fc
    bot = x[:, :, r1][:, :, :, c0] * (1 - fc) + x[:, :, r1][:, :, :, c1] * fc
    return top * (1 - fr)[:, None] + bot * fr[:, None]


def image_cropper(x: Tensor, resize: t.Optional[t.Tuple[int, int]] = (36, 64)) -> Tensor:
    """ImageCropper.forward image_cropper.py:120-140 at center_crop=1, shift_mode in (0,2):
    the nearest-neighbour grid_sample over the identity grid returns the image itself."""
    return x if resize is None else resize_bilinear(x, resize)


def image_shifter(cfg: Config, sd: SD, mouse_id: str, behaviors: Tensor, pupil_centers: Tensor) -> Tensor:
    """ImageShifter.forward image_cropper.py:41-47 as ImageCropper builds it (num_layers=3, hidden 10, :82-88):
    (pupil | behaviour + pupil for shift_mode 4) -> 10 -> 10 -> 2, tanh after every layer, times max_shift = 1 - crop."""
    p = f"image_cropper.image_shifter.{mouse_id}.mlp."
    x = torch.cat((behaviors, pupil_centers), dim=-1) if cfg.shift_mode == 4 else pupil_centers
    h = torch.tanh(linear(x, sd[p + "0.weight"], sd[p + "0.bias"]))
    h = torch.tanh(linear(h, sd[p + "2.weight"], sd[p + "2.bias"]))
    h = torch.tanh(linear(h, sd[p + "4.weight"], sd[p + "4.bias"]))
    return h * (1.0 - cfg.center_crop)


def crop_nearest(x: Tensor, crop_scale: float, shifts: t.Optional[Tensor]) -> Tensor:
    """The crop of ImageCropper.forward (image_cropper.py:101-110, 126-133): identity grid linspace(-s, s, int(size * s))
    per axis (+ per-image (x, y) shifts), F.grid_sample(mode="nearest", align_corners=True, zeros padding): source pixel
    = nearbyint((g + 1) / 2 * (size - 1)) (round half to even), outside the image -> 0. No gradient reaches the shifts."""
    b, c, h, w = x.shape
    ch, cw = (h, w) if crop_scale >= 1 else (int(h * crop_scale), int(w * crop_scale))
    # the sampling coordinates are fp32 in the reference whatever precision the rest of the oracle runs in
    gy = torch.linspace(-crop_scale, crop_scale, ch, dtype=torch.float32)[None, :].expand(b, -1)
    gx = torch.linspace(-crop_scale, crop_scale, cw, dtype=torch.float32)[None, :].expand(b, -1)
    if shifts is not None:
        sh = shifts.detach().to(torch.float32)
        gx = gx + sh[:, 0:1]
        gy = gy + sh[:, 1:2]
    iy = torch.round((gy + 1) / 2 * (h - 1)).long()
    ix = torch.round((gx + 1) / 2 * (w - 1)).long()
    oky = (iy >= 0) & (iy < h)
    okx = (ix >= 0) & (ix < w)
    out = x[torch.arange(b)[:, None, None], :, iy.clamp(0, h - 1)[:, :, None], ix.clamp(0, w - 1)[:, None, :]]  # (b, ch, cw, c)
    out = out * (oky[:, :, None, None] & okx[:, None, :, None]).to(x.dtype)
    return out.permute(0, 3, 1, 2)


def model_forward_raw(cfg: Config, sd: SD, raw: Tensor, mouse_id: str, behaviors: Tensor, pupil_centers: Tensor, **kw) -> Tensor:
    """Model.forward model.py:151-177 from the RAW image: crop (+ learned image shift for shift_mode 1/3/4), no resize,
    behaviour-as-channels for behavior_mode 1, then the core-input path of model_forward."""
    shifts = image_shifter(cfg, sd, mouse_id, behaviors, pupil_centers) if cfg.shift_mode in (1, 3, 4) else None
    x = crop_nearest(raw, cfg.center_crop, shifts)
    if cfg.behavior_mode == 1:
        x = torch.cat([x, behaviors[:, :, None, None].expand(-1, -1, x.shape[2], x.shape[3]).to(x.dtype)], dim=1)
    return model_forward(cfg, sd, x, mouse_id, behaviors, pupil_centers, **kw)


# --------------------------------------------------------------------------------------
# Model.forward / train step (model.py:151-177, train.py:42-111)
# --------------------------------------------------------------------------------------
def model_forward(
    cfg: Config,
    sd: SD,
    x: Tensor,
    mouse_id: str,
    behaviors: Tensor,
    pupil_centers: Tensor,
    eps: t.Optional[Tensor] = None,
    masks: t.Optional[t.Dict[str, Tensor]] = None,
    activate: bool = True,
    taps: t.Optional[t.Dict[str, Tensor]] = None,
) -> Tensor:
    """x is the CORE input (post-cropper)."""
    z = (cct_core if cfg.core == "cct" else vit_core)(cfg, sd, x, mouse_id, behaviors, pupil_centers, masks=masks, taps=taps)
    if taps is not None:
        taps["core"] = z
    shifts = core_shifter(sd, mouse_id, pupil_centers) if cfg.shift_mode in (2, 3, 4) else None
    u = gaussian2d_readout(cfg, sd, mouse_id, z, eps=eps, shifts=shifts)
    if taps is not None:
        taps["readout"] = u
    return elu1(u) if activate else u


def core_param_keys(sd: SD) -> t.List[str]:
    """Keys of nn.Parameters of the core (buffers `reg_scale`, `scale` w/o LSA, `keep_prop`, LSA `mask`/`max_value` excluded)."""
    out = []
    for k in sd:
        if not k.startswith("core."):
            continue
        leaf = k.rsplit(".", 1)[-1]
        if leaf in ("reg_scale", "keep_prop", "mask", "max_value") or k.endswith("tokenizer.pos_embedding"):
            continue
        if leaf == "scale" and sd[k].dim() == 0:
            continue
        out.append(k)
    return out


def readout_param_keys(sd: SD, mouse_id: str) -> t.List[str]:
    p = f"readouts.{mouse_id}."
    return [k for k in sd if k.startswith(p) and k.rsplit(".", 1)[-1] not in ("source_grid", "reg_scale")]


def shifter_param_keys(sd: SD, mouse_id: str) -> t.List[str]:
    p = f"core_shifter.{mouse_id}."
    return [k for k in sd if k.startswith(p) and not k.endswith("reg_scale")]


def image_shifter_param_keys(sd: SD, mouse_id: str) -> t.List[str]:
    p = f"image_cropper.image_shifter.{mouse_id}.mlp."
    return [k for k in sd if k.startswith(p)]


def regularizer(cfg: Config, sd: SD, mouse_id: str) -> Tensor:
    """Model.regularizer model.py:141-149: core L1 + readout feature L1 (+ shifter L1 * scale)."""
    reg = core_regularizer(cfg, sd, core_param_keys(sd))
    reg = reg + cfg.readout_reg_scale * sd[f"readouts.{mouse_id}.features"].abs().sum()
    if cfg.shift_mode in (2, 3, 4) and cfg.shifter_reg_scale != 0.0:
        reg = reg + cfg.shifter_reg_scale * sum(sd[k].abs().sum() for k in shifter_param_keys(sd, mouse_id))
    if cfg.shift_mode in (1, 3, 4) and cfg.cropper_reg_scale != 0.0:  # ImageShifter.regularizer image_cropper.py:38-39
        reg = reg + cfg.cropper_reg_scale * sum(sd[k].abs().sum() for k in image_shifter_param_keys(sd, mouse_id))
    return reg


def total_loss(
    cfg: Config,
    sd: SD,
    batch: t.Dict[str, Tensor],
    mouse_id: str,
    ds_size: float,
    eps: t.Optional[Tensor] = None,
    masks: t.Optional[t.Dict[str, Tensor]] = None,
    batch_size: t.Optional[int] = None,
) -> t.Tuple[Tensor, Tensor, Tensor]:
    """One micro-batch of train_step train.py:56-72 -> (loss, reg_loss, y_pred)."""
    b = batch["image"].shape[0]
    full = b if batch_size is None else batch_size
    fwd = model_forward_raw if cfg.raw_input_shape is not None else model_forward  # raw: batch["image"] is pre-cropper
    y = fwd(cfg, sd, batch["image"], mouse_id, batch["behavior"], batch["pupil_center"], eps=eps, masks=masks)
    loss = poisson_loss(batch["response"], y, ds_size, full)
    reg = (b / full) * regularizer(cfg, sd, mouse_id)
    return loss, reg, y


def adamw_step(
    params: t.Dict[str, Tensor],
    grads: t.Dict[str, Tensor],
    state: t.Dict[str, t.Dict[str, Tensor]],
    step: int,
    lr: float,
    beta1: float = 0.9,
    beta2: float = 0.9999,
    eps: float = 1e-8,
    weight_decay: float = 0.0,
) -> None:
    """torch.optim.AdamW as configured at train.py:216-223 (wd=0): in-place on `params`."""
    for k, p in params.items():
        g = grads[k]
        st = state.setdefault(k, {"m": torch.zeros_like(p), "v": torch.zeros_like(p)})
        if weight_decay != 0.0:
            p.mul_(1.0 - lr * weight_decay)
        st["m"].mul_(beta1).add_(g, alpha=1.0 - beta1)
        st["v"].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        bc1 = 1.0 - beta1**step
        bc2 = 1.0 - beta2**step
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(st["m"], denom, value=-lr / bc1)


# --------------------------------------------------------------------------------------
# attention rollout (utils/attention_rollout.py)
# --------------------------------------------------------------------------------------
def attention_rollout_row(attn: Tensor) -> Tensor:
    """attention_rollout attention_rollout.py:92-118 for ONE sample, up to the pre-normalisation
    heat vector: attn (L,H,T,T) -> J_last[0,1:] (T-1,). Full matrix chain as the reference does."""
    a = attn.max(dim=1).values
    a = a + torch.eye(a.shape[-1], dtype=a.dtype)
    a = a / a.sum(dim=-1, keepdim=True)
    j = a[0]
    for n in range(1, a.shape[0]):
        j = a[n] @ j
    return j[0, 1:]


def attention_rollout(attn: Tensor, image_hw: t.Tuple[int, int]) -> Tensor:
    """attention_rollout.py:118-122: reshape to find_shape, min-max normalise, bilinear resize."""
    heat = attention_rollout_row(attn)
    heat = heat.reshape(find_shape(heat.numel()))
    heat = (heat - heat.min()) / (heat.max() - heat.min())
    return resize_bilinear(heat[None, None], image_hw)[0, 0]


def to_dtype(sd: SD, dtype: torch.dtype) -> SD:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
