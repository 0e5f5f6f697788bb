"""Attention rollout on the native path (reference src/v1t/utils/attention_rollout.py).

The reference records every block's softmax output with forward hooks — a (B, blocks, heads, T, T) fp32
tensor, 175 MB per image at the default size (attention_rollout.py:28-36, 76) — then, per sample, takes the
max over heads, adds the identity, row-normalises, multiplies the L matrices together and keeps row 0 of the
product without the CLS column (:92-118), min-max normalises it and resizes it to the image (:119-121).

Here the fused attention never materialises P. After one eval forward that keeps each block's q/k and
log2-sum-exp, the head-max matrix of ONE block at a time is recomputed on the MFMAs
(`v1t_rollout_headmax`, (B, T, T) fp32), and because only row 0 of the product is used, the chain is run as
a row-vector chain from the last block down, `v <- v . (A_k + I) / rowsum` (`v1t_rollout_vecmat`): 3*2*T^2
instead of 3*2*T^3 flops per image, mathematically identical (matrix products are associative; the fp32
re-association difference is ~1e-9).
"""
from __future__ import annotations

import contextlib
import typing as t

import torch
from torch.nn import functional as F

from . import lib as L
from .core import ViTCore, find_shape


def release_rollout_scratch(core: ViTCore) -> None:
    """Drop the (B, T, TP) fp32 buffers a `full_chain` rollout keeps on the core (8.5 GB at batch 256) and hand the memory back
    to the device."""
    if getattr(core, "_rollout_scratch", None) is not None:
        core._rollout_scratch = None
        torch.cuda.empty_cache()


@torch.no_grad()
def attention_probabilities(core: ViTCore, images: torch.Tensor, behaviors: torch.Tensor, pupil_centers: torch.Tensor, mouse_id: str) -> torch.Tensor:
    """Per-head softmax probabilities of every block, (B, blocks, heads, T, T) fp32 - the tensor the reference's Recorder returns
    (attention_rollout.py:31-36, 76) - recomputed block by block from the saved q / k and log-sum-exp (`v1t_attention_probs`).
    175 MB per image at the default size, like the reference's: meant for a handful of images."""
    if not getattr(core, "cls_tokens", 1):
        # (the CCT plan's qkv rows are 3 * H * round_up(D / H^2, 32) wide, not 3 * H * padded_dim: reading them with the ViT geometry would
        # run past the plane)
        raise NotImplementedError("per-head attention probabilities are exposed for the ViT core only; the reference's Recorder does not find "
                                  "cct.py's Attention modules either (attention_rollout.py:31-36)")
    was_training = core.training
    core.train(False)
    try:
        tokens = core.forward_tokens(images, mouse_id, behaviors, pupil_centers, keep_workspace=True)
        core._last_tokens = tokens
    finally:
        core.train(was_training)
    lib = L.load()
    B, T = tokens.shape[0], core.num_tokens
    TP = (T + 3) // 4 * 4
    cfg = core._cfg
    H, DP = cfg.num_heads, core.padded_dim
    out = torch.empty((B, cfg.num_blocks, H, T, T), dtype=torch.float32, device=tokens.device)
    P = torch.empty((B, H, T, TP), dtype=torch.float32, device=tokens.device)
    nqkv, nlse = B * T * 3 * H * DP * 2, B * H * T * 4
    for k in range(cfg.num_blocks):
        qkv = core.workspace_tensor("qkv", k)[:nqkv]
        lse2 = core.workspace_tensor("lse2", k)[:nlse]
        scale = core.transformer.blocks[k]["mha"].scale
        L.check(lib.v1t_attention_probs(qkv.data_ptr(), lse2.data_ptr(), B, H, T, DP, scale.data_ptr(), int(cfg.use_lsa), int(cfg.use_lsa),
                                        P.data_ptr(), TP, L.stream()), "attention_probs")
        out[:, k] = P[..., :T]
    return out


@torch.no_grad()
def rollout_rows(core: ViTCore, images: torch.Tensor, behaviors: torch.Tensor, pupil_centers: torch.Tensor, mouse_id: str,
                 return_headmax: bool = False, full_chain: bool = False):
    """images: CORE input (B, C, H, W) (post image-cropper). Returns the pre-normalisation heat vector
    J_last[0, 1:] (B, T-1) of attention_rollout.py:118 (and optionally the list of head-max matrices).
    `full_chain`: multiply the (T x T) matrices out as the reference does (attention_rollout.py:113-117, `v1t_rollout_matmul`,
    2 T^3 flops per image and block on the MFMAs) instead of the row-vector chain (2 T^2): same result up to fp32
    re-association, ~1000x the arithmetic - there for parity with the reference's algorithm and as a benchmark (config C5)."""
    if not getattr(core, "cls_tokens", 1):
        raise NotImplementedError("attention rollout reads the class token's row (attention_rollout.py:118); the CCT core has no class token "
                                  "(the reference's Recorder does not find cct.py's Attention modules either)")
    if full_chain:
        return _rollout_rows_full(core, images, behaviors, pupil_centers, mouse_id, return_headmax)
    was_training = core.training
    core.train(False)
    try:
        tokens = core.forward_tokens(images, mouse_id, behaviors, pupil_centers, keep_workspace=True)
        core._last_tokens = tokens
    finally:
        core.train(was_training)
    lib = L.load()
    B, T = tokens.shape[0], core.num_tokens
    TP = (T + 31) // 32 * 32  # whole 32-key tiles: a query row's 128-B tile segments are line-aligned (T = 1654 -> 1664 floats = 52 lines per row)
    cfg = core._cfg
    H, DP = cfg.num_heads, core.padded_dim
    dev = tokens.device
    # The head-max maps do not depend on the chain, only the vector-matrix products do: block k - 1's map (MFMA + exp work, writes at ~2 TB/s)
    # is recomputed on a second stream while block k's product streams its 2.8 GB back in (HBM-read-bound) - two map buffers in turn
    # (2 x 2.8 GB at batch 256 of the 288 GB), events both ways. `return_headmax` keeps the simple serial form (it clones every map).
    pipelined = (not return_headmax) and cfg.num_blocks > 1 and tokens.is_cuda
    nbuf = 2 if pipelined else 1
    As = [torch.empty((B, T, TP), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    rss = [torch.empty((B, T), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    v: t.Optional[torch.Tensor] = None
    maps = []
    nqkv, nlse = B * T * 3 * H * DP * 2, B * H * T * 4
    main = torch.cuda.current_stream() if pipelined else None
    side = core._rollout_stream if (pipelined and getattr(core, "_rollout_stream", None) is not None) else (torch.cuda.Stream() if pipelined else None)
    if pipelined:
        core._rollout_stream = side
        side.wait_stream(main)  # the forward that produced q / k / lse
    used = [None] * nbuf   # event: the product that last read this buffer has finished
    for n_, k in enumerate(reversed(range(cfg.num_blocks))):
        A, rowsum = As[n_ % nbuf], rss[n_ % nbuf]
        qkv = core.workspace_tensor("qkv", k)[:nqkv]
        lse2 = core.workspace_tensor("lse2", k)[:nlse]
        scale = core.transformer.blocks[k]["mha"].scale
        with (torch.cuda.stream(side) if pipelined else contextlib.nullcontext()):
            if pipelined and used[n_ % nbuf] is not None:
                side.wait_event(used[n_ % nbuf])
            if v is None and n_ == 0 and not return_headmax:
                # the chain starts from e_0: of the LAST block's matrix only row 0 (and its row sum) is read - one 128-query workgroup per image
                # instead of 13 (a quarter of the rollout's head-max work at 4 blocks)
                L.check(lib.v1t_rollout_headmax_rows(qkv.data_ptr(), lse2.data_ptr(), B, H, T, DP, scale.data_ptr(), int(cfg.use_lsa), int(cfg.use_lsa),
                                                     A.data_ptr(), TP, rowsum.data_ptr(), 1, L.stream()), "rollout_headmax")
            else:
                L.check(lib.v1t_rollout_headmax(qkv.data_ptr(), lse2.data_ptr(), B, H, T, DP, scale.data_ptr(), int(cfg.use_lsa), int(cfg.use_lsa),
                                                A.data_ptr(), TP, rowsum.data_ptr(), L.stream()), "rollout_headmax")
            if pipelined:
                ready = torch.cuda.Event()
                ready.record(side)
        if pipelined:
            main.wait_event(ready)
        u = torch.empty((B, T), dtype=torch.float32, device=dev)
        L.check(lib.v1t_rollout_vecmat(A.data_ptr(), rowsum.data_ptr(), L.ptr(v), u.data_ptr(), B, T, TP, L.stream()), "rollout_vecmat")
        if pipelined:
            used[n_ % nbuf] = torch.cuda.Event()
            used[n_ % nbuf].record(main)
        v = u
        if return_headmax:
            maps.append(A[:, :, :T].clone())
    # (no record_stream for the buffers written on the side stream: the current stream has waited for every head-max launch - `ready` - so
    # by stream order it is past all of the side stream's work when these tensors are freed; record_stream would make the caching allocator
    # hold the 2 x 2.8 GB back until it has polled the side stream's events and map fresh segments meanwhile: 137-200 ms per call, measured)
    rows = v[:, 1:]
    return (rows, maps[::-1]) if return_headmax else rows


def _rollout_rows_full(core: ViTCore, images, behaviors, pupil_centers, mouse_id: str, return_headmax: bool):
    was_training = core.training
    core.train(False)
    try:
        tokens = core.forward_tokens(images, mouse_id, behaviors, pupil_centers, keep_workspace=True)
        core._last_tokens = tokens
    finally:
        core.train(was_training)
    lib = L.load()
    B, T = tokens.shape[0], core.num_tokens
    TP = (T + 3) // 4 * 4
    cfg = core._cfg
    H, DP = cfg.num_heads, core.padded_dim
    dev = tokens.device
    # A and the two ping-pong result^T buffers are (B, T, TP) fp32 each - 2.8 GB at batch 256: kept on the core while a caller
    # loops over batches (a fresh 8.5 GB request per call makes the caching allocator release and re-map segments, 10x the time
    # of the chain itself); `attention_rollouts(..., keep_scratch=False)` (the default), `release_rollout_scratch(core)` or
    # `Recorder.clear()` drop them
    key = (B, T, TP, str(dev))
    scratch = getattr(core, "_rollout_scratch", None)
    if scratch is None or scratch[0] != key:
        scratch = (key, torch.empty((B, T, TP), dtype=torch.float32, device=dev), torch.empty((B, T), dtype=torch.float32, device=dev),
                   [torch.empty((B, T, TP), dtype=torch.float32, device=dev) for _ in range(2)])
        core._rollout_scratch = scratch
    _, A, rowsum, X = scratch
    cur: t.Optional[torch.Tensor] = None
    maps = []
    nqkv, nlse = B * T * 3 * H * DP * 2, B * H * T * 4
    for k in range(cfg.num_blocks):  # the reference's order: result = a_k @ result, first block first
        qkv = core.workspace_tensor("qkv", k)[:nqkv]
        lse2 = core.workspace_tensor("lse2", k)[:nlse]
        scale = core.transformer.blocks[k]["mha"].scale
        L.check(lib.v1t_rollout_headmax(qkv.data_ptr(), lse2.data_ptr(), B, H, T, DP, scale.data_ptr(), int(cfg.use_lsa), int(cfg.use_lsa),
                                        A.data_ptr(), TP, rowsum.data_ptr(), L.stream()), "rollout_headmax")
        out = X[k & 1]
        L.check(lib.v1t_rollout_matmul(A.data_ptr(), rowsum.data_ptr(), L.ptr(cur), out.data_ptr(), B, T, TP, L.stream()), "rollout_matmul")
        cur = out
        if return_headmax:
            maps.append(A[:, :, :T].clone())
    rows = cur[:, 1:T, 0].contiguous()  # result[0, 1:] = column 0 of result^T
    return (rows, maps) if return_headmax else rows


class Recorder(torch.nn.Module):
    """Counterpart of the reference `Recorder` (attention_rollout.py:15-77) for the native core. The reference hooks every
    block's `Attention.attend` (nn.Softmax) module; the fused attention has no such module - P is never materialised - so the
    hook surface cannot exist. This class offers the same constructor / `forward` / `eject` / `clear` and returns what the
    reference's own `attention_rollouts(attentions, image_shape)` consumes: `attentions` of shape (B, blocks, 1, T, T) holding
    the MAX OVER HEADS of the softmax probabilities (recomputed from the saved q/k and log-sum-exp by `v1t_rollout_headmax`).
    The reference's first step is exactly that maximum (`torch.max(attention, dim=1)`, :105), which over a singleton head axis
    is the identity, so its heat-maps come out the same while the tensor is `heads` times smaller (43.8 MB instead of 175 MB
    per image at the default size). `Recorder(core, per_head=True)` returns the reference's own tensor instead, per-head
    probabilities (B, blocks, heads, T, T) (`attention_probabilities`), at the reference's memory cost."""

    def __init__(self, core: ViTCore, per_head: bool = False):
        super().__init__()
        self.core = core
        self.per_head = per_head
        self.ejected = False

    def eject(self):
        self.ejected = True
        return self.core

    def clear(self):
        release_rollout_scratch(self.core)

    @torch.no_grad()
    def forward(self, images: torch.Tensor, behaviors: torch.Tensor, pupil_centers: torch.Tensor, mouse_id: str):
        assert not self.ejected, "recorder has been ejected, cannot be used anymore"
        core = self.core
        if self.per_head:
            probs = attention_probabilities(core, images, behaviors, pupil_centers, mouse_id)
            return core.tokens_to_output(core._last_tokens), probs
        _, maps = rollout_rows(core, images, behaviors, pupil_centers, mouse_id, return_headmax=True)
        tokens = core._last_tokens
        return core.tokens_to_output(tokens), torch.stack(maps, dim=1)[:, :, None]


@torch.no_grad()
def attention_rollouts(core: ViTCore, images: torch.Tensor, behaviors: torch.Tensor, pupil_centers: torch.Tensor, mouse_id: str,
                       full_chain: bool = False, keep_scratch: bool = False) -> torch.Tensor:
    """Heat-maps (B, H, W) like reference `attention_rollouts` (attention_rollout.py:125-133) applied to the
    recorder output of `core` on `images`. `keep_scratch`: a `full_chain` rollout leaves its (B, T, TP) buffers on the core for
    the next call (loops over many batches, bench.py); by default they are released before returning."""
    rows = rollout_rows(core, images, behaviors, pupil_centers, mouse_id, full_chain=full_chain)
    if full_chain and not keep_scratch:
        rows = rows.clone()
        release_rollout_scratch(core)
    B = rows.shape[0]
    h, w = find_shape(rows.shape[1])
    heat = rows.reshape(B, h, w)
    lo = heat.amin(dim=(1, 2), keepdim=True)
    hi = heat.amax(dim=(1, 2), keepdim=True)
    heat = (heat - lo) / (hi - lo)
    return F.interpolate(heat[:, None], size=tuple(images.shape[2:]), mode="bilinear", align_corners=False, antialias=False)[:, 0]


@torch.no_grad()
def extract_attention_maps(model, batches: t.Iterable[t.Dict[str, torch.Tensor]], mouse_id: str, num_samples: t.Optional[int] = None) -> t.Dict[str, torch.Tensor]:
    """Counterpart of reference `extract_attention_maps` (attention_rollout.py:136-201) over an iterable of
    batches (image, behavior, pupil_center) that are already on the model's device; the dataset's inverse
    transforms of the reference are the caller's business here."""
    model.train(False)
    out = {"images": [], "heatmaps": [], "pupil_centers": [], "behaviors": []}
    count = 0
    for b in batches:
        images, _ = model.image_cropper(b["image"], mouse_id=mouse_id, behaviors=b["behavior"], pupil_centers=b["pupil_center"])
        heat = attention_rollouts(model.core, images, b["behavior"], b["pupil_center"], mouse_id)
        out["images"].append(images)
        out["heatmaps"].append(heat)
        out["behaviors"].append(b["behavior"])
        out["pupil_centers"].append(b["pupil_center"])
        count += images.shape[0]
        if num_samples is not None and count >= num_samples:
            break
    res = {k: torch.cat(v) for k, v in out.items()}
    if num_samples is not None:
        res = {k: v[:num_samples] for k, v in res.items()}
    return res
