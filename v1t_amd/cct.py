"""MI355X-native "cct" core: host-side mirror of the reference's CCTCore (src/v1t/models/core/cct.py:247-317) on the same
HIP plan as the ViT core (`v1t_vit_config.core_kind = 1`, include/v1t_amd.h).

What the reference's Compact Convolutional Transformer is (and what the plan therefore implements):
  * tokenizer (cct.py:30-104): Conv2d(C -> emb_dim, kernel patch_size, stride patch_stride, padding 3, NO bias, kaiming-normal
    init) -> ReLU -> MaxPool2d(3, 2, 1) -> "b c h w -> b (h w) c" -> + position table (`--pos_emb sine`: a fixed buffer,
    cct.py:17-27; `none`; `learn` raises AttributeError in the reference's constructor - cct.py:65 initialises `self.pos_emb`,
    which does not exist - and NotImplementedError here) -> Dropout(p_dropout). No class token.
  * attention (cct.py:107-143): qkv = Linear(emb_dim, 3 * inner, no bias) with inner = emb_dim // heads, then cut into `heads`
    heads of inner // heads columns; scores scaled by the buffer inner^-0.5; projection Linear(inner, emb_dim).
  * block (cct.py:146-198): x += BehaviorMLP(behaviors) (modes 3 / 4; the class is vit.py's, cct.py:14), x = drop_path(mha(x)) + x,
    x = drop_path(mlp(x)) + x with the block's own DropPath rate linspace(0, drop_path, blocks)[k] (cct.py:219-233).
  * output: every token, "b (h w) c -> b c h w" with (h, w) = find_shape(tokens) (cct.py:290-299).
Same registry name ("cct"), constructor, attributes, state-dict keys and error behaviour; compute = `v1t_vit_forward / _backward`.
"""
from __future__ import annotations

import math
import typing as t

import numpy as np
import torch
from torch import nn

from . import lib as L
from .core import Core, ViTCore, _ParamBag, _seq, register


def sinusoidal_embedding(num_tokens: int, dim: int) -> torch.Tensor:
    """(1, tokens, dim) fixed table: angle(p, i) = p / 10000^(2 (i // 2) / dim), sine on even columns, cosine on odd ones - the
    values of cct.py:17-27 (float32 arithmetic on the Python-float angles, like torch.FloatTensor of the nested list)."""
    p = np.arange(num_tokens, dtype=np.float64)[:, None]
    i = np.arange(dim, dtype=np.float64)[None, :]
    pe = torch.from_numpy(p / (10000.0 ** (2.0 * np.floor(i / 2.0) / dim))).to(torch.float32)
    pe[:, 0::2] = torch.sin(pe[:, 0::2])
    pe[:, 1::2] = torch.cos(pe[:, 1::2])
    return pe[None]


@register("cct")
class CCTCore(ViTCore):
    CONV_PADDING = 3  # Tokenizer's default (cct.py:37); CCTCore never passes another one

    def __init__(self, args, input_shape: t.Tuple[int, int, int], name: str = "CCTCore"):
        Core.__init__(self, args, input_shape=input_shape, name=name)  # asserts behavior_mode != 2 for non-ViT cores (core.py:27-28)
        self.register_buffer("reg_scale", torch.tensor(float(args.core_reg_scale)))
        if not hasattr(args, "patch_stride"):
            print("patch_stride is not defined, set to 1.")
            args.patch_stride = 1
        if not hasattr(args, "grad_checkpointing"):
            args.grad_checkpointing = False  # nothing is materialised that would need it (as in the ViT core)
        pos = getattr(args, "pos_emb", "sine")
        assert pos in ("sine", "learn", "none")
        if pos == "learn":
            raise NotImplementedError("--pos_emb learn: the reference's Tokenizer raises in its constructor for it (cct.py:64-65, `self.pos_emb`)")
        if self.behavior_mode == 1:
            raise AssertionError("behavior_mode 1 builds a BehaviorMLP in every block, whose constructor asserts mode in (2, 3, 4) (cct.py:171-174, vit.py:175)")
        inner = args.emb_dim // args.num_heads
        assert inner % args.num_heads == 0, f"MHA inner_dim ({inner}) must be divisible by num_heads ({args.num_heads})"
        self.drop_path_rate = float(getattr(args, "drop_path", 0.0))
        assert 0.0 <= self.drop_path_rate < 1.0
        rates64 = np.linspace(0, self.drop_path_rate, args.num_blocks)  # float64, as cct.py:209 keeps them
        self.drop_path_rates = torch.from_numpy(rates64).to(torch.float32)
        self._drop_path_rates64 = rates64
        self.pos_emb = pos
        c, h, w = input_shape
        self.mouse_ids = list(args.output_shapes.keys())
        cfg = L.VitConfig(
            in_channels=c, in_h=h, in_w=w, patch_size=args.patch_size, patch_stride=args.patch_stride, patch_mode=0,
            emb_dim=args.emb_dim, num_heads=args.num_heads, mlp_dim=int(args.mlp_dim), num_blocks=args.num_blocks,
            behavior_mode=self.behavior_mode if self.behavior_mode in (3, 4) else 0, num_mice=len(self.mouse_ids), use_lsa=0, use_bias=1,
            p_dropout=float(args.p_dropout), t_dropout=float(args.t_dropout), ln_eps=1e-5,
            core_kind=1, conv_pad=self.CONV_PADDING, pos_mode=1 if pos == "sine" else 0,
        )
        self._finish_init(_Args(args), cfg, c)

    def _build_modules(self, args, c: int) -> None:
        D, H, M, P = args.emb_dim, args.num_heads, int(args.mlp_dim), args.patch_size
        inner = D // H
        tok = _ParamBag()
        tok.conv2d = nn.Conv2d(c, D, kernel_size=P, stride=args.patch_stride, padding=self.CONV_PADDING, bias=False)
        nn.init.kaiming_normal_(tok.conv2d.weight)  # Tokenizer.init_weight (cct.py:82-85)
        if self.pos_emb == "sine":
            tok.register_buffer("pos_embedding", sinusoidal_embedding(self.num_tokens, D))
        else:
            tok.pos_embedding = None
        self.tokenizer = tok

        def lin(i, o, b=True):
            m = nn.Linear(i, o, bias=b)
            nn.init.trunc_normal_(m.weight, std=0.02)  # Transformer.init_weight (cct.py:236-244) reaches every Linear below
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
            return m

        tr = _ParamBag()
        tr.blocks = nn.ModuleList()
        for k in range(args.num_blocks):
            mha = _ParamBag()
            mha.register_buffer("scale", torch.tensor(inner ** -0.5))
            mha.layer_norm = nn.LayerNorm(D)
            mha.qkv = lin(D, 3 * inner, False)
            mha.projection = _seq(lin(inner, D), nn.Identity())
            dp = _ParamBag()
            dp.register_buffer("keep_prop", torch.tensor(1 - self._drop_path_rates64[k]))  # float64 buffer: torch.tensor(1 - np.float64), as the reference's DropPath
            block = nn.ModuleDict({"mha": mha, "mlp": _seq(nn.LayerNorm(D), lin(D, M), nn.Identity(), nn.Identity(), lin(M, D), nn.Identity()), "drop_path": dp})
            if self.behavior_mode in (3, 4):
                bm = _ParamBag()
                ids = self.mouse_ids if self.behavior_mode == 4 else ["share"]
                bm.models = nn.ModuleDict({m: _seq(lin(5, D // 2), nn.Identity(), nn.Identity(), lin(D // 2, D), nn.Identity()) for m in ids})
                block["b_mlp"] = bm
            tr.blocks.append(block)
        self.transformer = tr

    @staticmethod
    def find_shape(num_patches: int):
        """reference cct.py:293-299"""
        dim1 = math.ceil(math.sqrt(num_patches))
        while num_patches % dim1 != 0 and dim1 > 0:
            dim1 -= 1
        return dim1, num_patches // dim1


class _Args:
    """`args` as the shared initialisation reads it: the CCT command line has no --use_lsa / --disable_bias / --patch_mode."""

    def __init__(self, args):
        self._a = args

    def __getattr__(self, k):
        if k in ("use_lsa", "disable_bias"):
            return getattr(self._a, k, False)
        if k == "patch_mode":
            return getattr(self._a, k, 0)
        return getattr(self._a, k)
