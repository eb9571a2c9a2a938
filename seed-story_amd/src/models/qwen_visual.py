"""MI355X-native drop-in for the reference's ``src/models/qwen_visual.py`` hot-path classes:

* ``Resampler``  (reference qwen_visual.py:95-153)  — learnable-query cross-attention; the
  agent's input_resampler / output_resampler ("image-feature regressor") and the ViT attn_pool;
* ``VisionTransformerWithAttnPool``  (reference :321-422)  — Qwen ViT-G + attention pool.

Same constructor arguments, same parameter names/shapes (so the reference checkpoints load with
``load_state_dict``), same call protocol (``module(x) -> tensor``).  The modules are parameter
containers: ``forward`` enqueues the hand-written HIP kernels of ``libseedstory_hip.so`` through
one native call per module (``ss_resampler_forward`` / ``ss_vit_forward``); nothing is computed by
torch on the device and there is no CPU path.
"""
import ctypes as C
import math

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from seedstory import _lib, ops, tune
from seedstory._lib import check, lib


def get_abs_pos(abs_pos, tgt_size):
    """Bicubic resize of a square position table [L, C] -> [tgt, C] (reference :23-39).
    Weight preprocessing: done once per (table, length) on the host in fp32 and cached."""
    src = int(math.sqrt(abs_pos.size(0)))
    tgt = int(math.sqrt(tgt_size))
    if src == tgt:
        return abs_pos
    x = abs_pos.detach().float().cpu().reshape(1, src, src, -1).permute(0, 3, 1, 2)
    x = F.interpolate(x, size=(tgt, tgt), mode="bicubic", align_corners=False)
    return x.permute(0, 2, 3, 1).flatten(0, 2).to(device=abs_pos.device, dtype=abs_pos.dtype)


def get_2d_sincos_pos_embed(embed_dim, grid_size):
    """Fixed 2-D sin-cos table (reference :45-92): first half of the channels encodes the
    w coordinate, second half the h coordinate, each as [sin | cos]."""
    coords = np.arange(grid_size, dtype=np.float32)
    ww, hh = np.meshgrid(coords, coords)  # w varies fastest

    def axis(dim, pos):
        omega = np.arange(dim // 2, dtype=np.float32)
        omega /= dim / 2.0
        omega = 1.0 / 10000 ** omega
        ang = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)

    return np.concatenate([axis(embed_dim // 2, ww), axis(embed_dim // 2, hh)], axis=1)


class _Lin(nn.Module):
    """nn.Linear-shaped parameter holder (uninitialised storage: checkpoints or ``init_synthetic``
    fill it; a 1.9 B-parameter CPU random init would take longer than the whole story)."""

    def __init__(self, i, o, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(o), requires_grad=False) if bias else None


class _LN(nn.Module):
    def __init__(self, d, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(d), requires_grad=False)
        self.eps = eps


class _MHA(nn.Module):
    """Parameter names of nn.MultiheadAttention(embed_dim, num_heads) (reference :123)."""

    def __init__(self, e):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * e, e), requires_grad=False)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * e), requires_grad=False)
        self.out_proj = _Lin(e, e)


def _param_sig(module):
    return tuple((p.data_ptr(), p._version, p.dtype, str(p.device)) for p in module.parameters())


def _fill_(module, seed, std=0.02):
    """Deterministic synthetic weights on the module's current device (benchmark / smoke)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    for name, p in module.named_parameters():
        if name.endswith("pos_embed"):
            continue
        if p.dim() == 1 and ("ln" in name or "norm" in name) and name.endswith("weight"):
            p.data.fill_(1.0)
        elif p.dim() == 1:
            p.data.zero_()
        elif p.is_cuda:
            p.data.normal_(0.0, std)
        else:
            p.data.copy_(torch.randn(p.shape, generator=g) * std)


class Resampler(nn.Module):
    """Reference ``Resampler(grid_size, embed_dim, num_heads, kv_dim=None, norm_layer=nn.LayerNorm)``."""

    def __init__(self, grid_size, embed_dim, num_heads, kv_dim=None, norm_layer=None, ln_eps=1e-5):
        super().__init__()
        self.num_queries = grid_size ** 2
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        if norm_layer is not None:  # the ViT passes partial(nn.LayerNorm, eps=1e-6) (reference :353)
            ln_eps = getattr(norm_layer, "keywords", {}).get("eps", ln_eps)
        self.pos_embed = nn.Parameter(torch.from_numpy(get_2d_sincos_pos_embed(embed_dim, grid_size)).float(),
                                      requires_grad=False)
        self.query = nn.Parameter(torch.zeros(self.num_queries, embed_dim), requires_grad=False)
        if kv_dim is not None and kv_dim != embed_dim:
            self.kv_proj = _Lin(kv_dim, embed_dim, bias=False)
            self.out_dim = kv_dim
            self.kv_dim = kv_dim
        else:
            self.kv_proj = nn.Identity()
            self.out_dim = embed_dim
            self.kv_dim = embed_dim
        self.attn = _MHA(embed_dim)
        self.ln_q = _LN(embed_dim, ln_eps)
        self.ln_kv = _LN(embed_dim, ln_eps)
        self._cache = {}

    def init_synthetic(self, seed=0):
        _fill_(self, seed)
        return self

    def _weights(self, l_kv):
        key = (l_kv, _param_sig(self))
        hit = self._cache.get("w")
        if hit is not None and hit[0] == key:
            return hit[1]
        q_in = ops.layernorm(self.query.data, self.ln_q.weight.data, self.ln_q.bias.data, self.ln_q.eps)
        q_in = ops.add_bcast(q_in.unsqueeze(0), self.pos_embed.data).squeeze(0).contiguous()  # ln_q(query)+pos (:147)
        pos_kv = get_abs_pos(self.pos_embed.data, l_kv).contiguous()
        kvp = self.kv_proj.weight.data if isinstance(self.kv_proj, _Lin) else None
        keep = (q_in, pos_kv)
        w = _lib.ResamplerWeights(q_in.data_ptr(), pos_kv.data_ptr(), ops.p(kvp), self.ln_kv.weight.data_ptr(),
                                  self.ln_kv.bias.data_ptr(), self.attn.in_proj_weight.data_ptr(),
                                  self.attn.in_proj_bias.data_ptr(), self.attn.out_proj.weight.data_ptr(),
                                  self.attn.out_proj.bias.data_ptr(), self.num_queries, self.embed_dim,
                                  self.num_heads, self.kv_dim, l_kv, self.ln_kv.eps)
        self._cache["w"] = (key, (w, keep))
        return w, keep

    def forward(self, x, attn_mask=None):
        assert attn_mask is None, "attn_mask is never used on the hot path"
        x = x.to(dtype=self.query.dtype).contiguous()
        B, L, _ = x.shape
        w, _keep = self._weights(L)
        y = torch.empty(B, self.num_queries, self.embed_dim, dtype=x.dtype, device=x.device)
        code = ops.dt(x)
        E = self.embed_dim
        for m, n, k in ((B * L, E, self.kv_dim), (self.num_queries, E, E), (B * L, E, E), (B * self.num_queries, E, E)):
            tune.ensure_gemm(m, n, k, code, 0, x.device)           # tile table entries of this module's projections
        nbytes = lib().ss_resampler_workspace_bytes(C.byref(w), B, code)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        check(lib().ss_resampler_forward(C.byref(w), x.data_ptr(), y.data_ptr(), B, ws.data_ptr(), nbytes, code,
                                         ops.stream()), "ss_resampler_forward")
        return y


class _VisualAttention(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.in_proj = _Lin(width, 3 * width)
        self.out_proj = _Lin(width, width)


class _Block(nn.Module):
    def __init__(self, width, mlp_width, eps):
        super().__init__()
        self.ln_1 = _LN(width, eps)
        self.ln_2 = _LN(width, eps)
        self.attn = _VisualAttention(width)
        self.mlp = nn.ModuleDict({"c_fc": _Lin(width, mlp_width), "c_proj": _Lin(mlp_width, width)})


class _Transformer(nn.Module):
    def __init__(self, width, layers, mlp_width, eps):
        super().__init__()
        self.resblocks = nn.ModuleList([_Block(width, mlp_width, eps) for _ in range(layers)])

    def get_cast_dtype(self):
        return self.resblocks[0].mlp["c_fc"].weight.dtype

    def get_cast_device(self):
        return self.resblocks[0].mlp["c_fc"].weight.device


class VisionTransformerWithAttnPool(nn.Module):
    """Reference ``VisionTransformerWithAttnPool(image_size, patch_size, width, layers, heads,
    mlp_ratio, n_queries=256, output_dim=512, **kwargs)`` (:321-374); forward :376-399."""

    def __init__(self, image_size, patch_size, width, layers, heads, mlp_ratio, n_queries=256, output_dim=512,
                 **kwargs):
        super().__init__()
        self.image_size = (image_size, image_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (image_size // patch_size, image_size // patch_size)
        self.output_dim = output_dim
        self.width, self.heads, self.n_layers = width, heads, layers
        self.mlp_width = int(width * mlp_ratio)
        eps = 1e-6
        self.conv1 = nn.Module()
        self.conv1.weight = nn.Parameter(torch.empty(width, 3, patch_size, patch_size), requires_grad=False)
        self.positional_embedding = nn.Parameter(torch.empty(256, width), requires_grad=False)
        self.ln_pre = _LN(width, eps)
        self.transformer = _Transformer(width, layers, self.mlp_width, eps)
        self.attn_pool = Resampler(grid_size=int(math.sqrt(n_queries)), embed_dim=output_dim,
                                   num_heads=output_dim // 128, kv_dim=width, ln_eps=eps)
        self.ln_post = _LN(output_dim, eps)
        self.proj = nn.Parameter(torch.empty(output_dim, output_dim), requires_grad=False)
        self._cache = {}

    def init_synthetic(self, seed=0):
        _fill_(self, seed)
        self.positional_embedding.data.mul_(self.width ** -0.5 / 0.02)
        self.proj.data.mul_(self.output_dim ** -0.5 / 0.02)
        return self

    def _weights(self):
        key = _param_sig(self)
        hit = self._cache.get("w")
        if hit is not None and hit[0] == key:
            return hit[1]
        P = self.patch_size[0]
        tokens = self.grid_size[0] * self.grid_size[1]
        kraw = 3 * P * P
        kpad = (kraw + 63) // 64 * 64
        cw = self.conv1.weight.data
        conv_w = torch.zeros(self.width, kpad, dtype=cw.dtype, device=cw.device)
        conv_w[:, :kraw] = cw.reshape(self.width, kraw)
        pos = get_abs_pos(self.positional_embedding.data, tokens).contiguous()
        proj_t = self.proj.data.t().contiguous()  # x @ proj  ==  Linear with weight proj^T (:397)
        arr = (_lib.VitLayerWeights * self.n_layers)()
        for i, blk in enumerate(self.transformer.resblocks):
            arr[i] = _lib.VitLayerWeights(*[t.data_ptr() for t in (
                blk.ln_1.weight, blk.ln_1.bias, blk.ln_2.weight, blk.ln_2.bias, blk.attn.in_proj.weight,
                blk.attn.in_proj.bias, blk.attn.out_proj.weight, blk.attn.out_proj.bias, blk.mlp["c_fc"].weight,
                blk.mlp["c_fc"].bias, blk.mlp["c_proj"].weight, blk.mlp["c_proj"].bias)])
        w = _lib.VitWeights(conv_w.data_ptr(), pos.data_ptr(), self.ln_pre.weight.data_ptr(),
                            self.ln_pre.bias.data_ptr(), arr, self.width, self.n_layers, self.heads, self.mlp_width,
                            P, self.image_size[0], kpad, self.ln_pre.eps)
        keep = (conv_w, pos, proj_t, arr)
        self._cache["w"] = (key, (w, keep))
        return w, keep

    def forward(self, x):
        x = x.to(dtype=self.transformer.get_cast_dtype(), device=self.transformer.get_cast_device()).contiguous()
        B = x.shape[0]
        w, keep = self._weights()
        tokens = self.grid_size[0] * self.grid_size[1]
        feat = torch.empty(B, tokens, self.width, dtype=x.dtype, device=x.device)
        code = ops.dt(x)
        rows, Wd = B * tokens, self.width
        for n, k, epi in ((Wd, w.kpad, 0), (3 * Wd, Wd, 0), (Wd, Wd, 0), (self.mlp_width, Wd, _lib.EPI_GELU), (Wd, self.mlp_width, 0)):
            tune.ensure_gemm(rows, n, k, code, epi, x.device)
        nbytes = lib().ss_vit_workspace_bytes(C.byref(w), B, code)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        check(lib().ss_vit_forward(C.byref(w), x.data_ptr(), feat.data_ptr(), B, ws.data_ptr(), nbytes, code,
                                   ops.stream()), "ss_vit_forward")
        y = self.attn_pool(feat)                                                    # :394
        y = ops.layernorm(y, self.ln_post.weight.data, self.ln_post.bias.data, self.ln_post.eps)  # :395
        y = ops.gemm(y.view(-1, self.output_dim), keep[2]).view(B, -1, self.output_dim)             # :397
        return y

    @classmethod
    def from_pretrained(cls, pretrained_model_path=None, **kwargs):
        model = cls(**kwargs)
        if pretrained_model_path is not None:
            from seedstory import ckpt as _ckpt
            print("Load ckpt of qwen visual encoder")
            _ckpt.load_checked(model, _ckpt.read_weights(pretrained_model_path), "qwen visual,")
        return model
