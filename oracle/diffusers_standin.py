"""TEST INFRASTRUCTURE ONLY — never imported by the product path.

A stand-in for the diffusers ``Attention`` module (diffusers is not in this image): just the attributes and helper
methods the reference's processors touch (``/root/reference/src/models_ipa/attention_processor.py:19-79, 203-280``), written from
the published diffusers semantics: ``to_q/to_k/to_v`` without bias, ``to_out = [Linear(bias), Dropout]``, ``scale = dim_head ** -0.5``,
``head_to_batch_dim`` [B, L, H d] -> [B H, L, d], ``get_attention_scores`` = softmax(scale * q k^T) via ``baddbmm`` in the tensor dtype,
optional ``group_norm`` over channels, ``residual_connection`` and ``rescale_output_factor``.  It is the object the REAL reference
processor is run on by ``oracle/make_golden_attnproc.py`` and the object the product shim is handed in ``tests/test_attn_processor.py``.
"""
import torch
from torch import nn


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, norm_num_groups=None, residual_connection=False,
                 rescale_output_factor=1.0):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.spatial_norm = None
        self.norm_cross = None
        self.residual_connection = residual_connection
        self.rescale_output_factor = rescale_output_factor
        self.group_norm = nn.GroupNorm(num_channels=query_dim, num_groups=norm_num_groups, eps=1e-5, affine=True) if norm_num_groups else None
        cd = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cd, inner, bias=False)
        self.to_v = nn.Linear(cd, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])

    def prepare_attention_mask(self, attention_mask, target_length, batch_size):
        assert attention_mask is None
        return None

    def head_to_batch_dim(self, t):
        b, l, e = t.shape
        return t.reshape(b, l, self.heads, e // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, l, e // self.heads)

    def batch_to_head_dim(self, t):
        bh, l, d = t.shape
        b = bh // self.heads
        return t.reshape(b, self.heads, l, d).permute(0, 2, 1, 3).reshape(b, l, self.heads * d)

    def get_attention_scores(self, query, key, attention_mask=None):
        dtype = query.dtype
        empty = torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype, device=query.device)
        scores = torch.baddbmm(empty, query, key.transpose(-1, -2), beta=0, alpha=self.scale)
        return scores.softmax(dim=-1).to(dtype)


CASES = {
    # name: (constructor kwargs, hidden_states shape, encoder_hidden_states shape or None)
    "self_1d": (dict(query_dim=128, heads=4, dim_head=32), (2, 64, 128), None),
    "cross_1d": (dict(query_dim=128, cross_attention_dim=96, heads=4, dim_head=32), (2, 64, 128), (2, 16, 96)),
    "sdxl_self_hd64": (dict(query_dim=640, heads=10, dim_head=64), (2, 256, 640), None),
    "sdxl_cross_hd64": (dict(query_dim=640, cross_attention_dim=2048, heads=10, dim_head=64), (2, 256, 640), (2, 64, 2048)),
    "vae_4d_groupnorm_residual": (dict(query_dim=64, heads=1, dim_head=64, norm_num_groups=8, residual_connection=True), (1, 64, 8, 8), None),
}


def build(name, seed=0):
    kw, xs, es = CASES[name]
    g = torch.Generator().manual_seed(1000 + seed + sum(map(ord, name)))
    m = Attention(**kw)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.5 if p.ndim == 1 else p.shape[-1] ** -0.5))
        if m.group_norm is not None:
            m.group_norm.weight.add_(1.0)
    x = torch.randn(xs, generator=g)
    e = torch.randn(es, generator=g) if es else None
    return m.eval(), x, e
