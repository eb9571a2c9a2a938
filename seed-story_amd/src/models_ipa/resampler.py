"""MI355X-native drop-in for ``ResamplerXLV2`` of the reference's ``src/models_ipa/resampler.py``
(:228-284 with PerceiverAttention :31-76, FeedForward :10-17, AttentionPool2d :79-118): the
4-layer Perceiver that turns a 256x4096 ViT-space feature into the SDXL conditioning
(cross-attention context [B,64,2048] + pooled vector [B,1280]).

Same constructor, same parameter names (``latents``, ``proj_in``, ``layers.i.0.{norm1,norm2,to_q,
to_kv,to_out}``, ``layers.i.1.{0,1,3}``, ``norm_out``, ``unet_proj_1/2``, ``unet_attnpool.*``) so the
de-tokenizer checkpoint loads; ``forward`` composes the HIP kernels of libseedstory_hip.so (LayerNorm,
MFMA GEMM, fused attention, GELU epilogue); no torch compute, no CPU path.
"""
import math

import torch
from torch import nn

from seedstory import ops


class _Lin(nn.Module):
    def __init__(self, i, o, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(o), requires_grad=False) if bias else None


class _LN(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(d), requires_grad=False)
        self.eps = 1e-5


class _PerceiverAttention(nn.Module):
    def __init__(self, dim, dim_head, heads):
        super().__init__()
        inner = dim_head * heads
        self.dim_head, self.heads = dim_head, heads
        self.norm1, self.norm2 = _LN(dim), _LN(dim)
        self.to_q = _Lin(dim, inner, bias=False)
        self.to_kv = _Lin(dim, inner * 2, bias=False)
        self.to_out = _Lin(inner, dim, bias=False)


class _AttentionPool2d(nn.Module):
    def __init__(self, seq_len, embed_dim, num_heads, output_dim):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.empty(seq_len + 1, embed_dim), requires_grad=False)
        self.k_proj, self.q_proj, self.v_proj = _Lin(embed_dim, embed_dim), _Lin(embed_dim, embed_dim), _Lin(embed_dim, embed_dim)
        self.c_proj = _Lin(embed_dim, output_dim)
        self.num_heads = num_heads


def _ln(x, m):
    return ops.layernorm(x, m.weight.data, m.bias.data, m.eps)


class ResamplerXLV2(nn.Module):

    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output1_dim=768,
                 output2_dim=1280, ff_mult=4):
        super().__init__()
        self.latents = nn.Parameter(torch.empty(1, num_queries, dim), requires_grad=False)
        self.proj_in = _Lin(embedding_dim, dim)
        self.norm_out = _LN(dim)
        self.in_dim = dim
        self.out_dim = output1_dim + output2_dim
        self.heads, self.dim_head = heads, dim_head
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            ff = nn.ModuleList([_LN(dim), _Lin(dim, dim * ff_mult, bias=False), nn.Identity(),
                                _Lin(dim * ff_mult, dim, bias=False)])
            self.layers.append(nn.ModuleList([_PerceiverAttention(dim, dim_head, heads), ff]))
        self.unet_proj_1 = _Lin(dim, output1_dim)
        self.unet_proj_2 = _Lin(dim, output2_dim)
        self.unet_attnpool = _AttentionPool2d(num_queries, dim, heads, output2_dim)

    def init_synthetic(self, seed=0):
        for name, p in self.named_parameters():
            if p.dim() == 1:
                p.data.fill_(1.0 if name.endswith("weight") else 0.0)
            else:
                p.data.normal_(0.0, 0.03)
        return self

    @torch.no_grad()
    def forward(self, x, pooled_text_embeds=None):
        B, L, _ = x.shape
        D, nq = self.in_dim, self.latents.shape[1]
        dt = self.latents.dtype
        x = ops.l2normalize_dim1(x.to(dt).contiguous())                                  # F.normalize, dim=1 (:269)
        x = ops.gemm(x.view(B * L, -1), self.proj_in.weight.data, bias=self.proj_in.bias.data)      # (:271)
        lat = self.latents.data.expand(B, nq, D).reshape(B * nq, D).contiguous()
        scale = 1.0 / math.sqrt(self.dim_head)   # (q*s)(k*s)^T with s = dim_head^-1/4  (:69-70)
        for attn, ff in self.layers:
            xn = _ln(x, attn.norm1)
            ln = _ln(lat, attn.norm2)
            q = ops.gemm(ln, attn.to_q.weight.data)                                        # [B*nq, inner]
            kv_in = torch.cat((xn.view(B, L, D), ln.view(B, nq, D)), dim=1).contiguous()    # (:61)
            kv = ops.gemm(kv_in.view(B * (L + nq), D), attn.to_kv.weight.data)             # [.., 2*inner]
            inner = q.shape[1]
            kv = kv.view(B, L + nq, 2 * inner)
            k = kv[:, :, :inner].contiguous()
            v = kv[:, :, inner:].contiguous()
            o = ops.attention(q.view(B, nq, inner), k, v, self.heads, scale, False)         # (:69-72)
            lat = ops.gemm(o.view(B * nq, inner), attn.to_out.weight.data, residual=lat)   # attn(x,lat)+lat (:274)
            y = _ln(lat, ff[0])
            y = _gelu_linear(y, ff[1].weight.data)                                          # Linear -> GELU (:14-15)
            lat = ops.gemm(y, ff[3].weight.data, residual=lat)                             # ff(lat)+lat (:275)
        hidden = _ln(lat, self.norm_out)                                                    # [B*nq, D]
        e1 = ops.gemm(hidden, self.unet_proj_1.weight.data, bias=self.unet_proj_1.bias.data)
        e2 = ops.gemm(hidden, self.unet_proj_2.weight.data, bias=self.unet_proj_2.bias.data)
        prompt = torch.cat([e1.view(B, nq, -1), e2.view(B, nq, -1)], dim=-1)
        # AttentionPool2d (:90-118): mean token prepended, + positional embedding, query = token 0
        ap = self.unet_attnpool
        h3 = hidden.view(B, nq, D)
        mean_tok = _mean_tokens(h3)
        t = torch.cat([mean_tok, h3], dim=1).contiguous()
        t = ops.add_bcast(t, ap.positional_embedding.data.to(dt).contiguous())
        tf = t.view(B * (nq + 1), D)
        qp = ops.gemm(t[:, :1].reshape(B, D).contiguous(), ap.q_proj.weight.data, bias=ap.q_proj.bias.data)
        kp = ops.gemm(tf, ap.k_proj.weight.data, bias=ap.k_proj.bias.data).view(B, nq + 1, D)
        vp = ops.gemm(tf, ap.v_proj.weight.data, bias=ap.v_proj.bias.data).view(B, nq + 1, D)
        o = ops.attention(qp.view(B, 1, D), kp, vp, ap.num_heads, None, False)
        pooled = ops.gemm(o.view(B, D), ap.c_proj.weight.data, bias=ap.c_proj.bias.data)
        return prompt, pooled


def _gelu_linear(y, w):
    """Linear(no bias) + exact GELU (FeedForward, :13-16) via the GEMM's GELU epilogue with a zero bias."""
    zero = torch.zeros(w.shape[0], dtype=y.dtype, device=y.device)
    return ops.gemm(y, w, bias=zero, gelu=True)


def _mean_tokens(h3):
    """x.mean(dim=tokens) as a GEMM with a constant 1/L row: mean[b, :] = (1/L) * sum_l h3[b, l, :]."""
    B, L, D = h3.shape
    ones = torch.full((1, L), 1.0 / L, dtype=h3.dtype, device=h3.device)
    outs = [ops.gemm(ones, h3[b].t().contiguous()) for b in range(B)]
    return torch.stack(outs, dim=0)
