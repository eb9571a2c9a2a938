"""Tensor-level wrappers over the C ABI: each function takes torch CUDA(HIP) tensors, enqueues
the HIP kernel on torch's current stream and returns the output tensor.  torch is plumbing here
(device memory + streams); all arithmetic happens in ``libseedstory_hip.so``."""
import math

import torch

from . import _lib
from ._lib import EPI_BIAS, EPI_GELU, EPI_NONE, EPI_RESIDUAL, EPI_SILU_MUL, check, lib

_DT = {torch.float32: _lib.SS_F32, torch.bfloat16: _lib.SS_BF16, torch.float16: _lib.SS_F16}


def dt(t):
    try:
        return _DT[t.dtype if isinstance(t, torch.Tensor) else t]
    except KeyError:
        raise _lib.SSError("unsupported dtype %s" % (t.dtype if isinstance(t, torch.Tensor) else t))


def stream():
    return torch.cuda.current_stream().cuda_stream


def p(t):
    return None if t is None else t.data_ptr()


def _req(t, name="tensor"):
    if not t.is_cuda:
        raise _lib.SSError("%s must live on the GPU (there is no CPU path)" % name)
    if not t.is_contiguous():
        raise _lib.SSError("%s must be contiguous" % name)
    return t


def rmsnorm(x, w, eps):
    _req(x); _req(w)
    y = torch.empty_like(x)
    cols = x.shape[-1]
    check(lib().ss_rmsnorm(p(x), p(w), p(y), x.numel() // cols, cols, eps, dt(x), stream()), "ss_rmsnorm")
    return y


def layernorm(x, w, b, eps):
    _req(x); _req(w); _req(b)
    y = torch.empty_like(x)
    cols = x.shape[-1]
    check(lib().ss_layernorm(p(x), p(w), p(b), p(y), x.numel() // cols, cols, eps, dt(x), stream()), "ss_layernorm")
    return y


def add_bcast(x, pos):
    """x [B, R, C] + pos [R, C]"""
    _req(x); _req(pos)
    B, R, Cc = x.shape
    y = torch.empty_like(x)
    check(lib().ss_add_bcast(p(x), p(pos), p(y), B, R, Cc, R * Cc, dt(x), stream()), "ss_add_bcast")
    return y


def silu_mul(gu):
    _req(gu)
    rows, two_i = gu.shape
    out = torch.empty(rows, two_i // 2, dtype=gu.dtype, device=gu.device)
    check(lib().ss_silu_mul(p(gu), p(out), rows, two_i // 2, dt(gu), stream()), "ss_silu_mul")
    return out


def gather_rows(table, ids):
    _req(table)
    ids = ids.to(device=table.device, dtype=torch.int32).contiguous()
    out = torch.empty(ids.numel(), table.shape[1], dtype=table.dtype, device=table.device)
    check(lib().ss_gather_rows(p(table), p(ids), p(out), ids.numel(), table.shape[1], dt(table), stream()),
          "ss_gather_rows")
    return out


def scatter_rows_(dst, idx, src):
    _req(dst); _req(src)
    idx = idx.to(device=dst.device, dtype=torch.int32).contiguous()
    check(lib().ss_scatter_rows(p(src), p(idx), p(dst), idx.numel(), dst.shape[-1], dt(dst), stream()),
          "ss_scatter_rows")
    return dst


def im2col_patch(img, patch, kpad):
    _req(img)
    B, _, S, _ = img.shape
    G = S // patch
    out = torch.empty(B * G * G, kpad, dtype=img.dtype, device=img.device)
    check(lib().ss_im2col_patch(p(img), p(out), B, S, patch, kpad, dt(img), stream()), "ss_im2col_patch")
    return out


def l2normalize_dim1(x):
    _req(x)
    B, L, Cc = x.shape
    y = torch.empty_like(x)
    check(lib().ss_l2normalize_dim1(p(x), p(y), B, L, Cc, dt(x), stream()), "ss_l2normalize_dim1")
    return y


def rope_kv_append(qkv, kcache, vcache, cos, sin, n_heads, kv_start, pos_ids=None, pos_start=0):
    """qkv [M, 3*E]; caches [n_heads, cap, hd]; returns rotated q [M, E]."""
    _req(qkv); _req(kcache); _req(vcache)
    M = qkv.shape[0]
    E = qkv.shape[1] // 3
    hd = E // n_heads
    q = torch.empty(M, E, dtype=qkv.dtype, device=qkv.device)
    pid = None if pos_ids is None else pos_ids.to(device=qkv.device, dtype=torch.int32).contiguous()
    check(lib().ss_rope_kv_append(p(qkv), p(q), p(kcache), p(vcache), p(cos), p(sin), p(pid), pos_start, M, n_heads,
                                  hd, kv_start, kcache.shape[1], dt(qkv), stream()), "ss_rope_kv_append")
    return q


def attention(q, k, v, n_heads, scale=None, causal_br=False, out=None):
    """q [B, Lq, E], k/v [B, Lk, E] (heads packed along E) -> [B, Lq, E]."""
    _req(q); _req(k); _req(v)
    B, Lq, E = q.shape
    Lk = k.shape[1]
    hd = E // n_heads
    if out is None:
        out = torch.empty_like(q)
    scale = 1.0 / math.sqrt(hd) if scale is None else scale
    kb = 0 if k.shape[0] == 1 and B > 1 else Lk * E
    check(lib().ss_attention(p(q), p(k), p(v), p(out), B, n_heads, Lq, Lk, hd, Lq * E, hd, E, kb, hd, E, kb, hd, E,
                             Lq * E, hd, E, scale, int(causal_br), dt(q), stream()), "ss_attention")
    return out


def attention_cache(q, kcache, vcache, kv_len, causal_br=True):
    """q [M, n_heads*hd] against cache planes [n_heads, cap, hd] (first kv_len slots)."""
    _req(q); _req(kcache)
    M, E = q.shape
    n_heads, cap, hd = kcache.shape
    out = torch.empty_like(q)
    check(lib().ss_attention(p(q), p(kcache), p(vcache), p(out), 1, n_heads, M, kv_len, hd, 0, hd, E, 0, cap * hd, hd,
                             0, cap * hd, hd, 0, hd, E, 1.0 / math.sqrt(hd), int(causal_br), dt(q), stream()),
          "ss_attention")
    return out


def attn_decode(q, kcache, vcache, kv_len_dev):
    _req(q); _req(kcache)
    n_heads, cap, hd = kcache.shape
    ws = torch.empty(lib().ss_attn_decode_workspace_bytes(n_heads, hd), dtype=torch.uint8, device=q.device)
    out = torch.empty_like(q)
    check(lib().ss_attn_decode(p(q), p(kcache), p(vcache), p(out), p(ws), p(kv_len_dev), n_heads, hd, cap, dt(q),
                               stream()), "ss_attn_decode")
    return out


def gemm(a, w, bias=None, residual=None, gelu=False, out=None):
    """a [M, K] @ w[N, K]^T (+bias)(+gelu)(+residual) -> [M, N]"""
    _req(a); _req(w)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    epi = (EPI_BIAS if bias is not None else 0) | (EPI_GELU if gelu else 0) | (EPI_RESIDUAL if residual is not None else 0)
    check(lib().ss_gemm(p(a), p(w), p(out), M, N, K, K, w.stride(0), N, p(bias), p(residual), N, epi, dt(a), stream()),
          "ss_gemm")
    return out


def gemv(w, x, norm_w=None, eps=0.0, bias=None, residual=None, silu_mul=False):
    _req(w); _req(x)
    N, K = w.shape
    if silu_mul:
        N //= 2
    y = torch.empty(N, dtype=x.dtype, device=x.device)
    epi = (EPI_BIAS if bias is not None else 0) | (EPI_RESIDUAL if residual is not None else 0) | \
          (EPI_SILU_MUL if silu_mul else 0)
    check(lib().ss_gemv(p(w), p(x), p(y), N, K, p(norm_w), eps, p(bias), p(residual), epi, dt(x), stream()), "ss_gemv")
    return y


def imgproc_argmax(logits, last_id, img_ids):
    """In-place processor + argmax; returns an int32 device scalar tensor."""
    _req(logits)
    dev = logits.device
    last = torch.tensor([last_id], dtype=torch.int32, device=dev)
    ids = torch.as_tensor(list(img_ids), dtype=torch.int32, device=dev)
    tok = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().ss_imgproc_argmax(p(logits), logits.numel(), p(last), p(ids), ids.numel(), p(tok), dt(logits),
                                  stream()), "ss_imgproc_argmax")
    return tok
