"""Tensor-level wrappers over the C ABI: each function takes torch CUDA(HIP) tensors, enqueues
the HIP kernel on torch's current stream and returns the output tensor.  torch is plumbing here
(device memory + streams); all arithmetic happens in ``libseedstory_hip.so``."""
import math

import torch

from . import _lib, tune
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


def attention_cache_slots(q, kcache, vcache, kv_lens, causal_br=True):
    """Stacked slots: q [S * M, n_heads*hd] (M rows per slot) against caches [S, n_heads, cap, hd]; slot b attends to its
    first kv_lens[b] entries (ss_attention_ragged: one launch for the S slots)."""
    import ctypes
    _req(q); _req(kcache)
    S, n_heads, cap, hd = kcache.shape
    E = n_heads * hd
    M = q.shape[0] // S
    out = torch.empty_like(q)
    lens = (ctypes.c_int32 * S)(*[int(x) for x in kv_lens])
    check(lib().ss_attention_ragged(p(q), p(kcache), p(vcache), p(out), S, n_heads, M, ctypes.cast(lens, ctypes.c_void_p), hd,
                                    M * E, hd, E, n_heads * cap * hd, cap * hd, hd, n_heads * cap * hd, cap * hd, hd, M * E, hd, E,
                                    1.0 / math.sqrt(hd), int(causal_br), dt(q), stream()), "ss_attention_ragged")
    return out


def attn_decode(q, kcache, vcache, kv_len_dev):
    _req(q); _req(kcache)
    n_heads, cap, hd = kcache.shape
    ws = torch.empty(lib().ss_attn_decode_workspace_bytes(n_heads, hd), dtype=torch.uint8, device=q.device)
    out = torch.empty_like(q)
    check(lib().ss_attn_decode(p(q), p(kcache), p(vcache), p(out), p(ws), p(kv_len_dev), n_heads, hd, cap, dt(q),
                               stream()), "ss_attn_decode")
    return out


def gemm(a, w, bias=None, residual=None, gelu=False, out=None, rowstat=None, rowpart=None):
    """a [M, K] @ w[N, K]^T (+bias)(+gelu)(+residual) -> [M, N].  ``rowstat`` (fp64 [M, 2], zeroed by the caller): the
    epilogue also accumulates (sum, sum of squares) of every stored output row into it (ss_gemm_rowstat)."""
    _req(a); _req(w)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    epi = (EPI_BIAS if bias is not None else 0) | (EPI_GELU if gelu else 0) | (EPI_RESIDUAL if residual is not None else 0)
    tune.ensure_gemm(M, N, K, dt(a), epi, a.device)
    if rowstat is not None:
        assert rowstat.dtype == torch.float64 and rowstat.is_contiguous() and rowstat.numel() == 2 * M
        check(lib().ss_gemm_rowstat(p(a), p(w), p(out), M, N, K, K, w.stride(0), N, p(bias), p(residual), N, epi, p(rowstat),
                                    dt(a), stream()), "ss_gemm_rowstat")
        return out
    if rowpart is not None:      # [M, strips, 2] fp32: per-strip (sum, sum of squares) of the stored rows, no atomics
        assert rowpart.dtype == torch.float32 and rowpart.is_contiguous() and rowpart.shape[0] == M
        if rowpart.shape[1] != rowpart_strips(M, N, K, a.dtype):
            raise _lib.SSError("gemm(rowpart=): buffer has %d strips, this shape's tile writes %d"
                               % (rowpart.shape[1], rowpart_strips(M, N, K, a.dtype)))
        check(lib().ss_gemm_rowpart(p(a), p(w), p(out), M, N, K, K, w.stride(0), N, p(bias), p(residual), N, epi, p(rowpart),
                                    dt(a), stream()), "ss_gemm_rowpart")
        return out
    check(lib().ss_gemm(p(a), p(w), p(out), M, N, K, K, w.stride(0), N, p(bias), p(residual), N, epi, dt(a), stream()),
          "ss_gemm")
    return out


def gemm_splitk(a, w, bias=None, residual=None):
    """Small-M weight-streaming GEMM (ss_gemm_splitk): a [M, K] @ w[N, K]^T for 128 < M <= 512; falls back to ss_gemm."""
    _req(a); _req(w)
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    nbytes = lib().ss_gemm_splitk_workspace_bytes(M, N, K)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=a.device)
    check(lib().ss_gemm_splitk(p(a), p(w), p(out), M, N, K, p(bias), p(residual), p(ws), nbytes, dt(a), stream()), "ss_gemm_splitk")
    return out


def gemm_geglu(a, w_pairs, bias_pairs):
    """a [M, K] @ w_pairs[2D, K]^T with rows interleaved (value_i, gate_i) -> [M, D] = value * gelu(gate)."""
    _req(a); _req(w_pairs)
    M, K = a.shape
    N = w_pairs.shape[0]
    out = torch.empty(M, N // 2, dtype=a.dtype, device=a.device)
    tune.ensure_gemm(M, N, K, dt(a), _lib.EPI_GEGLU_PAIR, a.device)
    check(lib().ss_gemm(p(a), p(w_pairs), p(out), M, N, K, K, K, N // 2, p(bias_pairs), None, 0,
                        _lib.EPI_BIAS | _lib.EPI_GEGLU_PAIR, dt(a), stream()), "ss_gemm(geglu)")
    return out


def rowstats(x, eps):
    """Per-row LayerNorm statistics for `gemm_lnfold`: (rstd [M], -mean * rstd [M]) fp32."""
    _req(x)
    M, K = x.shape
    st = torch.empty(2, M, dtype=torch.float32, device=x.device)
    check(lib().ss_rowstats(p(x), x.stride(0), M, K, float(eps), p(st[0]), p(st[1]), dt(x), stream()), "ss_rowstats")
    return st[0], st[1]


def gemm_lnfold(x, wg, rstd, shift, colsum, bias_d=None, gelu=False, geglu=False):
    """LN(x) @ W^T + b with the LayerNorm folded: x raw rows, wg = gamma-scaled weight, colsum = wg.sum(1) fp32, bias_d = d."""
    _req(x); _req(wg)
    M, K = x.shape
    N = wg.shape[0]
    No = N // 2 if geglu else N
    out = torch.empty(M, No, dtype=x.dtype, device=x.device)
    epi = (EPI_BIAS if bias_d is not None else 0) | (EPI_GELU if gelu else 0) | (_lib.EPI_GEGLU_PAIR if geglu else 0)
    tune.ensure_gemm(M, N, K, dt(x), epi, x.device)
    check(lib().ss_gemm_lnfold(p(x), p(wg), p(out), M, N, K, No, p(rstd), p(shift), p(colsum), p(bias_d), epi, dt(x), stream()),
          "ss_gemm_lnfold")
    return out


def rowpart_strips(M, N, K, dtype, epi=0):
    """Number of (sum, sum of squares) partials per row that ``gemm(..., rowpart=)`` writes for this shape (0: not eligible).
    The answer depends on the tile the table holds for the shape, so the shape is tuned FIRST (a first forward on a
    shape absent from the table used to size the strip buffer from the closed-form tile and then re-tune inside `gemm`) — with
    the PRODUCER's epilogue flags (`epi`: GELU / GEGLU), because the table key has no epilogue component and the first tune of
    a shape is the one that sticks."""
    tune.ensure_gemm(M, N, K, dt(dtype), int(epi), None)
    return int(lib().ss_gemm_rowpart_strips(M, N, K, dt(dtype)))


def gemm_lnfold_part(x, wg, rowpart, width, eps, colsum, bias_d=None, gelu=False, geglu=False):
    """`gemm_lnfold` fed by the producer's per-strip partials ``rowpart`` [M, strips, 2]: the row statistics are formed in the
    GEMM's own epilogue (no finalize launch)."""
    _req(x); _req(wg); _req(rowpart)
    M, K = x.shape
    N = wg.shape[0]
    No = N // 2 if geglu else N
    out = torch.empty(M, No, dtype=x.dtype, device=x.device)
    epi = (EPI_BIAS if bias_d is not None else 0) | (EPI_GELU if gelu else 0) | (_lib.EPI_GEGLU_PAIR if geglu else 0)
    tune.ensure_gemm(M, N, K, dt(x), epi, x.device)
    check(lib().ss_gemm_lnfold_part(p(x), p(wg), p(out), M, N, K, No, p(rowpart), rowpart.shape[1], int(width), float(eps),
                                    p(colsum), p(bias_d), epi, dt(x), stream()), "ss_gemm_lnfold_part")
    return out


def rowstat_finalize(rowstat, width, eps, out=None):
    """(sum, sum of squares) accumulated by ``gemm(..., rowstat=)`` -> (rstd [M], -mean * rstd [M]) fp32 for
    `gemm_lnfold`; ``rowstat`` is re-zeroed for its next producer."""
    M = rowstat.shape[0]
    st = torch.empty(2, M, dtype=torch.float32, device=rowstat.device) if out is None else out
    check(lib().ss_rowstat_finalize(p(rowstat), M, int(width), float(eps), p(st[0]), p(st[1]), stream()), "ss_rowstat_finalize")
    return st[0], st[1]


def quantize_rows_fp8(x, ln=None):
    """x [M, K] (bf16 / fp16) -> (q uint8 [M, K] holding OCP e4m3 bytes, scale fp32 [M]); ``ln`` = (gamma, beta, eps)
    fuses a LayerNorm in front (the normalised bf16 tensor is never written)."""
    _req(x)
    M, K = x.shape
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    sc = torch.empty(M, dtype=torch.float32, device=x.device)
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    check(lib().ss_quantize_rows_fp8(p(x), x.stride(0), M, K, p(q), p(sc), p(g), p(b), float(eps), dt(x), stream()),
          "ss_quantize_rows_fp8")
    return q, sc


def gemm_fp8(a8, sa, w8, sw, bias=None, residual=None, gelu=False, geglu=False, out=None):
    """(a8 * sa[:, None]) @ (w8 * sw[:, None])^T (+bias)(+gelu | geglu pairs)(+residual) -> bf16 [M, N (N/2 for geglu)]."""
    M, K = a8.shape
    N = w8.shape[0]
    No = N // 2 if geglu else N
    if out is None:
        out = torch.empty(M, No, dtype=torch.bfloat16, device=a8.device)
    epi = (EPI_BIAS if bias is not None else 0) | (EPI_GELU if gelu else 0) | (EPI_RESIDUAL if residual is not None else 0) | \
        (_lib.EPI_GEGLU_PAIR if geglu else 0)
    check(lib().ss_gemm_fp8(p(a8), p(sa), p(w8), p(sw), p(out), M, N, K, No, p(bias), p(residual), No, epi, stream()), "ss_gemm_fp8")
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


def gemv_batched(w, x, norm_w=None, eps=0.0, bias=None, residual=None, silu_mul=False):
    """x [nb, K] (nb <= 4) -> y [nb, N]: one sweep of W for all rows."""
    _req(w); _req(x)
    N, K = w.shape
    if silu_mul:
        N //= 2
    nb = x.shape[0]
    y = torch.empty(nb, N, dtype=x.dtype, device=x.device)
    epi = (EPI_BIAS if bias is not None else 0) | (EPI_RESIDUAL if residual is not None else 0) | \
          (EPI_SILU_MUL if silu_mul else 0)
    check(lib().ss_gemv_batched(p(w), p(x), p(y), N, K, nb, p(norm_w), eps, p(bias), p(residual), epi, dt(x),
                                stream()), "ss_gemv_batched")
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


# ---- SDXL de-tokenizer ops (NHWC activations: [B, H*W, C]) ------------------------------------------
def conv3x3(x, w, B, H, W, stride=1, upsample=False, bias=None, rowvec=None, residual=None, rowvec_stride=0):
    """x [B*H*W, Cin] (NHWC) -> [B*Ho*Wo, Cout]; w [Cout, 9*Cin] (tap-major, channel-minor)."""
    _req(x); _req(w)
    Cin = x.shape[-1]
    Cout = w.shape[0]
    Hin, Win = (2 * H, 2 * W) if upsample else (H, W)
    Ho, Wo = (Hin + 2 - 3) // stride + 1, (Win + 2 - 3) // stride + 1
    y = torch.empty(B * Ho * Wo, Cout, dtype=x.dtype, device=x.device)
    tune.ensure_conv(B, H, W, Cin, Cout, stride, upsample, dt(x), x.device)
    check(lib().ss_conv3x3(p(x), p(w), p(y), B, H, W, Cin, Cout, stride, int(upsample), p(bias), p(rowvec),
                           rowvec_stride, p(residual), dt(x), stream()), "ss_conv3x3")
    return y, Ho, Wo


def groupnorm(x, gamma, beta, B, groups, eps, silu=False):
    """x [B*HW, C] NHWC"""
    _req(x)
    C = x.shape[-1]
    HW = x.shape[0] // B
    y = torch.empty_like(x)
    nb = int(lib().ss_groupnorm_workspace_bytes(B, HW, C, groups, dt(x)))      # final sums + deterministic block partials
    ws = torch.empty(max(nb // 8, 1), dtype=torch.float64, device=x.device)
    check(lib().ss_groupnorm(p(x), p(gamma), p(beta), p(y), p(ws), B, HW, C, groups, eps, int(silu), dt(x), stream()),
          "ss_groupnorm")
    return y


# ---- loss heads of the training-side forward (forward only) ---------------------------------------
def _mean2(vals, mask):
    n = vals.numel()
    ws = torch.empty(max(int(lib().ss_loss_workspace_bytes(n)) // 8, 1), dtype=torch.float64, device=vals.device)
    out = torch.empty(2, dtype=torch.float32, device=vals.device)
    check(lib().ss_masked_mean(p(vals), p(mask) if mask is not None else None, n, p(ws), p(out), stream()), "ss_masked_mean")
    return out


def cross_entropy(logits, labels, ignore_index=-100):
    """CrossEntropyLoss()(logits [R, V], labels [R]) as modeling_llama_xformer.py:769-772 calls it: the mean over the rows whose
    label is not ``ignore_index`` -> (loss scalar tensor fp32, number of valid rows tensor)."""
    _req(logits)
    R, V = logits.shape
    labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
    assert labels.numel() == R and logits.stride(1) == 1
    row = torch.empty(R, dtype=torch.float32, device=logits.device)
    valid = torch.empty(R, dtype=torch.float32, device=logits.device)
    check(lib().ss_cross_entropy_rows(p(logits), logits.stride(0), p(labels), R, V, ignore_index, p(row), p(valid), dt(logits),
                                      stream()), "ss_cross_entropy_rows")
    out = _mean2(row, valid)
    return out[0], out[1]


def cosine_loss(rec, target):
    """models.py:13-17: (1 - (target/|target| * rec/|rec|).sum(-1)).mean() -> scalar tensor fp32."""
    _req(rec)
    _req(target)
    assert rec.shape == target.shape and rec.dtype == target.dtype
    dim = rec.shape[-1]
    r2, t2 = rec.reshape(-1, dim).contiguous(), target.reshape(-1, dim).contiguous()
    val = torch.empty(r2.shape[0], dtype=torch.float32, device=rec.device)
    check(lib().ss_cosine_rows(p(r2), p(t2), r2.shape[0], dim, p(val), dt(rec), stream()), "ss_cosine_rows")
    return _mean2(val, None)[0]


def mse_loss(a, b):
    """F.mse_loss(a.float(), b.float(), reduction='mean') (adapter_modules.py:339) -> scalar tensor fp32."""
    _req(a)
    _req(b)
    assert a.shape == b.shape and a.dtype == b.dtype
    a2, b2 = a.contiguous(), b.contiguous()
    n = a2.numel()
    ws = torch.empty(max(int(lib().ss_loss_workspace_bytes(n)) // 8, 1), dtype=torch.float64, device=a.device)
    out = torch.empty(2, dtype=torch.float32, device=a.device)
    check(lib().ss_mse(p(a2), p(b2), n, p(ws), p(out), dt(a), stream()), "ss_mse")
    return out[0]


def geglu(x):
    _req(x)
    rows, two_d = x.shape
    out = torch.empty(rows, two_d // 2, dtype=x.dtype, device=x.device)
    check(lib().ss_geglu(p(x), p(out), rows, two_d // 2, dt(x), stream()), "ss_geglu")
    return out


def silu(x):
    _req(x)
    y = torch.empty_like(x)
    check(lib().ss_unary(p(x), p(y), x.numel(), 0, dt(x), stream()), "ss_unary")
    return y


def attention_qkv_packed(qkv, B, L, n_heads):
    """Self-attention on a fused projection qkv [B*L, 3E] = [q | k | v] per token -> [B*L, E]."""
    _req(qkv)
    E = qkv.shape[1] // 3
    hd = E // n_heads
    out = torch.empty(B * L, E, dtype=qkv.dtype, device=qkv.device)
    esz = qkv.element_size()
    base = qkv.data_ptr()
    check(lib().ss_attention(base, base + E * esz, base + 2 * E * esz, p(out), B, n_heads, L, L, hd, L * 3 * E, hd, 3 * E,
                             L * 3 * E, hd, 3 * E, L * 3 * E, hd, 3 * E, L * E, hd, E, 1.0 / math.sqrt(hd), 0, dt(qkv),
                             stream()), "ss_attention")
    return out


def attention_q_kvpacked(q, kv, B, Lq, Lk, n_heads):
    """Cross-attention: q [B*Lq, E], kv [B*Lk, 2E] = [k | v] per context token -> [B*Lq, E]."""
    _req(q); _req(kv)
    E = q.shape[1]
    hd = E // n_heads
    out = torch.empty_like(q)
    esz = q.element_size()
    kb = kv.data_ptr()
    check(lib().ss_attention(p(q), kb, kb + E * esz, p(out), B, n_heads, Lq, Lk, hd, Lq * E, hd, E, Lk * 2 * E, hd, 2 * E,
                             Lk * 2 * E, hd, 2 * E, Lq * E, hd, E, 1.0 / math.sqrt(hd), 0, dt(q), stream()), "ss_attention")
    return out


def softmax_rows_(s, scale):
    _req(s)
    check(lib().ss_softmax_rows(p(s), s.shape[0], s.shape[1], scale, dt(s), stream()), "ss_softmax_rows")
    return s


def transpose(x):
    _req(x)
    R, Cc = x.shape
    out = torch.empty(Cc, R, dtype=x.dtype, device=x.device)
    check(lib().ss_transpose(p(x), p(out), R, Cc, dt(x), stream()), "ss_transpose")
    return out


def concat_channels(a, b):
    _req(a); _req(b)
    rows = a.shape[0]
    out = torch.empty(rows, a.shape[1] + b.shape[1], dtype=a.dtype, device=a.device)
    check(lib().ss_concat_channels(p(a), p(b), p(out), rows, a.shape[1], b.shape[1], dt(a), stream()),
          "ss_concat_channels")
    return out


def nchw_to_nhwc(x, cpad):
    _req(x)
    B, Cc = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    out = torch.empty(B * HW, cpad, dtype=x.dtype, device=x.device)
    check(lib().ss_layout_nchw_nhwc(p(x), p(out), B, Cc, HW, cpad, 1, dt(x), stream()), "ss_layout_nchw_nhwc")
    return out


def nhwc_to_nchw(x, B, C, H, W):
    _req(x)
    out = torch.empty(B, C, H, W, dtype=x.dtype, device=x.device)
    check(lib().ss_layout_nchw_nhwc(p(x), p(out), B, C, H * W, x.shape[-1], 0, dt(x), stream()), "ss_layout_nchw_nhwc")
    return out


def euler_scale_dup(x, sigma):
    _req(x)
    xin = torch.empty((2,) + tuple(x.shape), dtype=x.dtype, device=x.device)
    check(lib().ss_euler_scale_dup(p(x), p(xin), x.numel(), float(sigma), dt(x), stream()), "ss_euler_scale_dup")
    return xin


def euler_cfg_step_(x, eps, guidance, sigma, sigma_next):
    _req(x); _req(eps)
    check(lib().ss_euler_cfg_step(p(x), p(eps), x.numel(), float(guidance), float(sigma), float(sigma_next), dt(x),
                                  stream()), "ss_euler_cfg_step")
    return x


def image_to_u8(x, pixels):
    _req(x)
    out = torch.empty(pixels, 3, dtype=torch.uint8, device=x.device)
    check(lib().ss_image_to_u8(p(x), p(out), pixels, x.shape[-1], dt(x), stream()), "ss_image_to_u8")
    return out
