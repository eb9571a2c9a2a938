"""LayerNorm folded into the consuming GEMM (``ss_rowstats`` + ``ss_gemm_lnfold``; UNet norm1/2/3 -> q|k|v / to_q / GEGLU ff1).
The fold changes WHERE values are rounded (the normalised activations are never rounded to bf16; gamma is rounded into
the weight), so the gates are the bf16 ones: distance to the fp64 truth no larger than 1.5 x the unfused HIP path's own
distance (+ eps), and the full-size transformer block against the oracle exactly as tests/test_fulldim_gpu.py gates it."""
import math

import pytest
import torch
import torch.nn.functional as F

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("M,N,K,geglu,bias", [(8192, 3840, 1280, False, False), (8192, 1280, 1280, False, False),
                                              (2048, 10240, 1280, True, True), (4096, 1920, 640, False, False),
                                              (300, 640, 640, False, True), (1000, 5120, 640, True, True)])
def test_gemm_lnfold_vs_layernorm_then_gemm(M, N, K, geglu, bias):
    from seedstory import ops
    x = (synth.normal_like(M + N, (M, K), 1.0) * 3.0 + 0.7).to(BF).to(DEV)          # non-zero row means
    W = synth.normal_like(N + 1, (N, K), 1.0 / math.sqrt(K)).to(BF).to(DEV)
    gamma = synth.normal_like(3, (K,), 0.3, 1.0).to(BF).to(DEV)
    beta = synth.normal_like(4, (K,), 0.2).to(BF).to(DEV)
    b = synth.normal_like(5, (N,), 0.3).to(BF).to(DEV) if bias else None
    # fold (what UNet._prepare_lnfold does)
    wg = (W.float() * gamma.float()[None, :]).to(BF).contiguous()
    c = wg.float().sum(1).contiguous()
    d = W.float() @ beta.float()
    if b is not None:
        d = d + b.float()
    rstd, shift = ops.rowstats(x, 1e-5)
    y = ops.gemm_lnfold(x, wg, rstd, shift, c, bias_d=d.to(BF).contiguous(), geglu=geglu)
    # unfused HIP path
    ln = ops.layernorm(x, gamma, beta, 1e-5)
    y0 = ops.gemm_geglu(ln, W, b if b is not None else torch.zeros(N, device=DEV, dtype=BF)) if geglu else ops.gemm(ln, W, bias=b)
    # fp64 truth on a row sample
    rows = torch.arange(0, M, max(1, M // 64))
    xr = x[rows.to(DEV)].double().cpu()
    t = F.layer_norm(xr, (K,), gamma.double().cpu(), beta.double().cpu(), 1e-5) @ W.double().cpu().T
    if b is not None:
        t = t + b.double().cpu()
    ref = t[:, 0::2] * F.gelu(t[:, 1::2]) if geglu else t
    e_fold, e_unf = rel(y[rows.to(DEV)], ref), rel(y0[rows.to(DEV)], ref)
    print("lnfold [%d,%d,%d]%s: folded vs fp64 %.3e | unfused vs fp64 %.3e" % (M, N, K, " geglu" if geglu else "", e_fold, e_unf))
    assert y.shape == y0.shape and e_fold <= 1.5 * e_unf + 1e-3
    st = ops.rowstats(x, 1e-5)
    xf = x.float()
    assert rel(st[0], 1.0 / torch.sqrt(xf.var(1, unbiased=False) + 1e-5)) < 1e-5
    assert rel(st[1], -xf.mean(1) / torch.sqrt(xf.var(1, unbiased=False) + 1e-5)) < 1e-4


@pytest.mark.parametrize("name,ch,heads,res", [("mid_block.attentions.0", 1280, 20, 32), ("down_blocks.1.attentions.0", 640, 10, 64)])
def test_transformer_block_lnfold_full_size_vs_oracle(name, ch, heads, res):
    import sdxl_oracle as S
    from seedstory.diffusion import UNet2DConditionModel
    m = UNet2DConditionModel().to(DEV, BF).init_synthetic(1)
    m.enable_lnfold(True)
    P = m._prepare()
    assert any(k.endswith(".lnf") for k in P)
    B, G = 2, 32
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items() if k.startswith(name + ".")}
    wd = {k: v for k, v in sd.items() if ".transformer_blocks." not in k or ".transformer_blocks.0." in k}
    x = synth.normal_like(41, (B, ch, res, res), 1.0)
    ctx = synth.normal_like(42, (B, 64, 2048), 1.0)
    ref32 = S.transformer_2d(wd, name, x, ctx, heads, 1, G)
    bf = {k: v.to(BF) for k, v in wd.items()}
    refbf = S.transformer_2d(bf, name, x.to(BF), ctx.to(BF), heads, 1, G)
    xh = x.permute(0, 2, 3, 1).reshape(B * res * res, ch).contiguous().to(DEV, BF)
    outs = {}
    for mode in (True, False):
        m.enable_lnfold(mode)
        P = m._prepare()
        m._ctx_kv = {}
        y = m._transformer(P, name, xh, B, res * res, ctx.to(DEV, BF).reshape(B * 64, -1).contiguous(), 64, heads, 1, G)
        m._ctx_kv = {}
        outs[mode] = y.reshape(B, res, res, -1).permute(0, 3, 1, 2).float().cpu()
    e_bf, e_32, theirs = rel(outs[True], refbf), rel(outs[True], ref32), rel(refbf, ref32)
    u_bf, u_32 = rel(outs[False], refbf), rel(outs[False], ref32)
    print("transformer %s, LayerNorm folded: vs oracle-bf16 %.3e | vs oracle-fp32 %.3e (unfused: %.3e | %.3e; oracle bf16 vs fp32 %.3e)"
          % (name, e_bf, e_32, u_bf, u_32, theirs))
    assert e_bf < 1e-2 and e_32 <= 1.5 * theirs + 1e-3
