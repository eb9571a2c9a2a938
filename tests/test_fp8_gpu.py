"""fp8 (OCP e4m3fn) GEMM path of the SDXL UNet's linear layers (SURVEY.md §8 ★ row, BASELINE configs[4]).

The reference has no fp8 arithmetic, so parity has two parts:
  (1) the kernels are EXACT with respect to their quantised operands — the quantiser reproduces torch's
      ``.to(torch.float8_e4m3fn)`` byte for byte, and ``ss_gemm_fp8`` equals the fp64 product of the de-quantised
      operands up to the final bf16 rounding (products of e4m3 values are exact in fp32; only the summation order differs);
  (2) the quantisation error against the bf16 path is BOUNDED and reported: per GEMM, per full-size transformer block,
      and for the whole SDXL-base-shaped UNet forward."""
import math

import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
F8 = torch.float8_e4m3fn


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def torch_quant(x):
    """The quantiser restated with torch ops (fp32 arithmetic, RNE cast)."""
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    inv = torch.where(amax > 0, torch.full_like(amax, 448.0) / amax, torch.zeros_like(amax))   # true division (scalar / tensor is scalar * reciprocal)
    q = (xf * inv[:, None]).to(F8)
    return q.view(torch.uint8), amax / 448.0


def deq(q, s):
    return q.view(F8).double() * s.double()[:, None]


@pytest.mark.parametrize("M,K", [(7, 128), (300, 640), (1024, 1280), (65, 5120), (4, 8)])
def test_quantize_rows_matches_torch_bytes(M, K):
    from seedstory import ops
    x = (synth.normal_like(M + K, (M, K), 1.0) * torch.logspace(-3, 2, M)[:, None]).to(BF)
    x[0, :] = 0                                       # an all-zero row: scale 0, bytes 0
    q, s = ops.quantize_rows_fp8(x.to(DEV))
    qr, sr = torch_quant(x)
    assert torch.equal(s.cpu(), sr)
    assert torch.equal(q.cpu(), qr), int((q.cpu() != qr).sum())
    assert float(deq(q.cpu(), s.cpu()).abs().max()) > 0 and rel(deq(q.cpu(), s.cpu()), x) < 4e-2


def test_quantize_rows_fused_layernorm():
    from seedstory import ops
    M, K = 520, 1280
    x = synth.normal_like(3, (M, K), 2.0).to(BF).to(DEV)
    g = synth.normal_like(4, (K,), 0.2, 1.0).to(BF).to(DEV)
    b = synth.normal_like(5, (K,), 0.1).to(BF).to(DEV)
    y = ops.layernorm(x, g, b, 1e-5)
    q0, s0 = ops.quantize_rows_fp8(y)
    q1, s1 = ops.quantize_rows_fp8(x, ln=(g, b, 1e-5))
    # same values up to the bf16 rounding of the normalised row falling on the other side in a few places
    assert rel(s1, s0) < 5e-3
    d = (deq(q1.cpu(), s1.cpu()) - deq(q0.cpu(), s0.cpu())).abs()
    assert float((d > 0).double().mean()) < 0.02 and rel(deq(q1.cpu(), s1.cpu()), y) < 4e-2


def _ref_gemm(a8, sa, w8, sw, bias, residual, gelu, geglu):
    c = deq(a8.cpu(), sa.cpu()) @ deq(w8.cpu(), sw.cpu()).T
    if bias is not None:
        c = c + bias.double().cpu()
    if gelu:
        c = torch.nn.functional.gelu(c)
    if geglu:
        c = c[:, 0::2] * torch.nn.functional.gelu(c[:, 1::2])
    if residual is not None:
        c = c + residual.double().cpu()
    return c


@pytest.mark.parametrize("cfg", [0, 80, 81, 82, 85, 86, 88])
@pytest.mark.parametrize("M,N,K", [(512, 640, 128), (300, 320, 256), (1000, 1280, 384), (257, 160, 640), (64, 4000, 1280)])
def test_gemm_fp8_is_exact_on_its_quantised_operands(cfg, M, N, K):
    from seedstory import _lib, ops
    a = synth.normal_like(M + 1, (M, K), 1.0).to(BF).to(DEV)
    w = synth.normal_like(N + 2, (N, K), 1.0 / math.sqrt(K)).to(BF).to(DEV)
    bias = synth.normal_like(7, (N,), 0.5).to(BF).to(DEV)
    res = synth.normal_like(8, (M, N), 1.0).to(BF).to(DEV)
    a8, sa = ops.quantize_rows_fp8(a)
    w8, sw = ops.quantize_rows_fp8(w)
    _lib.set_tuning("gemm_fp8_cfg", cfg)
    try:
        for kw in (dict(), dict(bias=bias), dict(bias=bias, residual=res), dict(bias=bias, gelu=True), dict(bias=bias, geglu=True)):
            if kw.get("geglu") and N % 2:
                continue
            y = ops.gemm_fp8(a8, sa, w8, sw, **kw)
            ref = _ref_gemm(a8, sa, w8, sw, kw.get("bias"), kw.get("residual"), kw.get("gelu", False), kw.get("geglu", False))
            assert y.dtype == BF and y.shape == ref.shape
            e = rel(y, ref)
            # the bf16 rounding of the result (1.7e-3 rms); GELU / GEGLU round the pre-activation to bf16 as well, like
            # the bf16 path's epilogue (reference semantics: nn.GELU on a bf16 tensor).  Anything structural is O(1).
            assert e < (6e-3 if (kw.get("gelu") or kw.get("geglu")) else 3e-3), (kw.keys(), e)
            assert float((y.double().cpu() - ref).abs().max()) < 0.02 * float(ref.abs().max()) + 1e-3
    finally:
        _lib.set_tuning("gemm_fp8_cfg", 0)


def test_gemm_fp8_identity_times_asymmetric_matrix():
    """A = I (exactly representable) against an asymmetric W: the result must be W^T's bytes back — catches any row/column
    or k-block permutation in the 128-deep fragment layout."""
    from seedstory import ops
    K = N = 256
    a = torch.eye(K, dtype=BF, device=DEV)
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 13 - 6).to(BF).to(DEV)     # integers -6 .. 6
    w[:, 0] = 7.0                                         # every row has amax 7: q = 64 w, exactly representable in e4m3
    a8, sa = ops.quantize_rows_fp8(a)
    w8, sw = ops.quantize_rows_fp8(w)
    y = ops.gemm_fp8(a8, sa, w8, sw)
    assert torch.equal(y.float().cpu(), w.float().cpu().T)


@pytest.mark.parametrize("M,N,K,geglu", [(8192, 1280, 1280, False), (8192, 10240, 1280, True), (8192, 1280, 5120, False),
                                         (32768, 640, 640, False), (32768, 1920, 640, False)])
def test_gemm_fp8_quantisation_error_vs_bf16(M, N, K, geglu):
    """UNet shapes at batch 8: the fp8 result vs the bf16 GEMM of the same bf16 operands, both against the fp64 product."""
    from seedstory import ops
    a = synth.normal_like(11, (M, K), 1.0).to(BF).to(DEV)
    w = synth.normal_like(12, (N, K), 1.0 / math.sqrt(K)).to(BF).to(DEV)
    bias = synth.normal_like(13, (N,), 0.1).to(BF).to(DEV)
    a8, sa = ops.quantize_rows_fp8(a)
    w8, sw = ops.quantize_rows_fp8(w)
    y8 = ops.gemm_fp8(a8, sa, w8, sw, bias=bias, geglu=geglu)
    if geglu:
        y16 = ops.gemm_geglu(a, w, bias)
    else:
        y16 = ops.gemm(a, w, bias=bias)
    rows = torch.arange(0, M, 97)
    c = a[rows].double().cpu() @ w.double().cpu().T + bias.double().cpu()
    ref = c[:, 0::2] * torch.nn.functional.gelu(c[:, 1::2]) if geglu else c
    e8, e16 = rel(y8[rows.to(DEV)], ref), rel(y16[rows.to(DEV)], ref)
    print("fp8 GEMM [%d,%d,%d]%s: fp8 vs fp64 %.3e | bf16 vs fp64 %.3e" % (M, N, K, " geglu" if geglu else "", e8, e16))
    assert e8 < 6e-2 and e16 < 5e-3


@pytest.mark.parametrize("name,ch,heads,res", [("mid_block.attentions.0", 1280, 20, 32), ("down_blocks.1.attentions.0", 640, 10, 64)])
def test_transformer_block_fp8_vs_bf16(name, ch, heads, res):
    from seedstory.diffusion import UNet2DConditionModel
    m = UNet2DConditionModel().to(DEV, BF).init_synthetic(1)
    B, G = 2, 32
    x = synth.normal_like(41, (B * res * res, ch), 1.0).to(BF).to(DEV)
    ctx = synth.normal_like(42, (B * 64, 2048), 1.0).to(BF).to(DEV)
    outs = {}
    for mode in (False, True):
        m.enable_fp8(mode)
        P = m._prepare()
        m._ctx_kv = {}
        outs[mode] = m._transformer(P, name, x, B, res * res, ctx, 64, heads, 1, G).float().cpu()
        m._ctx_kv = {}
        assert any(k.endswith(".fp8") for k in P) == mode
    e = rel(outs[True] - x.float().cpu(), outs[False] - x.float().cpu())     # the block's own contribution (residual removed)
    print("transformer block %s: fp8 vs bf16 path, relative deviation of the block's update %.3e" % (name, e))
    assert e < 8e-2


def test_unet_forward_fp8_vs_bf16_full_size():
    """Whole SDXL-base-shaped UNet (synthetic weights), batch 2 (the CFG pair): eps prediction, fp8 linears vs bf16."""
    from seedstory.diffusion import UNet2DConditionModel
    m = UNet2DConditionModel().to(DEV, BF).init_synthetic(1)
    x = synth.normal_like(5, (2, 4, 128, 128), 1.0).to(BF).to(DEV)
    ctx = synth.normal_like(6, (2, 64, 2048), 1.0).to(BF).to(DEV)
    cond = {"text_embeds": synth.normal_like(7, (2, 1280), 1.0).to(BF).to(DEV),
            "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * 2, dtype=torch.float32)}
    y16 = m(x, 500.0, ctx, added_cond_kwargs=cond).sample.float().cpu()
    m.enable_fp8(True)
    y8 = m(x, 500.0, ctx, added_cond_kwargs=cond).sample.float().cpu()
    m.enable_fp8(False)
    y16b = m(x, 500.0, ctx, added_cond_kwargs=cond).sample.float().cpu()
    assert rel(y16b, y16) < 2e-3                       # switching back restores the bf16 path (GroupNorm statistics are
    #                                                    accumulated with fp32 atomics: run-to-run identical up to bf16 flips)
    e = rel(y8, y16)
    print("UNet forward (SDXL-base shape, synthetic weights): fp8 linears vs bf16, rel %.3e, max |d| %.3e (|eps| rms %.3e)"
          % (e, float((y8 - y16).abs().max()), float(y16.pow(2).mean().sqrt())))
    # 70 transformer blocks, each ~7e-2 off in its update, on RANDOM weights (no trained structure damps the drift);
    # the number is reported, the gate only catches a broken path
    assert torch.isfinite(y8).all() and e < 0.6


def test_whole_render_fp8_vs_bf16_image_deviation():
    """The number a user of ``--unet-fp8`` needs: the SAME render (conditioning, seed-42 latents, 8 Euler + CFG steps through
    the pipeline, SDXL-base-shaped UNet with synthetic weights, full-size VAE decode to 1024^2 uint8) with bf16 and with fp8
    (e4m3) transformer linears — deviation of the final image in uint8 levels and of the latents.  Random weights are the
    chaotic worst case (two bf16 runs of this net differ by 2e-2 when a GroupNorm sum differs in its last bit), so the gate
    is loose; the printed numbers are the record."""
    from seedstory import ops
    from seedstory.diffusion import AutoencoderKL, EulerDiscreteScheduler, StableDiffusionXLPipeline, UNet2DConditionModel
    unet = UNet2DConditionModel().to(DEV, BF).init_synthetic(31)
    vae = AutoencoderKL().to(DEV, BF).init_synthetic(32)
    pipe = StableDiffusionXLPipeline(vae=vae, unet=unet, scheduler=EulerDiscreteScheduler())
    kw = dict(prompt_embeds=synth.normal_like(1, (1, 64, 2048), 1.0).to(DEV, BF),
              negative_prompt_embeds=synth.normal_like(2, (1, 64, 2048), 1.0).to(DEV, BF),
              pooled_prompt_embeds=synth.normal_like(3, (1, 1280), 1.0).to(DEV, BF),
              negative_pooled_prompt_embeds=synth.normal_like(4, (1, 1280), 1.0).to(DEV, BF),
              guidance_scale=7.5, num_inference_steps=8, latents=synth.normal_like(5, (1, 4, 128, 128), 1.0).to(DEV, BF))
    out = {}
    for mode in (False, True):
        unet.enable_fp8(mode)
        lat = pipe(output_type="latent", **kw).images.float().cpu()
        img = pipe(output_type="pt", **kw).images.cpu()
        out[mode] = (lat, img)
    unet.enable_fp8(False)
    (l0, i0), (l1, i1) = out[False], out[True]
    assert i0.shape == (1024, 1024, 3) and i0.dtype == torch.uint8
    # context: the same render in exact-fp32 arithmetic on the same (bf16-representable) weights = how far bf16 ITSELF drifts
    u32 = UNet2DConditionModel().to(DEV, torch.float32)
    u32.load_state_dict({k: v.float() for k, v in unet.state_dict().items()})
    p32 = StableDiffusionXLPipeline(vae=None, unet=u32, scheduler=EulerDiscreteScheduler())
    kw32 = {k: (v.float() if torch.is_tensor(v) else v) for k, v in kw.items()}
    lt = p32(output_type="latent", **kw32).images.float().cpu()
    del u32, p32
    torch.cuda.empty_cache()
    d = (i0.int() - i1.int()).abs().float()
    print("whole render (8 Euler + CFG steps, SDXL-base shape, RANDOM weights): latents vs the fp32-arithmetic render: bf16 %.3e, "
          "fp8 linears %.3e; fp8 vs bf16 %.3e | decoded image (a random-weight VAE saturates: pixels carry no structure) "
          "uint8 mean |dev| %.1f" % (rel(l0, lt), rel(l1, lt), rel(l1, l0), float(d.mean())))
    assert torch.isfinite(l1).all() and rel(l1, l0) < 0.6
    # measured on MI355X: bf16 2.0e-2, fp8 2.4e-1 from the fp32 render (each random-weight forward is already 0.18 - 0.20 off
    # with fp8 linears: no trained structure damps the drift) — recorded, gated only against a broken path
    assert rel(l1, lt) < 0.6
