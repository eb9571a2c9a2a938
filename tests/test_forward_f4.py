"""SURVEY §8 row f4 — the training-side FORWARD (no backward kernels): ``ContinuousLVLM.forward`` (reference
src/models_clm/models.py:33-96: batched splice, LLM with the shifted-label cross-entropy, cosine loss on the regressed
features) and ``SDXLAdapter.forward`` (src/models_ipa/adapter_modules.py:330-343: UNet + MSE).

Truth = ``tests/golden/forward_f4.safetensors``, written by ``oracle/make_golden_forward.py`` from the REAL reference
``ContinuousLVLM`` / ``LlamaForCausalLM`` / ``Resampler`` classes (tiny config: every tensor; hidden 4096 / 2 layers with the
real-size resamplers: losses + every 16th row of ``recon_image_embeds``).  CPU tests pin the oracle restatement on it; the
GPU tests run the PRODUCT classes through the C ABI.  The SDXL half has no reference to import (diffusers is absent):
``SDXLAdapter.forward`` is compared with the independent restatement ``oracle/sdxl_oracle.py`` (parity unpinned there).

Tolerances: fp32 1e-4 relative (summation order only).  bf16: distance to the fp32 reference <= 1.5 x the reference's OWN
bf16-vs-fp32 distance + eps (a bf16 network is a chaotic function of its rounding points)."""
import json
import os

import pytest
import torch

import make_golden_forward as MF
import seedstory_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"
DTYPES = [(torch.float32, "f32"), (torch.bfloat16, "bf16")]


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def relf(a, b):
    return abs(float(a) - float(b)) / (abs(float(b)) + 1e-30)


@pytest.fixture(scope="module")
def gold():
    from safetensors import safe_open
    from safetensors.torch import load_file
    path = os.path.join(ROOT, "tests", "golden", "forward_f4.safetensors")
    with safe_open(path, "pt") as f:
        meta = f.metadata()
    return load_file(path), json.loads(meta["report"])


# ---------------------------------------------------------------------------------------------------------------------
# CPU: the oracle restatement against the real reference's outputs (tiny configuration)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,dtag", DTYPES)
def test_oracle_lvlm_forward_tiny(gold, dtype, dtag):
    g, report = gold
    c = MF.TINY
    wd = MF.weights(c, dtype)
    b = MF.batch(c, dtype, 3)
    t = "tiny_%s." % dtag
    for k in ("input_ids", "labels"):                       # the fixture's stored inputs are what the builder regenerates
        assert torch.equal(b[k], g[t + "in." + k].to(b[k].dtype))
    assert torch.equal(b["ids_gen_mask"], g[t + "in.ids_gen_mask"].bool())
    dims = O.LlamaDims(c["hidden"], c["n_heads"], c["n_layers"], c["inter"], c["vocab"])
    with torch.no_grad():
        mine = O.lvlm_forward(wd, dims, b["input_ids"], b["labels"], b["image_embeds"], b["embeds_gen_mask"], b["embeds_cmp_mask"],
                              b["ids_gen_mask"], b["ids_cmp_mask"], n_heads_resampler=c["res_heads"])
        m0 = O.lvlm_forward(wd, dims, b["input_ids"], b["labels"], None, None, None, None, None, n_heads_resampler=c["res_heads"])
    tol_l, tol_f = (2e-6, 2e-6) if dtype == torch.float32 else (2e-2, 1e-2)
    for k in ("total_loss", "lm_loss", "rec_loss"):
        assert relf(mine[k], g[t + k]) <= tol_l, k
    rows = mine["recon_image_embeds"].reshape(-1, c["hidden"])
    assert rel(rows, g[t + "recon_rows"]) <= tol_f
    assert float(m0["rec_loss"]) == 0.0 and relf(m0["lm_loss"], g[t + "noimg.lm_loss"]) <= tol_l
    assert relf(m0["total_loss"], g[t + "noimg.total_loss"]) <= tol_l


def test_forward_goldens_are_pinned(gold):
    """The generator asserted the restatement against the reference at BOTH sizes; the report it stored says how close."""
    _, report = gold
    for tag in ("tiny", "full"):
        e = report["%s_f32_oracle_vs_reference" % tag]
        assert max(e.values()) <= 5e-6, e
        assert ("%s_bf16_vs_f32_reference" % tag) in report


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the product classes
# ---------------------------------------------------------------------------------------------------------------------
def _agent(c, dtype):
    from src.models.qwen_visual import Resampler
    from src.models_clm.modeling_llama_xformer import LlamaConfig, LlamaForCausalLM
    from src.models_clm.models import ContinuousLVLM
    import synth
    wd = synth.llama_weights(c["seed"], c["hidden"], c["n_heads"], c["n_layers"], c["inter"], c["vocab"], dtype=dtype)
    cfg = LlamaConfig(hidden_size=c["hidden"], intermediate_size=c["inter"], num_hidden_layers=c["n_layers"],
                      num_attention_heads=c["n_heads"], vocab_size=c["vocab"])
    llm = LlamaForCausalLM(cfg)
    missing, unexpected = llm.load_state_dict(wd, strict=False)
    assert not missing and not unexpected
    llm.cache_cap, llm.max_new, llm.max_prefill_rows = 256, 16, 192
    llm.use_kv_cache_head = False
    H = c["hidden"]
    rin = Resampler(grid_size=c["grid_in"], embed_dim=H, num_heads=c["res_heads"], kv_dim=H)
    rin.load_state_dict(synth.resampler_weights(21, "", c["grid_in"], H, dtype=dtype))
    rout = Resampler(grid_size=c["grid_out"], embed_dim=H, num_heads=c["res_heads"], kv_dim=H)
    rout.load_state_dict(synth.resampler_weights(22, "", c["grid_out"], H, dtype=dtype))
    return ContinuousLVLM(llm, rin, rout, lm_loss_scale=1.0, rec_loss_scale=1.0).eval().to(DEV, dtype)


def _check(out, g, report, tag, dtag, stride, H):
    t = "%s_%s." % (tag, dtag)
    t32 = "%s_f32." % tag
    rows = out["recon_image_embeds"].reshape(-1, H)[::stride]
    if dtag == "f32":
        errs = {k: relf(out[k], g[t + k]) for k in ("total_loss", "lm_loss", "rec_loss")}
        errs["recon"] = rel(rows, g[t + "recon_rows"])
        print("ContinuousLVLM.forward %s fp32: HIP vs REFERENCE %s" % (tag, {k: "%.2e" % v for k, v in errs.items()}))
        assert max(errs.values()) <= 1e-4, errs
    else:
        gap = report["%s_bf16_vs_f32_reference" % tag]
        e_same = rel(rows, g[t + "recon_rows"])
        e_32 = rel(rows, g[t32 + "recon_rows"])
        print("ContinuousLVLM.forward %s bf16: recon HIP vs ref-bf16 %.3e | HIP vs ref-fp32 %.3e | reference bf16 vs fp32 %.3e"
              % (tag, e_same, e_32, gap["recon"]))
        assert e_32 <= 1.5 * gap["recon"] + 2e-3
        for k in ("total_loss", "lm_loss", "rec_loss"):
            e = relf(out[k], g[t32 + k])
            print("   %s: HIP-bf16 %.5f | ref-bf16 %.5f | ref-fp32 %.5f" % (k, float(out[k]), float(g[t + k]), float(g[t32 + k])))
            assert e <= 1.5 * gap[k] + 2e-2, (k, e, gap[k])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,dtag", DTYPES)
def test_continuous_lvlm_forward_tiny(gold, dtype, dtag):
    g, report = gold
    c = MF.TINY
    agent = _agent(c, dtype)
    b = MF.batch(c, dtype, 3)
    out = agent(b["input_ids"], b["attention_mask"], b["labels"], b["image_embeds"], b["embeds_gen_mask"], b["embeds_cmp_mask"],
                b["ids_gen_mask"], b["ids_cmp_mask"], return_recon_image_embeds=True)
    assert set(out) == {"total_loss", "lm_loss", "rec_loss", "recon_image_embeds"}
    _check(out, g, report, "tiny", dtag, 1, c["hidden"])
    # the branch without images (reference :41-47, 58-62, 82-90): placeholder terms are exactly zero
    o0 = agent(b["input_ids"], b["attention_mask"], b["labels"], None, None, None, None, None)
    assert set(o0) == {"total_loss", "lm_loss", "rec_loss"} and float(o0["rec_loss"]) == 0.0
    t = "tiny_%s." % dtag
    assert relf(o0["lm_loss"], g[t + "noimg.lm_loss"]) <= (1e-4 if dtype == torch.float32 else 2e-2)
    # determinism: the losses are fixed-order reductions
    o1 = agent(b["input_ids"], b["attention_mask"], b["labels"], b["image_embeds"], b["embeds_gen_mask"], b["embeds_cmp_mask"],
               b["ids_gen_mask"], b["ids_cmp_mask"])
    assert float(o1["total_loss"]) == float(out["total_loss"])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,dtag", DTYPES)
def test_continuous_lvlm_forward_hidden4096(gold, dtype, dtag):
    """hidden 4096 / 32 heads / inter 11008 / vocab 32066 (2 layers), real-size resamplers, batch of 2 padded sequences."""
    g, report = gold
    c = MF.FULL
    agent = _agent(c, dtype)
    b = MF.batch(c, dtype, 2)
    out = agent(b["input_ids"], b["attention_mask"], b["labels"], b["image_embeds"], b["embeds_gen_mask"], b["embeds_cmp_mask"],
                b["ids_gen_mask"], b["ids_cmp_mask"], return_recon_image_embeds=True)
    _check(out, g, report, "full", dtag, 16, c["hidden"])


@pytest.mark.gpu
def test_loss_heads_vs_torch():
    """The three loss kernels against torch on the host: cross-entropy with ignored rows, cosine loss, MSE; fp32 and bf16
    (bf16 rounds where torch's graph rounds: log_softmax output, normalised operands, products, sums)."""
    import torch.nn.functional as F
    from seedstory import ops
    import synth
    for dtype, tol in ((torch.float32, 2e-6), (torch.bfloat16, 4e-3)):
        lg = synth.normal_like(81, (77, 32066), 2.0, dtype=dtype)
        lab = synth.randint(82, (77,), 0, 32066)
        lab[::5] = -100
        ref = F.cross_entropy(lg.float() if dtype == torch.float32 else lg, lab)
        loss, n = ops.cross_entropy(lg.to(DEV), lab.to(DEV))
        assert int(n) == int((lab != -100).sum()) and relf(loss, ref) <= tol
        a = synth.normal_like(83, (3, 256, 4096), 1.0, dtype=dtype)
        b_ = (a.float() * 0.3 + synth.normal_like(84, (3, 256, 4096), 1.0)).to(dtype)
        assert relf(ops.cosine_loss(a.to(DEV), b_.to(DEV)), O.cosine_loss(a, b_)) <= tol
        x = synth.normal_like(85, (2, 4, 128, 128), 1.0, dtype=dtype)
        y = synth.normal_like(86, (2, 4, 128, 128), 1.0, dtype=dtype)
        assert relf(ops.mse_loss(x.to(DEV), y.to(DEV)), F.mse_loss(x.float(), y.float())) <= 2e-6
    lab = torch.full((5,), -100)
    assert int(ops.cross_entropy(torch.zeros(5, 16, device=DEV), lab.to(DEV))[1]) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sdxl_adapter_forward_tiny(dtype):
    """SDXLAdapter.forward (adapter_modules.py:330-343) on the tiny UNet of tests/test_sdxl_gpu.py and a small ResamplerXLV2
    against the restatements: resampler -> UNet at PER-SAMPLE timesteps -> mse_loss(noise_pred.float(), noise.float()).
    (The UNet side of the oracle restates diffusers, which is absent: parity unpinned at that boundary.)"""
    import torch.nn.functional as F
    import sdxl_oracle as S
    import synth
    import test_sdxl_gpu as TS
    from src.models_ipa.adapter_modules import SDXLAdapter
    from src.models_ipa.resampler import ResamplerXLV2
    m, wd, c = TS._unet(dtype)
    cx = dict(dim=128, depth=2, dim_head=32, heads=4, num_queries=8, embedding_dim=256, output1_dim=48, output2_dim=80, ff_mult=4)
    xwd = synth.resampler_xlv2_weights(41, **cx)
    rs = ResamplerXLV2(**cx)
    missing, unexpected = rs.load_state_dict(xwd, strict=False)
    assert not missing and not unexpected
    adapter = SDXLAdapter.from_pretrained(unet=m, resampler=rs).to(DEV, dtype).eval()
    feat = synth.normal_like(91, (2, 16, 256), 1.0)
    noisy = synth.normal_like(92, (2, 4, 16, 16), 1.0)
    noise = synth.normal_like(93, (2, 4, 16, 16), 1.0)
    ts = torch.tensor([801.0, 333.0])
    tid = torch.tensor([[128, 128, 0, 0, 128, 128]] * 2, dtype=torch.float32)

    def truth(dt_):
        cast = lambda d: {k: v.to(dt_) for k, v in d.items()}  # noqa: E731
        ctx, pooled = O.resampler_xlv2_forward(cast(xwd), feat.to(dt_), depth=cx["depth"], heads=cx["heads"], dim_head=cx["dim_head"])
        pred = S.unet_forward(cast(wd), c, noisy.to(dt_), ts, ctx, pooled, tid)
        return pred, F.mse_loss(pred.float(), noise.to(dt_).float())
    pred32, loss32 = truth(torch.float32)
    out = adapter(noisy.to(DEV, dtype), ts, feat.to(DEV, dtype), None, noise.to(DEV, dtype), tid)
    assert set(out) == {"total_loss", "noise_pred"} and out["noise_pred"].shape == pred32.shape
    if dtype == torch.float32:
        assert rel(out["noise_pred"], pred32) < 2e-4 and relf(out["total_loss"], loss32) < 2e-4
    else:
        predbf, lossbf = truth(torch.bfloat16)
        theirs = rel(predbf, pred32)
        assert rel(out["noise_pred"], pred32) <= 1.5 * theirs + 2e-3
        assert relf(out["total_loss"], loss32) <= 1.5 * relf(lossbf, loss32) + 1e-2
    # a noise tensor of another dtype takes the .float() route of the reference
    out2 = adapter(noisy.to(DEV, dtype), ts, feat.to(DEV, dtype), None, noise.to(DEV, torch.float32), tid)
    assert relf(out2["total_loss"], out["total_loss"]) < 1e-2
