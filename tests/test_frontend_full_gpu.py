"""GPU parity of the front end and the image-feature REGRESSOR at their REAL dimensions, through the
product classes (``src.models.qwen_visual.Resampler`` / ``VisionTransformerWithAttnPool``,
``src.models_ipa.resampler.ResamplerXLV2``, ``src.models_clm.models.ContinuousLVLM.generate``), against

  (1) output rows produced by the REAL reference classes on CPU (``tests/golden/frontend_full.safetensors``, written by
      ``oracle/make_golden_full.py``; weights and inputs regenerate from the seeds of ``oracle/synth.py``), and
  (2) the oracle restatement's whole tensor (pinned on the same reference run, <= 2e-6), computed on the host here.

Shapes: Resampler 4096 / 32 heads as input (2 x 256 -> 64) and output (64 -> 256) resampler; ViT-G ends (patch-embed
3x448^2 -> 1024x1664, bicubic position table 256 -> 1024, one trunk block, attn_pool 256 q x 1024 kv with kv_proj
1664 -> 4096, ln_post, @proj); ResamplerXLV2 (1024, depth 4, 16 x 64, 64 queries, 4096 -> 64 x 2048 + 1280) on
[2, 256, 4096]; ContinuousLVLM.generate at hidden 4096 / 32 heads / inter 11008 / vocab 32066 (2 layers) feeding the
full-size output resampler -> ``img_gen_feat`` [1, 256, 4096] (the north-star quantity: <= 1e-3 relative).

Tolerances.  fp32: 1e-4 relative Frobenius (exact-fp32 MFMA chains; summation order only).  bf16: (i) HIP vs the
reference's own bf16 run, and (ii) HIP-bf16 vs the reference's fp32 run must not exceed 1.5x the reference's OWN
bf16-vs-fp32 distance (+1e-3): a whole bf16 module is a chaotic function of its rounding points."""
import json
import os

import pytest
import torch

import seedstory_oracle as O
import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DTYPES = [(torch.float32, "f32"), (torch.bfloat16, "bf16")]
F32_TOL = [1e-4]      # fp32 tolerance of gate(); the split-bf16 "gate mode" tests below widen it to the north-star 1e-3


class split_mode:
    """`gemm_f32_split` = 1: every GEMM on fp32 tensors runs as three bf16 MFMA products on hi / lo operand halves with fp32
    accumulation (csrc/ss_gemm.hip, SPLIT) instead of the exact 1/16-rate fp32 MFMA chain — the affordable mode that still
    meets the 1e-3 gate on `img_gen_feat`."""

    def __enter__(self):
        from seedstory import _lib
        _lib.set_tuning("gemm_f32_split", 1)
        F32_TOL[0] = 1e-3

    def __exit__(self, *a):
        from seedstory import _lib
        _lib.set_tuning("gemm_f32_split", 0)
        F32_TOL[0] = 1e-4


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def full():
    from safetensors.torch import load_file
    g = load_file(os.path.join(ROOT, "tests", "golden", "frontend_full.safetensors"))
    with open(os.path.join(ROOT, "tests", "golden", "frontend_full.json")) as f:
        meta = json.load(f)
    return g, meta


def gate(name, y, g, tag_base, dtag, stride, oracle_f32=None):
    """y: HIP output (any shape, last dim = channels).  Compares rows [::stride] with the reference's rows, whole-tensor
    statistics with the reference's, and (fp32) the whole tensor with the oracle."""
    flat = y.reshape(-1, y.shape[-1]).float().cpu()
    rows = flat[::stride]
    ref = g["%s_%s.rows" % (tag_base, dtag)].float()
    ref32 = g["%s_f32.rows" % tag_base].float()
    assert rows.shape == ref.shape, (rows.shape, ref.shape)
    r_same = rel(rows, ref)
    nrm = float(flat.norm())
    ref_nrm = float(g["%s_%s.norm" % (tag_base, dtag)])
    if dtag == "f32":
        print("%s fp32: HIP vs REFERENCE rows %.3e | norm %.6g vs %.6g" % (name, r_same, nrm, ref_nrm))
        assert r_same < F32_TOL[0], (name, r_same)
        assert abs(nrm - ref_nrm) <= F32_TOL[0] * ref_nrm
        if oracle_f32 is not None:
            r_full = rel(y, oracle_f32)
            print("%s fp32: HIP vs oracle, whole tensor %.3e" % (name, r_full))
            assert r_full < F32_TOL[0], (name, r_full)
        return r_same
    r_32 = rel(rows, ref32)
    gap = rel(ref, ref32)
    print("%s bf16: HIP vs REFERENCE-bf16 rows %.3e | HIP vs reference-fp32 %.3e | reference bf16 vs fp32 %.3e"
          % (name, r_same, r_32, gap))
    assert r_32 <= 1.5 * gap + 1e-3, (name, r_32, gap)
    assert r_same <= 2.5 * gap + 2e-3, (name, r_same, gap)     # two independent bf16 roundings of one fp32 function
    assert abs(nrm - ref_nrm) <= 2e-2 * ref_nrm
    return r_same


@pytest.mark.parametrize("which", ["res_in", "res_out"])
@pytest.mark.parametrize("dtype,dtag", DTYPES)
def test_resampler_4096_32heads(full, which, dtype, dtag):
    """qwen_visual.py:95-153 at the agent's configuration (agent_7b_sft.yaml): 4096-wide LayerNorm, 32-head MHA,
    4096^2 in/out projections; input (2 images x 256 kv -> 64 q) and output = the regressor (64 kv -> 256 q)."""
    from src.models.qwen_visual import Resampler
    g, meta = full
    c = meta["RES_IN" if which == "res_in" else "RES_OUT"]
    wd = synth.resampler_weights(c["seed"], "", c["grid"], c["embed"], dtype=dtype)
    m = Resampler(grid_size=c["grid"], embed_dim=c["embed"], num_heads=c["heads"], kv_dim=c["embed"])
    missing, unexpected = m.load_state_dict(wd, strict=False)
    assert not missing and not unexpected
    m = m.to(DEV, dtype)
    x = synth.normal_like(c["seed"] + 100, (c["batch"], c["n_kv"], c["embed"]), 1.0, dtype=dtype)
    y = m(x.to(DEV))
    assert y.shape == (c["batch"], c["grid"] ** 2, c["embed"])
    orc = O.resampler_forward(wd, "", x, c["heads"]) if dtype == torch.float32 else None
    gate(which, y, g, which, dtag, c["row_stride"], orc)


@pytest.mark.parametrize("dtype,dtag", DTYPES)
def test_vit_ends_real_size(full, dtype, dtag):
    """qwen_visual.py:376-399 around ONE trunk block at ViT-G size: 588(+pad)->1664 patch GEMM, bicubic 256->1024
    position table, ln_pre, block, attn_pool with kv_proj 1664->4096 (256 q x 1024 kv, 32 heads), ln_post, @proj."""
    from src.models.qwen_visual import VisionTransformerWithAttnPool
    g, meta = full
    c = meta["VIT"]
    wd = synth.vit_weights(c["seed"], c["width"], c["layers"], c["heads"], c["mlp_width"], c["patch"], c["out_dim"],
                           c["n_queries"], dtype=dtype)
    m = VisionTransformerWithAttnPool(image_size=c["image"], patch_size=c["patch"], width=c["width"], layers=c["layers"],
                                      heads=c["heads"], mlp_ratio=c["mlp_ratio"], n_queries=c["n_queries"],
                                      output_dim=c["out_dim"])
    assert m.mlp_width == c["mlp_width"]
    missing, unexpected = m.load_state_dict(wd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    m = m.to(DEV, dtype)
    x = synth.normal_like(c["seed"] + 100, (1, 3, c["image"], c["image"]), 1.0, dtype=dtype)
    y = m(x.to(DEV))
    assert y.shape == (1, c["n_queries"], c["out_dim"])
    orc = None
    if dtype == torch.float32:
        orc = O.vit_forward(wd, x, width=c["width"], layers=c["layers"], heads=c["heads"], patch=c["patch"],
                            out_dim=c["out_dim"], n_queries=c["n_queries"])
    gate("vit ends", y, g, "vit", dtag, c["row_stride"], orc)


@pytest.mark.parametrize("dtype,dtag", DTYPES)
def test_resampler_xlv2_real_config(full, dtype, dtag):
    """src/models_ipa/resampler.py:228-284 at the de-tokenizer's configuration (detokenizer_sdxl_qwen_vit_adapted.yaml)."""
    from src.models_ipa.resampler import ResamplerXLV2
    g, meta = full
    c, r = meta["XLV2"], meta["XLV2_RUN"]
    wd = synth.resampler_xlv2_weights(r["seed"], dtype=dtype, **c)
    m = ResamplerXLV2(**c)
    missing, unexpected = m.load_state_dict(wd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    m = m.to(DEV, dtype)
    x = synth.normal_like(r["seed"] + 100, (r["batch"], r["tokens"], c["embedding_dim"]), 1.0, dtype=dtype)
    ctx, pooled = m(x.to(DEV))
    assert ctx.shape == (r["batch"], c["num_queries"], c["output1_dim"] + c["output2_dim"])
    assert pooled.shape == (r["batch"], c["output2_dim"])
    oc = op = None
    if dtype == torch.float32:
        oc, op = O.resampler_xlv2_forward(wd, x, depth=c["depth"], heads=c["heads"], dim_head=c["dim_head"])
    gate("xlv2 ctx", ctx, g, "xlv2_ctx", dtag, r["row_stride"], oc)
    gate("xlv2 pooled", pooled, g, "xlv2_pooled", dtag, 1, op)


class _Tok:
    def __init__(self, ids):
        self.ids = ids

    def encode(self, s, add_special_tokens=False):
        if s == "<img>":
            return [self.ids[0]]
        if s == "</img>":
            return [self.ids[-1]]
        return list(self.ids)

    def decode(self, ids, skip_special_tokens=False):
        return " ".join(str(int(i)) for i in ids)


@pytest.mark.parametrize("dtype,dtag", DTYPES)
def test_generate_hidden4096_full_size_regressor(full, dtype, dtag):
    """models.py:98-221 at LLaMA-7B width (2 layers): real-size input resampler -> splice -> prefill + greedy decode with
    the image-token processor -> 64 last-layer rows in front of </img> -> FULL-SIZE output resampler.  Prints the
    north-star number (img_gen_feat relative error) in fp32 and bf16."""
    from src.models.qwen_visual import Resampler
    from src.models_clm.modeling_llama_xformer import LlamaConfig, LlamaForCausalLM
    from src.models_clm.models import ContinuousLVLM
    g, meta = full
    d, gen = meta["LLAMA"], meta["GEN"]
    lo, hi = meta["IMG_IDS"]
    img_ids = list(range(lo, hi + 1))
    E = d["hidden"]
    wd = synth.llama_weights(gen["seed"], d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], dtype=dtype)
    cfg = LlamaConfig(hidden_size=d["hidden"], intermediate_size=d["inter"], num_hidden_layers=d["n_layers"],
                      num_attention_heads=d["n_heads"], vocab_size=d["vocab"])
    llm = LlamaForCausalLM(cfg)
    missing, unexpected = llm.load_state_dict(wd, strict=False)
    assert not missing and not unexpected
    llm.cache_cap, llm.max_new, llm.max_prefill_rows = 256, 128, 128
    llm.use_kv_cache_head = False
    rin = Resampler(grid_size=meta["RES_IN"]["grid"], embed_dim=E, num_heads=32, kv_dim=E)
    rin.load_state_dict(synth.resampler_weights(meta["RES_IN"]["seed"], "", meta["RES_IN"]["grid"], E, dtype=dtype))
    rout = Resampler(grid_size=meta["RES_OUT"]["grid"], embed_dim=E, num_heads=32, kv_dim=E)
    rout.load_state_dict(synth.resampler_weights(meta["RES_OUT"]["seed"], "", meta["RES_OUT"]["grid"], E, dtype=dtype))
    agent = ContinuousLVLM(llm, rin, rout).eval().to(DEV, dtype)
    del wd
    # inputs of oracle/make_golden_full.py::gen_inputs
    boi, eoi = img_ids[0], img_ids[-1]
    n_text = gen["n_text"]
    prompt = [1] + synth.randint(50, (n_text,), 3, 32000).tolist() + [boi] + img_ids[1:65] + [eoi]
    input_ids = torch.tensor([prompt])
    mask = torch.zeros_like(input_ids, dtype=torch.bool)
    mask[0, n_text + 2:n_text + 2 + 64] = True
    image_embeds = synth.normal_like(51, (1, 256, E), 1.0, dtype=dtype)
    forced = synth.randint(52, (6,), 3, 32000).tolist() + [boi]
    if dtype != torch.float32:
        forced = forced + img_ids[1:] + [2]
    out = agent.generate(tokenizer=_Tok(img_ids), input_ids=input_ids, image_embeds=image_embeds.to(DEV),
                         embeds_cmp_mask=torch.tensor([True]), ids_cmp_mask=mask, max_new_tokens=gen["max_new"],
                         num_img_gen_tokens=64, forced_tokens=forced)
    ref_ids = g["gen_%s.generate_ids" % dtag].tolist()
    assert out["generate_ids"].tolist() == ref_ids
    assert out["has_img_output"] and out["num_gen_imgs"] == 1
    feat = out["img_gen_feat"]
    assert feat.shape == (1, 256, E)
    # the fixture names are gen_<dtag>.img_gen_feat.rows: go through gate() with an explicit tag mapping
    gg = {("nf_%s" % t) + k[len("gen_%s.img_gen_feat" % t):]: v for t in ("f32", "bf16") for k, v in g.items()
          if k.startswith("gen_%s.img_gen_feat" % t)}
    r = gate("img_gen_feat (north-star quantity)", feat, gg, "nf", dtag, gen["feat_stride"])
    if dtype == torch.float32:
        assert r < 1e-3          # the north-star gate itself (measured ~1e-6)
    else:
        # mixed mode: same bf16 decoder stack, regressor in exact fp32 on the 64 bf16 rows
        agent.enable_fp32_regressor(True)
        out2 = agent.generate(tokenizer=_Tok(img_ids), input_ids=input_ids, image_embeds=image_embeds.to(DEV),
                              embeds_cmp_mask=torch.tensor([True]), ids_cmp_mask=mask, max_new_tokens=gen["max_new"],
                              num_img_gen_tokens=64, forced_tokens=forced)
        agent.enable_fp32_regressor(False)
        assert out2["generate_ids"].tolist() == ref_ids
        st = gen["feat_stride"]
        ref32 = gg["nf_f32.rows"].float()
        d_plain = rel(feat.reshape(-1, E)[::st], ref32)
        d_mixed = rel(out2["img_gen_feat"].reshape(-1, E)[::st], ref32)
        print("img_gen_feat bf16 vs the reference's fp32 rows: bf16 regressor %.3e | fp32 regressor on the bf16 rows %.3e"
              % (d_plain, d_mixed))
        assert d_mixed <= d_plain * 1.05 + 1e-3


# ---- the gate mode: fp32 tensors, split-bf16 MFMA products (gemm_f32_split) — must meet the north-star 1e-3 vs the REAL rows ----

@pytest.mark.parametrize("which", ["res_in", "res_out"])
def test_gate_mode_resampler_4096_32heads(full, which):
    with split_mode():
        test_resampler_4096_32heads(full, which, torch.float32, "f32")


def test_gate_mode_vit_ends_and_xlv2(full):
    with split_mode():
        test_vit_ends_real_size(full, torch.float32, "f32")
        test_resampler_xlv2_real_config(full, torch.float32, "f32")


def test_gate_mode_generate_hidden4096_img_gen_feat(full):
    """The north-star quantity itself in the affordable gate mode: LLaMA-7B-width decoder layers (prefill through the split GEMM,
    decode through the fp32-weight GEMV), full-size regressor — img_gen_feat <= 1e-3 against the REAL reference's fp32 rows and
    the same generated ids."""
    with split_mode():
        test_generate_hidden4096_full_size_regressor(full, torch.float32, "f32")
