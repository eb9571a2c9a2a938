"""fp16 — the dtype the reference scripts actually run (gen_george.py:19-20,57,62-67) — through every layer of the HIP path.

Truth = rows produced by the REAL reference modules in fp16 on CPU (``oracle/make_golden_fp16.py`` ->
``tests/golden/hotpath_tiny_fp16.safetensors``) next to their fp32 rows (``hotpath_tiny.safetensors``).  Gates, written per test:
(i) distance to the reference's fp16 rows, (ii) distance to the reference's fp32 rows <= 1.5 x the reference's OWN fp16-vs-fp32
distance + eps (eps = 3e-4: fp16 carries 11 significant bits, products accumulate in fp32 on both sides).  The kernel-level tests
(GEMV, GEMM tiles incl. the ping-pong ones, conv3x3, attention) compare with torch fp32 on the fp16-rounded operands.  The SDXL half
is compared with ``oracle/sdxl_oracle.py`` run in fp16 on CPU (that oracle is parity-unpinned, as for bf16); the VAE test exercises the
``force_upcast`` policy end to end: an fp16 module decodes in fp32 like diffusers, never in fp16."""
import math
import os

import pytest
import torch
import torch.nn.functional as F
from safetensors.torch import load_file

import sdxl_oracle as S
import seedstory_oracle as O
import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H = torch.float16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPS = 3e-4


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def g16():
    return load_file(os.path.join(ROOT, "tests", "golden", "hotpath_tiny_fp16.safetensors"))


def gate(y, ref16, ref32, what, to16=5e-3):
    d16, d32, own = rel(y, ref16), rel(y, ref32), rel(ref16, ref32)
    print("%s fp16: HIP vs ref-fp16 %.2e | HIP vs ref-fp32 %.2e | ref fp16 vs fp32 %.2e" % (what, d16, d32, own))
    assert d16 < to16, (what, d16)
    assert d32 <= 1.5 * own + EPS, (what, d32, own)


# ---- kernels ---------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("nb", [1, 2, 4, 8])
@pytest.mark.parametrize("N,K", [(4096, 4096), (4096, 11008), (1000, 1664), (32066, 4096)])
def test_gemv_fp16(nb, N, K):
    from seedstory import ops
    w = synth.normal_like(301, (N, K), 0.02, dtype=H)
    x = synth.normal_like(302, (nb, K), 1.0, dtype=H)
    b = synth.normal_like(303, (N,), 0.1, dtype=H)
    ref = x.float() @ w.float().t() + b.float()
    fn = ops.gemv if nb == 1 else ops.gemv_batched
    y = fn(w.to(DEV), (x[0] if nb == 1 else x).to(DEV), bias=b.to(DEV))
    assert rel(y.reshape(nb, N), ref) < 6e-4          # output rounding 2^-11 / sqrt(3) + accumulation order


@pytest.mark.parametrize("nb", [1, 4, 8])
def test_gemv_fused_rmsnorm_and_silu_fp16(nb):
    from seedstory import ops
    K, I = 4096, 11008
    nw = synth.normal_like(311, (K,), 0.1, 1.0, dtype=H)
    w = synth.normal_like(312, (2 * I, K), 0.02, dtype=H)
    x = synth.normal_like(313, (nb, K), 1.0, dtype=H)
    xn = O.rmsnorm(x, nw, 1e-5).float()                 # rounds to fp16 where LlamaRMSNorm does
    gu = (xn @ w.float().t()).to(H).float()
    ref = (F.silu(gu[:, :I]).to(H).float() * gu[:, I:]).to(H)
    fn = ops.gemv if nb == 1 else ops.gemv_batched
    y = fn(w.to(DEV), (x[0] if nb == 1 else x).to(DEV), norm_w=nw.to(DEV), eps=1e-5, silu_mul=True)
    assert rel(y.reshape(nb, I), ref) < 2e-3


@pytest.mark.parametrize("cfg", [0, 54, 55, 56, 57, 60, 62, 65])
@pytest.mark.parametrize("M,N,K", [(512, 640, 1280), (1024, 1280, 640), (333, 1000, 192), (2048, 2560, 320)])
def test_gemm_tiles_fp16(cfg, M, N, K):
    """Every tile family instantiated for fp16 (table / closed-form choice = cfg 0, the ping-pong tiles 54-57, one-barrier tiles)."""
    from seedstory import _lib, ops
    a = synth.normal_like(321, (M, K), 1.0, dtype=H)
    w = synth.normal_like(322, (N, K), 0.05, dtype=H)
    b = synth.normal_like(323, (N,), 0.5, dtype=H)
    r = synth.normal_like(324, (M, N), 1.0, dtype=H)
    ref = (a.float() @ w.float().t() + b.float()).to(H).float() + r.float()
    _lib.set_tuning("gemm_cfg", cfg)
    try:
        y = ops.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), residual=r.to(DEV))
    finally:
        _lib.set_tuning("gemm_cfg", 0)
    assert rel(y, ref) < 6e-4


def test_gemm_geglu_and_gelu_fp16():
    from seedstory import ops
    M, K, N = 1024, 640, 2560
    a = synth.normal_like(331, (M, K), 1.0, dtype=H)
    w = synth.normal_like(332, (2 * N, K), 0.05, dtype=H)
    b = synth.normal_like(333, (2 * N,), 0.2, dtype=H)
    h = (a.float() @ w.float().t() + b.float()).to(H)
    ref = (h[:, :N].float() * F.gelu(h[:, N:].float()).to(H).float()).to(H)       # diffusers GEGLU: value * gelu(gate)
    wp = torch.stack([w[:N], w[N:]], 1).reshape(2 * N, K).contiguous()            # (value_i, gate_i) interleaved rows
    bp = torch.stack([b[:N], b[N:]], 1).reshape(2 * N).contiguous()
    y = ops.gemm_geglu(a.to(DEV), wp.to(DEV), bp.to(DEV))
    assert rel(y, ref) < 1.5e-3
    y2 = ops.gemm(a.to(DEV), w[:N].contiguous().to(DEV), bias=b[:N].contiguous().to(DEV), gelu=True)
    assert rel(y2, F.gelu(h[:, :N].float())) < 1e-3


@pytest.mark.parametrize("B,Ci,Co,Hh,Ww,stride,up", [(2, 64, 320, 16, 16, 1, False), (1, 320, 640, 32, 32, 1, False), (2, 320, 64, 8, 8, 2, False),
                                                     (1, 32, 40, 6, 5, 1, True), (4, 128, 256, 16, 16, 1, False)])
def test_conv3x3_fp16(B, Ci, Co, Hh, Ww, stride, up):
    from seedstory import ops
    from seedstory.diffusion import _conv_w
    x = synth.normal_like(341, (B, Ci, Hh, Ww), 1.0, dtype=H)
    w = synth.normal_like(342, (Co, Ci, 3, 3), 1.0 / math.sqrt(9 * Ci), dtype=H)
    b = synth.normal_like(343, (Co,), 0.5, dtype=H)
    xi = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    ref = F.conv2d(xi, w.float(), b.float(), stride=stride, padding=1)
    xn = x.permute(0, 2, 3, 1).reshape(B * Hh * Ww, Ci).contiguous()
    y, ho, wo = ops.conv3x3(xn.to(DEV), _conv_w(w).to(DEV), B, Hh, Ww, stride=stride, upsample=up, bias=b.to(DEV))
    assert (ho, wo) == tuple(ref.shape[2:])
    assert rel(y.reshape(B, ho, wo, Co).permute(0, 3, 1, 2), ref) < 6e-4


@pytest.mark.parametrize("B,heads,hd,Lq,Lk,causal", [(1, 32, 128, 343, 343, True), (1, 2, 128, 9, 46, True), (2, 10, 64, 1024, 1024, False),
                                                     (2, 20, 64, 256, 64, False), (1, 16, 104, 1024, 1024, False)])
def test_attention_fp16(B, heads, hd, Lq, Lk, causal):
    from seedstory import ops
    E = heads * hd
    q = synth.normal_like(351, (B, Lq, E), 1.0, dtype=H)
    k = synth.normal_like(352, (B, Lk, E), 1.0, dtype=H)
    v = synth.normal_like(353, (B, Lk, E), 1.0, dtype=H)
    sp = lambda t, L: t.float().view(B, L, heads, hd).transpose(1, 2)     # noqa: E731
    s = sp(q, Lq) @ sp(k, Lk).transpose(-1, -2) / math.sqrt(hd)
    if causal:
        i = torch.arange(Lq)[:, None]
        j = torch.arange(Lk)[None, :]
        s = s.masked_fill(j > i + (Lk - Lq), float("-inf"))
    ref = (torch.softmax(s, -1) @ sp(v, Lk)).transpose(1, 2).reshape(B, Lq, E)
    y = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), heads, causal_br=causal)
    assert rel(y, ref) < 1.5e-3          # P is rounded to fp16 for the PV product (bf16: 6e-3)


# ---- MLLM half: engines and the reference-API mirror vs the REAL reference's fp16 rows ---------------------------------------

def _img_ids(meta):
    lo, hi = meta["IMG_IDS"]
    return list(range(lo, hi + 1))


def test_llama_engine_fp16(golden, g16):
    from seedstory.llama import LlamaEngine
    g, meta = golden
    d = meta["LLAMA"]
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], dtype=H)
    eng = LlamaEngine(wd, hidden=d["hidden"], n_heads=d["n_heads"], n_layers=d["n_layers"], inter=d["inter"], vocab=d["vocab"], dtype=H,
                      device=DEV, cache_cap=256, max_new=128, max_prefill_rows=64, img_ids=_img_ids(meta))
    emb = wd["model.embed_tokens.weight"]
    t16, t32 = "llama_f16", "llama_f32"
    assert torch.equal(g16[t16 + ".ids"], g[t32 + ".ids"])
    hid = eng.prefill(emb[g16[t16 + ".ids"][0]], want_hidden=True)
    gate(hid, g16[t16 + ".prefill_hidden"][0], g[t32 + ".prefill_hidden"][0], "prefill hidden")
    gate(eng.logits, g16[t16 + ".prefill_logits"][0, -1], g[t32 + ".prefill_logits"][0, -1], "prefill logits")
    pkv = eng.past_key_values()
    gate(pkv[0][0], g16[t16 + ".prefill_k0"], g[t32 + ".prefill_k0"], "prefill k0")
    hid2 = eng.prefill(emb[g16[t16 + ".ids2"][0]], want_hidden=True)
    gate(hid2, g16[t16 + ".cont_hidden"][0], g[t32 + ".cont_hidden"][0], "continuation hidden")
    gate(eng.logits, g16[t16 + ".cont_logits"][0, -1], g[t32 + ".cont_logits"][0, -1], "continuation logits")
    tok = int(g16[t16 + ".ids3"][0, 0])
    n = eng.generate(2, last_prompt_id=5, forced=[tok, 3])
    assert n == 2 and eng.gen_ids[:2].tolist() == [tok, 3]
    gate(eng.hidden_rows[0], g16[t16 + ".decode_hidden"][0, 0], g[t32 + ".decode_hidden"][0, 0], "graph decode hidden")


@pytest.mark.parametrize("n_seq", [4, 8])
def test_llama_slots_fp16_equal_single(golden, n_seq):
    """The lock-step decode (MFMA-form GEMV at 3+ slots) in fp16 reproduces the batch-1 engine slot by slot."""
    from seedstory.llama import LlamaEngine
    g, meta = golden
    d = meta["LLAMA"]
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], dtype=H)
    kw = dict(hidden=d["hidden"], n_heads=d["n_heads"], n_layers=d["n_layers"], inter=d["inter"], vocab=d["vocab"], dtype=H, device=DEV,
              cache_cap=256, max_new=64, max_prefill_rows=64, img_ids=_img_ids(meta))
    emb = wd["model.embed_tokens.weight"]
    prompts = [synth.randint(40 + b, (17 + 3 * b,), 3, 250) for b in range(n_seq)]
    forced = [[7 + b, 9, 11 + b] for b in range(n_seq)]
    single = []
    for b in range(n_seq):
        e1 = LlamaEngine(wd, **kw)
        e1.prefill(emb[prompts[b]])
        n = e1.generate(10, last_prompt_id=int(prompts[b][-1]), forced=forced[b])
        single.append((e1.gen_ids[:n].tolist(), e1.hidden_rows[:n - 1].clone()))
        del e1
    eng = LlamaEngine(wd, n_seq=n_seq, **kw)
    for b in range(n_seq):
        eng.select(b).prefill(emb[prompts[b]])
    ns = eng.generate_batch(10, [int(p[-1]) for p in prompts], forced=forced)
    for b in range(n_seq):
        eng.select(b)
        ids = eng.gen_ids[:ns[b]].tolist()
        assert ids[:3] == single[b][0][:3], b
        if ids == single[b][0]:          # a free-running tail may flip on a near-tie between the dot-product and the MFMA form
            assert rel(eng.hidden_rows[:ns[b] - 1], single[b][1]) < 5e-3, b


def test_resamplers_vit_xlv2_fp16(golden, g16):
    from src.models.qwen_visual import Resampler, VisionTransformerWithAttnPool
    from src.models_ipa.resampler import ResamplerXLV2
    g, meta = golden
    for tag, key, seed in (("res_in", "RES_IN", 21), ("res_out", "RES_OUT", 22)):
        c = meta[key]
        m = Resampler(grid_size=c["grid"], embed_dim=c["embed"], num_heads=c["heads"], kv_dim=c["embed"])
        m.load_state_dict(synth.resampler_weights(seed, "", c["grid"], c["embed"]), strict=False)
        y = m.to(DEV, H)(g[tag + ".x"].to(DEV, H))
        assert y.dtype == H
        gate(y, g16[tag + "_f16.y"], g[tag + ".y"], tag)
    c = meta["VIT"]
    m = VisionTransformerWithAttnPool(image_size=c["image"], patch_size=c["patch"], width=c["width"], layers=c["layers"], heads=c["heads"],
                                      mlp_ratio=c["mlp_width"] / c["width"], n_queries=c["n_queries"], output_dim=c["out_dim"])
    m.load_state_dict(synth.vit_weights(31, c["width"], c["layers"], c["heads"], c["mlp_width"], c["patch"], c["out_dim"], c["n_queries"]), strict=False)
    y = m.to(DEV, H)(g["vit.x"].to(DEV))
    gate(y, g16["vit_f16.y"], g["vit.y"], "ViT + attention pool")
    c = meta["XLV2"]
    m = ResamplerXLV2(**c)
    m.load_state_dict(synth.resampler_xlv2_weights(41, **c), strict=False)
    ctx, pooled = m.to(DEV, H)(g["xlv2.x"].to(DEV, H))
    gate(ctx, g16["xlv2_f16.ctx"], g["xlv2.ctx"], "ResamplerXLV2 ctx")
    gate(pooled, g16["xlv2_f16.pooled"], g["xlv2.pooled"], "ResamplerXLV2 pooled")


def test_vit_block_full_width_fp16(golden, g16):
    """One block at ViT-G width (1664 / 16 heads x 104 / MLP 8192, 1024 tokens) through ss_vit_blocks vs the REAL VisualAttentionBlock's
    fp16 rows."""
    import ctypes as C
    from seedstory import ops
    from seedstory._lib import check, lib
    from src.models.qwen_visual import VisionTransformerWithAttnPool
    g, meta = golden
    c = meta["VITBLK"]
    wd = synth.vit_block_weights(61, c["width"], c["mlp_width"], dtype=H)
    vit = VisionTransformerWithAttnPool(image_size=448, patch_size=14, width=c["width"], layers=1, heads=c["heads"],
                                        mlp_ratio=c["mlp_width"] / c["width"], output_dim=256)
    blk = vit.transformer.resblocks[0]
    missing, unexpected = blk.load_state_dict(wd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    for p in vit.parameters():               # the other (unused here) parameters are uninitialised storage
        if not torch.isfinite(p.data).all():
            p.data.zero_()
    vit = vit.to(DEV, H)
    w, _keep = vit._weights()
    xd = synth.normal_like(161, (c["tokens"], 1, c["width"]), 1.0, dtype=H).transpose(0, 1).contiguous().to(DEV)   # [1, L, W]
    code = ops.dt(xd)
    nbytes = lib().ss_vit_workspace_bytes(C.byref(w), 1, code)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    check(lib().ss_vit_blocks(C.byref(w), xd.data_ptr(), 1, c["tokens"], 0, 1, ws.data_ptr(), nbytes, code, ops.stream()), "ss_vit_blocks")
    gate(xd[0][::c["row_stride"]], g16["vitblk_f16.y_rows"], g["vitblk_f32.y_rows"], "ViT-G block")


class _Tok:
    def __init__(self, ids):
        self.ids = ids

    def encode(self, s, add_special_tokens=False):
        return [self.ids[0]] if s == "<img>" else [self.ids[-1]] if s == "</img>" else list(self.ids)

    def decode(self, ids, skip_special_tokens=False):
        return " ".join(str(int(i)) for i in ids)


def test_continuous_lvlm_generate_fp16(golden, g16):
    """``ContinuousLVLM.generate`` with fp16 modules (agent.to(fp16), gen_george.py:57): ids equal the reference's fp16 run on the forced
    schedule, ``img_gen_feat`` inside the reference's own fp16 distance."""
    from src.models.qwen_visual import Resampler
    from src.models_clm.modeling_llama_xformer import LlamaConfig, LlamaForCausalLM
    from src.models_clm.models import ContinuousLVLM
    g, meta = golden
    d = meta["LLAMA"]
    cfg = LlamaConfig(hidden_size=d["hidden"], intermediate_size=d["inter"], num_hidden_layers=d["n_layers"], num_attention_heads=d["n_heads"],
                      vocab_size=d["vocab"])
    llm = LlamaForCausalLM(cfg)
    llm.load_state_dict(synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], dtype=H), strict=False)
    llm.cache_cap, llm.max_new, llm.max_prefill_rows = 256, 128, 64
    llm.use_kv_cache_head = False
    rin = Resampler(grid_size=meta["RES_IN"]["grid"], embed_dim=256, num_heads=2, kv_dim=256)
    rin.load_state_dict(synth.resampler_weights(21, "", meta["RES_IN"]["grid"], 256, dtype=H))
    rout = Resampler(grid_size=meta["RES_OUT"]["grid"], embed_dim=256, num_heads=2, kv_dim=256)
    rout.load_state_dict(synth.resampler_weights(22, "", meta["RES_OUT"]["grid"], 256, dtype=H))
    agent = ContinuousLVLM(llm, rin, rout).eval().to(DEV, H)
    input_ids = g16["gen_f16.input_ids"]
    n_in = meta["RES_IN"]["grid"] ** 2
    mask = torch.zeros_like(input_ids, dtype=torch.bool)
    mask[0, 14:14 + n_in] = True
    out = agent.generate(tokenizer=_Tok(_img_ids(meta)), input_ids=input_ids, image_embeds=g16["gen_f16.image_embeds"].to(DEV, H),
                         embeds_cmp_mask=torch.tensor([True]), ids_cmp_mask=mask, max_new_tokens=90, num_img_gen_tokens=64,
                         forced_tokens=g16["gen_f16.forced"].tolist())
    assert out["generate_ids"].tolist() == g16["gen_f16.generate_ids"].tolist()
    assert out["img_gen_feat"].dtype == H
    # the fp32 fixture forces a shorter schedule (free tail), so the fp32 truth of THIS schedule is the oracle's fp32 run
    wd32 = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    wd32.update(synth.resampler_weights(21, "input_resampler.", meta["RES_IN"]["grid"], 256))
    wd32.update(synth.resampler_weights(22, "output_resampler.", meta["RES_OUT"]["grid"], 256))
    dims = O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    r32 = O.lvlm_generate(wd32, dims, input_ids, g16["gen_f16.image_embeds"].float(), torch.tensor([True]), mask, _img_ids(meta),
                          max_new_tokens=90, forced=g16["gen_f16.forced"].tolist(), n_heads_resampler=2)
    gate(out["img_gen_feat"], g16["gen_f16.img_gen_feat"], r32["img_gen_feat"], "img_gen_feat")


# ---- SDXL half (oracle: sdxl_oracle, parity-unpinned) in fp16 -------------------------------------------------------------

def test_unet_forward_tiny_fp16():
    from seedstory.diffusion import UNet2DConditionModel
    c = S.TINY_UNET
    wd = S.synth_weights(S.unet_shapes(c), 1)
    m = UNet2DConditionModel(c)
    m.load_state_dict(wd, strict=False)
    m = m.to(DEV, H)
    x = synth.normal_like(5, (2, 4, 16, 16), 1.0)
    ctx = synth.normal_like(6, (2, 8, 128), 1.0)
    pooled = synth.normal_like(7, (2, 80), 1.0)
    tid = torch.tensor([[128, 128, 0, 0, 128, 128]] * 2, dtype=torch.float32)
    ref = S.unet_forward(wd, c, x, torch.tensor(801.0), ctx, pooled, tid)
    ref16 = S.unet_forward({k: v.to(H) for k, v in wd.items()}, c, x.to(H), torch.tensor(801.0), ctx.to(H), pooled.to(H), tid)
    y = m(x.to(DEV, H), 801.0, ctx.to(DEV, H), added_cond_kwargs={"text_embeds": pooled.to(DEV, H), "time_ids": tid}).sample
    assert y.dtype == H
    gate(y, ref16, ref, "tiny UNet", to16=8e-3)


@pytest.fixture(scope="module")
def sdxl_unet_fp16():
    from seedstory.diffusion import UNet2DConditionModel
    torch.cuda.manual_seed(13)
    m = UNet2DConditionModel().to(DEV, H).init_synthetic(13)
    gen = torch.Generator(device=DEV).manual_seed(5)
    for n, p in m.named_parameters():        # non-trivial norm affine parameters and biases
        if p.dim() == 1:
            p.data.copy_(((1.0 if n.endswith("weight") else 0.0) + 0.1 * torch.randn(p.shape, device=DEV, generator=gen)).to(p.dtype))
    m._prep = None
    return m


def _nhwc(x):
    B, C, Hh, Ww = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * Hh * Ww, C).contiguous()


def _nchw(y, B, Hh, Ww):
    return y.reshape(B, Hh, Ww, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("cfg", [0, 56])
def test_sdxl_transformer_block_full_size_fp16(sdxl_unet_fp16, cfg):
    """Transformer2DModel with one BasicTransformerBlock of the SDXL-base network (2.57 B parameters held in fp16) at 1280 channels,
    20 heads x 64, 1024 tokens, context 64 x 2048: q|k|v, ff1 (GEGLU), ff2 and the N = 1280 projections — with the table's tiles and with
    the 256x320 ping-pong tile forced (cfg 56; ineligible shapes take their fallback) — vs the oracle in fp16 and fp32."""
    from seedstory import _lib
    m = sdxl_unet_fp16
    P = m._prepare()
    name, ch, heads, res, B, G = "mid_block.attentions.0", 1280, 20, 32, 2, 32
    wd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()
          if k.startswith(name + ".") and (".transformer_blocks." not in k or ".transformer_blocks.0." in k)}
    x = synth.normal_like(41, (B, ch, res, res), 1.0)
    ctx = synth.normal_like(42, (B, 64, 2048), 1.0)
    ref32 = S.transformer_2d(wd, name, x, ctx, heads, 1, G)
    ref16 = S.transformer_2d({k: v.to(H) for k, v in wd.items()}, name, x.to(H), ctx.to(H), heads, 1, G)
    m._ctx_kv = {}
    _lib.set_tuning("gemm_cfg", cfg)
    try:
        y = m._transformer(P, name, _nhwc(x).to(DEV, H), B, res * res, ctx.to(DEV, H).contiguous().view(B * 64, -1), 64, heads, 1, G)
    finally:
        _lib.set_tuning("gemm_cfg", 0)
        m._ctx_kv = {}
    gate(_nchw(y, B, res, res), ref16, ref32, "SDXL transformer block (cfg %d)" % cfg, to16=3e-3)


@pytest.mark.parametrize("cfg", [0, 56])
def test_sdxl_resblock_full_size_fp16(sdxl_unet_fp16, cfg):
    """ResnetBlock2D 640 -> 1280 at 32 x 32 (GroupNorm + SiLU, two 3x3 convolutions with the projected time embedding, 1x1 shortcut) in
    fp16; cfg 56 routes the convolutions through the ping-pong implicit-GEMM kernel."""
    from seedstory import _lib, ops
    m = sdxl_unet_fp16
    P = m._prepare()
    name, cin, cout, res, B, G = "down_blocks.2.resnets.0", 640, 1280, 32, 4, 32
    wd = {k: v.detach().float().cpu() for k, v in m.state_dict().items() if k.startswith(name + ".")}
    x = synth.normal_like(31, (B, cin, res, res), 1.0)
    emb = synth.normal_like(32, (B, 1280), 1.0)
    ref32 = S.resnet_block(wd, name, x, emb, G)
    ref16 = S.resnet_block({k: v.to(H) for k, v in wd.items()}, name, x.to(H), emb.to(H), G)
    _lib.set_tuning("gemm_cfg", cfg)
    try:
        temb_act = ops.gemm(ops.silu(emb.to(DEV, H)), P["temb_all.weight"], bias=P["temb_all.bias"])
        y = m._resnet(P, name, _nhwc(x).to(DEV, H), B, res, res, temb_act, G)
    finally:
        _lib.set_tuning("gemm_cfg", 0)
    gate(_nchw(y, B, res, res), ref16, ref32, "SDXL ResBlock (cfg %d)" % cfg, to16=3e-3)


def test_vae_fp16_module_decodes_in_fp32_force_upcast():
    """diffusers up-casts an fp16 VAE to fp32 for the decode (``force_upcast``; the reference loads the VAE in fp16 at gen_george.py:62):
    the product's fp16 module must produce the FP32-arithmetic image — compared with the fp32 oracle at fp32 tolerance, which an fp16
    decode (1e-3 class) cannot meet — and the `vae_bf16` knob opts into the faster 16-bit decode."""
    from seedstory import _lib
    from seedstory.diffusion import AutoencoderKL
    c = S.TINY_VAE
    wd = S.synth_weights(S.vae_decoder_shapes(c), 2)
    m = AutoencoderKL(c)
    m.load_state_dict(wd, strict=False)
    m = m.to(DEV, H)
    lat = synth.normal_like(20, (1, 4, 12, 12), 1.0)
    wd16 = {k: v.to(H).float() for k, v in wd.items()}          # the module's weights ARE fp16 values; the arithmetic is fp32
    ref = S.vae_decode(wd16, c, lat.to(H).float())
    y = m.decode((lat / c["scaling_factor"]).to(DEV, H)).sample
    r = rel(y, ref)
    print("fp16 VAE module, force_upcast decode vs fp32 oracle on the fp16-valued weights: %.2e" % r)
    assert r < 6e-4, r          # fp32 arithmetic; the returned sample is rounded to the module dtype once (2^-11 / sqrt(3) = 2.8e-4)
    _lib.set_tuning("vae_bf16", 1)
    try:
        yb = m.decode((lat / c["scaling_factor"]).to(DEV, H)).sample
    finally:
        _lib.set_tuning("vae_bf16", 0)
    rb = rel(yb, ref)
    assert 1e-3 < rb < 3e-2, rb   # the opt-in 16-bit decode is visibly a different arithmetic
