"""TEST INFRASTRUCTURE ONLY — FULL-DIMENSION goldens of the front end and the image-feature regressor, produced by
the REAL reference modules.

Run in the build container (needs ``/root/reference``; about 10 minutes and 12 GB of RAM on 8 cores):

    python oracle/make_golden_full.py

What runs (every module below is the reference's own class, imported behind ``oracle/ref_shims.py`` and loaded with
the seeded synthetic weights of ``oracle/synth.py``; both fp32 and bf16):

  res_in   ``Resampler(grid 8, embed 4096, 32 heads, kv_dim 4096)``  — the agent's input resampler, 2 images,
           256 ViT tokens -> 64 LLM tokens                    (src/models/qwen_visual.py:95-153, agent_7b_sft.yaml)
  res_out  ``Resampler(grid 16, 4096, 32, 4096)`` — the image-feature REGRESSOR, 64 hidden states -> 256 x 4096
  vit      ``VisionTransformerWithAttnPool(448, 14, 1664, layers=1, 16 heads, mlp 8192, 256 queries, 4096)`` — the
           ViT-G ENDS at real size around one trunk block: patch-embed 3x448^2 -> 1024x1664, bicubic position table
           256 -> 1024, ln_pre, attn_pool (256 q x 1024 kv, kv_proj 1664 -> 4096), ln_post, @proj   (:376-399)
  xlv2     ``ResamplerXLV2(dim 1024, depth 4, dim_head 64, heads 16, 64 queries, 4096 -> 768 + 1280)`` on a
           [2, 256, 4096] batch                                (src/models_ipa/resampler.py:228-284)
  gen      ``ContinuousLVLM.generate`` semantics at hidden 4096 / 32 heads / inter 11008 / vocab 32066 (2 layers):
           real input resampler -> splice -> real LlamaForCausalLM forward + real logits processor in the restated
           HF greedy loop -> 64 hidden rows -> real output resampler -> ``img_gen_feat`` [1, 256, 4096]
                                                               (src/models_clm/models.py:98-221)

For every piece the oracle restatement (``oracle/seedstory_oracle.py``) is asserted against the reference (fp32:
<= 2e-6; bf16: within 1.5x of the reference's own bf16-vs-fp32 distance, see ``pin``) — this is the pin at REAL
dimensions.  Inputs and weights regenerate from seeds, so the fixture
(``tests/golden/frontend_full.safetensors``, < 3 MB) holds every k-th output row plus whole-tensor norms.
"""
import json
import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402
import seedstory_oracle as O  # noqa: E402
import synth  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")

E = 4096
RES_IN = dict(grid=8, embed=E, heads=32, n_kv=256, batch=2, seed=21, row_stride=8)
RES_OUT = dict(grid=16, embed=E, heads=32, n_kv=64, batch=1, seed=22, row_stride=16)
VIT = dict(width=1664, layers=1, heads=16, mlp_width=8192, mlp_ratio=4.9231, patch=14, out_dim=E, n_queries=256,
           image=448, seed=31, row_stride=16)
XLV2 = dict(dim=1024, depth=4, dim_head=64, heads=16, num_queries=64, embedding_dim=E, output1_dim=768,
            output2_dim=1280, ff_mult=4)
XLV2_RUN = dict(seed=41, batch=2, tokens=256, row_stride=4)
LLAMA = dict(hidden=E, n_heads=32, n_layers=2, inter=11008, vocab=32066)
IMG_IDS = list(range(32000, 32066))            # <img>, 64 x <img_000xx>, </img> (the tokenizer's 66 added ids)
GEN = dict(seed=11, n_text=12, max_new=90, hidden_stride=4, feat_stride=16)


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def check(name, mine, ref, tol):
    r = rel(mine, ref)
    print("  %-40s rel=%.3e" % (name, r), flush=True)
    assert r <= tol, (name, r)
    return r


_F32 = {}      # name -> the reference's fp32 output (the "truth" the bf16 gap is measured against)
GAPS = {}      # name -> [reference bf16 vs reference fp32, oracle bf16 vs reference bf16]


def pin(name, dtype, mine, ref):
    """fp32: restatement == reference within 2e-6.  bf16: a whole module in bf16 is a chaotic function of its rounding
    points (with random weights the attention output is a noise-like average, so a one-ulp score flip moves it by
    ~1e-2), therefore the restatement must sit within 1.5x of the reference's OWN bf16-vs-fp32 distance."""
    if dtype == torch.float32:
        _F32[name] = ref.float().clone()
        return check(name + " f32", mine, ref, 2e-6)
    gap = rel(ref, _F32[name])
    r = check(name + " bf16 (reference bf16 vs fp32: %.3e)" % gap, mine, ref, 1.5 * gap + 1e-3)
    GAPS[name] = [gap, r]
    return r


def store(out, tag, ref, stride):
    """Rows ``[::stride]`` of the flattened [rows, C] reference output (dtype preserved: bf16 rows stay bf16) and
    whole-tensor statistics."""
    flat = ref.reshape(-1, ref.shape[-1])
    out[tag + ".rows"] = flat[::stride].clone()
    out[tag + ".norm"] = ref.float().norm().reshape(1)
    out[tag + ".absmean"] = ref.float().abs().mean().reshape(1)


DTYPES = ((torch.float32, "f32"), (torch.bfloat16, "bf16"))


def golden_resamplers(qwen_mod, out):
    for name, c in (("res_in", RES_IN), ("res_out", RES_OUT)):
        for dtype, dtag in DTYPES:
            wd = synth.resampler_weights(c["seed"], "", c["grid"], c["embed"], dtype=dtype)
            m = qwen_mod.Resampler(grid_size=c["grid"], embed_dim=c["embed"], num_heads=c["heads"],
                                   kv_dim=c["embed"]).eval()
            missing, unexpected = m.load_state_dict(wd, strict=False)
            assert not missing and not unexpected, (missing, unexpected)
            m = m.to(dtype)
            x = synth.normal_like(c["seed"] + 100, (c["batch"], c["n_kv"], c["embed"]), 1.0, dtype=dtype)
            with torch.no_grad():
                ref = m(x)
            mine = O.resampler_forward(wd, "", x, c["heads"])
            pin(name, dtype, mine, ref)
            store(out, "%s_%s" % (name, dtag), ref, c["row_stride"])


def golden_vit_ends(qwen_mod, out):
    c = VIT
    for dtype, dtag in DTYPES:
        wd = synth.vit_weights(c["seed"], c["width"], c["layers"], c["heads"], c["mlp_width"], c["patch"],
                               c["out_dim"], c["n_queries"], dtype=dtype)
        m = qwen_mod.VisionTransformerWithAttnPool(image_size=c["image"], patch_size=c["patch"], width=c["width"],
                                                   layers=c["layers"], heads=c["heads"], mlp_ratio=c["mlp_ratio"],
                                                   n_queries=c["n_queries"], output_dim=c["out_dim"]).eval()
        assert m.transformer.resblocks[0].mlp.c_fc.weight.shape[0] == c["mlp_width"]
        missing, unexpected = m.load_state_dict(wd, strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        m = m.to(dtype)
        x = synth.normal_like(c["seed"] + 100, (1, 3, c["image"], c["image"]), 1.0, dtype=dtype)
        with torch.no_grad():
            ref = m(x)
        mine = O.vit_forward(wd, x, width=c["width"], layers=c["layers"], heads=c["heads"], patch=c["patch"],
                             out_dim=c["out_dim"], n_queries=c["n_queries"])
        pin("vit ends", dtype, mine, ref)
        store(out, "vit_" + dtag, ref, c["row_stride"])


def golden_xlv2(ipa_mod, out):
    c, r = XLV2, XLV2_RUN
    for dtype, dtag in DTYPES:
        wd = synth.resampler_xlv2_weights(r["seed"], dtype=dtype, **c)
        m = ipa_mod.ResamplerXLV2(**c).eval()
        missing, unexpected = m.load_state_dict(wd, strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        m = m.to(dtype)
        x = synth.normal_like(r["seed"] + 100, (r["batch"], r["tokens"], c["embedding_dim"]), 1.0, dtype=dtype)
        with torch.no_grad():
            ctx, pooled = m(x)
        mc, mp = O.resampler_xlv2_forward(wd, x, depth=c["depth"], heads=c["heads"], dim_head=c["dim_head"])
        pin("xlv2 ctx", dtype, mc, ctx)
        pin("xlv2 pooled", dtype, mp, pooled)
        store(out, "xlv2_ctx_" + dtag, ctx, r["row_stride"])
        store(out, "xlv2_pooled_" + dtag, pooled, 1)


class _FakeTok:
    def encode(self, s, add_special_tokens=False):
        return list(IMG_IDS)


def gen_weights(dtype):
    d = LLAMA
    wd = synth.llama_weights(GEN["seed"], d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], dtype=dtype)
    wd.update(synth.resampler_weights(RES_IN["seed"], "input_resampler.", RES_IN["grid"], E, dtype=dtype))
    wd.update(synth.resampler_weights(RES_OUT["seed"], "output_resampler.", RES_OUT["grid"], E, dtype=dtype))
    return wd


def gen_inputs(dtype):
    """Prompt ``<s> 12 text ids <img> 64 placeholders </img>``, one 256 x 4096 input image feature, forced schedule."""
    boi, eoi = IMG_IDS[0], IMG_IDS[-1]
    n_text = GEN["n_text"]
    prompt = [1] + synth.randint(50, (n_text,), 3, 32000).tolist() + [boi] + IMG_IDS[1:65] + [eoi]
    input_ids = torch.tensor([prompt])
    ids_cmp_mask = torch.zeros_like(input_ids, dtype=torch.bool)
    ids_cmp_mask[0, n_text + 2:n_text + 2 + 64] = True
    image_embeds = synth.normal_like(51, (1, 256, E), 1.0, dtype=dtype)
    # fp32: 6 teacher-forced caption tokens then <img>, free-running afterwards.  bf16: the whole tail is forced too
    # (placeholders, </img>, EOS) so that a one-ulp argmax flip cannot change the sequence the features are compared on
    forced = synth.randint(52, (6,), 3, 32000).tolist() + [boi]
    if dtype != torch.float32:
        forced = forced + IMG_IDS[1:] + [2]
    return input_ids, ids_cmp_mask, image_embeds, forced


def golden_generate(llama_mod, gen_mod, qwen_mod, out):
    from transformers import LlamaConfig
    d = LLAMA
    for dtype, dtag in DTYPES:
        cfg = LlamaConfig(hidden_size=d["hidden"], intermediate_size=d["inter"], num_hidden_layers=d["n_layers"],
                          num_attention_heads=d["n_heads"], vocab_size=d["vocab"], max_position_embeddings=4096,
                          rms_norm_eps=1e-5)
        wd = gen_weights(dtype)
        m = llama_mod.LlamaForCausalLM(cfg).eval()
        m.load_state_dict({k: v for k, v in wd.items() if "resampler" not in k}, strict=False)
        m = m.to(dtype)
        m.use_kv_cache_head = False
        rin = qwen_mod.Resampler(grid_size=RES_IN["grid"], embed_dim=E, num_heads=32, kv_dim=E).eval()
        rin.load_state_dict({k[len("input_resampler."):]: v for k, v in wd.items() if k.startswith("input_resampler.")})
        rout = qwen_mod.Resampler(grid_size=RES_OUT["grid"], embed_dim=E, num_heads=32, kv_dim=E).eval()
        rout.load_state_dict({k[len("output_resampler."):]: v for k, v in wd.items() if k.startswith("output_resampler.")})
        rin, rout = rin.to(dtype), rout.to(dtype)
        proc = gen_mod.AutoImageTokenGenerationProcessor(tokenizer=_FakeTok(), num_img_gen_tokens=64)
        input_ids, ids_cmp_mask, image_embeds, forced = gen_inputs(dtype)
        embeds_cmp_mask = torch.tensor([True])
        eoi = IMG_IDS[-1]
        prompt = input_ids[0].tolist()
        with torch.no_grad():
            emb = m.get_input_embeddings()(input_ids)
            emb[ids_cmp_mask] = rin(image_embeds)[embeds_cmp_mask].view(-1, E)           # models.py:133-135
            S = input_ids.shape[1]
            r = m(inputs_embeds=emb, position_ids=torch.arange(S).unsqueeze(0), use_cache=True,
                  output_hidden_states=True, return_dict=True)
            seq = list(prompt)
            gen, hid = [], []
            logits = r.logits[:, -1]
            kv = r.past_key_values
            while True:
                sc = proc(torch.tensor([seq]), logits.clone())
                tok = int(sc.argmax(-1))
                if len(gen) < len(forced):
                    tok = forced[len(gen)]
                gen.append(tok)
                seq.append(tok)
                if tok == 2 or len(gen) >= GEN["max_new"]:
                    break
                r = m(input_ids=torch.tensor([[tok]]), position_ids=torch.tensor([[len(seq) - 1]]),
                      past_key_values=kv, use_cache=True, output_hidden_states=True, return_dict=True)
                kv = r.past_key_values
                hid.append(r.hidden_states[-1][0, -1])
                logits = r.logits[:, -1]
            hidden = torch.stack(hid)
            e = max(i for i, t in enumerate(gen) if t == eoi)
            feat = rout(hidden[e - 64:e].unsqueeze(0))                                    # models.py:197,205
        dims = O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
        mine = O.lvlm_generate(wd, dims, input_ids, image_embeds, embeds_cmp_mask, ids_cmp_mask, IMG_IDS,
                               max_new_tokens=GEN["max_new"], forced=forced, n_heads_resampler=32)
        assert mine["generate_ids"] == gen, (mine["generate_ids"], gen)
        pin("gen feed rows", dtype, mine["hidden"][e - 64:e], hidden[e - 64:e])
        pin("gen img_gen_feat", dtype, mine["img_gen_feat"], feat)
        print("  generate %s: %d tokens, </img> at %d" % (dtag, len(gen), e), flush=True)
        tag = "gen_" + dtag
        out[tag + ".generate_ids"] = torch.tensor(gen)
        store(out, tag + ".feed", hidden[e - 64:e], GEN["hidden_stride"])               # the 64 regressor inputs
        store(out, tag + ".img_gen_feat", feat, GEN["feat_stride"])
        del m, rin, rout, wd


def main():
    torch.set_num_threads(8)
    llama_mod, qwen_mod, gen_mod, ipa_mod = ref_shims.import_reference()
    os.makedirs(GOLD, exist_ok=True)
    out = {}
    print("resamplers, 4096 / 32 heads"); golden_resamplers(qwen_mod, out)
    print("ViT-G ends, real size"); golden_vit_ends(qwen_mod, out)
    print("ResamplerXLV2, real config"); golden_xlv2(ipa_mod, out)
    print("generate, hidden 4096"); golden_generate(llama_mod, gen_mod, qwen_mod, out)
    out = {k: v.contiguous() for k, v in out.items()}
    path = os.path.join(GOLD, "frontend_full.safetensors")
    save_file(out, path)
    meta = dict(RES_IN=RES_IN, RES_OUT=RES_OUT, VIT=VIT, XLV2=XLV2, XLV2_RUN=XLV2_RUN, LLAMA=LLAMA,
                IMG_IDS=[IMG_IDS[0], IMG_IDS[-1]], GEN=GEN, BF16_GAPS=GAPS,
                source="reference modules under /root/reference run on CPU via oracle/ref_shims.py",
                torch=torch.__version__)
    with open(os.path.join(GOLD, "frontend_full.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote %d tensors, %.1f KiB" % (len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
