"""TEST INFRASTRUCTURE ONLY — pins the oracle against the REAL reference and writes goldens.

Run in the build container (needs ``/root/reference``):

    python oracle/make_golden.py

For every piece of the hot path whose reference module imports here this
script (1) builds the reference module, loads the synthetic weights of ``oracle/synth.py``
into it, runs it on CPU, (2) runs the restatement in ``oracle/seedstory_oracle.py`` on the
same weights/inputs, (3) asserts they agree (bit-exact where the op order is identical,
else <= 2e-6 relative in fp32), and (4) stores the *reference's* outputs under
``tests/golden/*.safetensors``.  ``tests/test_oracle_golden.py`` re-checks the oracle
against these files on any box (the reference tree does not travel to the GPU box).
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

# Tiny-but-structurally-faithful configs (head_dim 128 like LLaMA-7B; ViT head_dim 104 like ViT-G).
LLAMA = dict(hidden=256, n_heads=2, n_layers=2, inter=512, vocab=320)
IMG_IDS = list(range(320 - 66, 320))  # <img>, 64 x <img_000xx>, </img>
RES_IN = dict(grid=4, embed=256, heads=2)     # 16 queries over 64 kv tokens
RES_OUT = dict(grid=8, embed=256, heads=2)    # 64 queries over 16 kv tokens
VIT = dict(width=208, layers=2, heads=2, mlp_width=512, patch=14, out_dim=256, n_queries=16, image=56)
XLV2 = dict(dim=128, depth=2, dim_head=32, heads=4, num_queries=8, embedding_dim=256, output1_dim=48,
            output2_dim=80, ff_mult=4)


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def check(name, mine, ref, tol):
    r = rel(mine, ref)
    exact = bool(torch.equal(mine, ref))
    print("  %-34s rel=%.3e exact=%s" % (name, r, exact))
    assert r <= tol, (name, r)
    return r


def golden_llama(llama_mod, dtype, tag, out):
    from transformers import LlamaConfig
    d = LLAMA
    cfg = LlamaConfig(hidden_size=d["hidden"], intermediate_size=d["inter"], num_hidden_layers=d["n_layers"],
                      num_attention_heads=d["n_heads"], vocab_size=d["vocab"], max_position_embeddings=4096,
                      rms_norm_eps=1e-5)
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], dtype=dtype)
    m = llama_mod.LlamaForCausalLM(cfg).eval()
    missing, unexpected = m.load_state_dict(wd, strict=False)
    assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    m = m.to(dtype)
    m.use_kv_cache_head = False
    dims = O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    ids = synth.randint(5, (1, 37), 3, 250)
    emb = wd["model.embed_tokens.weight"][ids]
    tol = 2e-6 if dtype == torch.float32 else 1e-2  # bf16: one-ulp flips from fp32 matmul blocking
    with torch.no_grad():
        # prefill
        r1 = m(inputs_embeds=emb, position_ids=torch.arange(37).unsqueeze(0), use_cache=True,
               output_hidden_states=True, return_dict=True)
        o1 = O.llama_forward(wd, dims, emb, torch.arange(37).unsqueeze(0), None)
        check(tag + " prefill logits", o1[0], r1.logits, tol)
        check(tag + " prefill hidden", o1[1], r1.hidden_states[-1], tol)
        check(tag + " prefill k0", o1[2][0][0], r1.past_key_values[0][0], tol)
        # continuation of 9 tokens against the cached prefix (bottom-right causal)
        ids2 = synth.randint(6, (1, 9), 3, 250)
        emb2 = wd["model.embed_tokens.weight"][ids2]
        pos2 = torch.arange(37, 46).unsqueeze(0)
        r2 = m(inputs_embeds=emb2, position_ids=pos2, past_key_values=r1.past_key_values, use_cache=True,
               output_hidden_states=True, return_dict=True)
        o2 = O.llama_forward(wd, dims, emb2, pos2, o1[2])
        check(tag + " continuation logits", o2[0], r2.logits, tol)
        # single-token decode
        ids3 = synth.randint(7, (1, 1), 3, 250)
        emb3 = wd["model.embed_tokens.weight"][ids3]
        pos3 = torch.tensor([[46]])
        r3 = m(inputs_embeds=emb3, position_ids=pos3, past_key_values=r2.past_key_values, use_cache=True,
               output_hidden_states=True, return_dict=True)
        o3 = O.llama_forward(wd, dims, emb3, pos3, o2[2])
        check(tag + " decode logits", o3[0], r3.logits, tol)
        # building blocks straight from the reference classes
        layer0 = m.model.layers[0]
        x = synth.normal_like(99, (1, 5, d["hidden"]), 1.0, dtype=dtype)
        check(tag + " rmsnorm", O.rmsnorm(x, wd["model.layers.0.input_layernorm.weight"], 1e-5),
              layer0.input_layernorm(x), 0.0)
        q = synth.normal_like(98, (1, 2, 5, 128), 1.0, dtype=dtype)
        cos, sin = layer0.self_attn.rotary_emb(q, seq_len=64)
        pid = torch.tensor([[3, 9, 10, 40, 63]])
        rq, _ = llama_mod.apply_rotary_pos_emb(q, q, cos, sin, pid)
        c2, s2 = O.rope_tables(128, 4096, dtype)
        check(tag + " rope", O.apply_rope(q, c2, s2, pid), rq, 0.0)
    out.update({
        tag + ".ids": ids, tag + ".ids2": ids2, tag + ".ids3": ids3,
        tag + ".prefill_logits": r1.logits.float(), tag + ".prefill_hidden": r1.hidden_states[-1].float(),
        tag + ".prefill_k0": r1.past_key_values[0][0].float(), tag + ".prefill_v1": r1.past_key_values[1][1].float(),
        tag + ".cont_logits": r2.logits.float(), tag + ".cont_hidden": r2.hidden_states[-1].float(),
        tag + ".decode_logits": r3.logits.float(), tag + ".decode_hidden": r3.hidden_states[-1].float(),
        tag + ".rmsnorm_in": x.float(), tag + ".rmsnorm_out": layer0.input_layernorm(x).float(),
        tag + ".rope_in": q.float(), tag + ".rope_out": rq.float(),
    })


class _FakeTok:
    def encode(self, s, add_special_tokens=False):
        return list(IMG_IDS)


def golden_processor(gen_mod, out):
    proc = gen_mod.AutoImageTokenGenerationProcessor(tokenizer=_FakeTok(), num_img_gen_tokens=64)
    cases = [17, IMG_IDS[0], IMG_IDS[5], IMG_IDS[64], IMG_IDS[65], 2]
    res = []
    for i, last in enumerate(cases):
        for dt in (torch.float32, torch.bfloat16):
            sc = synth.normal_like(300 + i, (1, LLAMA["vocab"]), 2.0, dtype=dt)
            ref = proc(torch.tensor([[5, last]]), sc.clone())
            mine = O.image_token_logits_processor(last, sc[0].clone(), IMG_IDS)
            assert torch.equal(mine, ref[0]), (last, dt)
            if dt == torch.float32:
                res.append(ref[0])
    print("  logits processor: %d cases bit-exact (fp32 + bf16)" % (2 * len(cases)))
    out["proc.last_ids"] = torch.tensor(cases)
    out["proc.out"] = torch.stack(res)


def golden_resampler(qwen_mod, out):
    for tag, cfg, n_kv, seed in (("res_in", RES_IN, 64, 21), ("res_out", RES_OUT, 16, 22)):
        wd = synth.resampler_weights(seed, "", cfg["grid"], cfg["embed"])
        m = qwen_mod.Resampler(grid_size=cfg["grid"], embed_dim=cfg["embed"], num_heads=cfg["heads"],
                               kv_dim=cfg["embed"]).eval()
        missing, unexpected = m.load_state_dict(wd, strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        x = synth.normal_like(seed + 100, (3, n_kv, cfg["embed"]), 1.0)
        with torch.no_grad():
            ref = m(x)
        mine = O.resampler_forward(wd, "", x, cfg["heads"])
        check(tag, mine, ref, 2e-6)
        assert torch.equal(m.pos_embed, wd["pos_embed"])  # sincos table restatement is bit-exact
        out[tag + ".x"] = x
        out[tag + ".y"] = ref


def golden_vit(qwen_mod, out):
    c = VIT
    wd = synth.vit_weights(31, c["width"], c["layers"], c["heads"], c["mlp_width"], c["patch"], c["out_dim"],
                           c["n_queries"])
    m = qwen_mod.VisionTransformerWithAttnPool(image_size=c["image"], patch_size=c["patch"], width=c["width"],
                                               layers=c["layers"], heads=c["heads"],
                                               mlp_ratio=c["mlp_width"] / c["width"], n_queries=c["n_queries"],
                                               output_dim=c["out_dim"]).eval()
    assert m.transformer.resblocks[0].mlp.c_fc.weight.shape[0] == c["mlp_width"]
    missing, unexpected = m.load_state_dict(wd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    x = synth.normal_like(131, (2, 3, c["image"], c["image"]), 1.0)
    with torch.no_grad():
        ref = m(x)
    mine = O.vit_forward(wd, x, width=c["width"], layers=c["layers"], heads=c["heads"], patch=c["patch"],
                         out_dim=c["out_dim"], n_queries=c["n_queries"])
    check("vit", mine, ref, 2e-6)
    out["vit.x"] = x
    out["vit.y"] = ref


def golden_xlv2(ipa_mod, out):
    c = XLV2
    wd = synth.resampler_xlv2_weights(41, **c)
    m = ipa_mod.ResamplerXLV2(**c).eval()
    missing, unexpected = m.load_state_dict(wd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    x = synth.normal_like(141, (2, 16, c["embedding_dim"]), 1.0)
    with torch.no_grad():
        ctx, pooled = m(x)
    mc, mp = O.resampler_xlv2_forward(wd, x, depth=c["depth"], heads=c["heads"], dim_head=c["dim_head"])
    check("xlv2 ctx", mc, ctx, 2e-6)
    check("xlv2 pooled", mp, pooled, 2e-6)
    out["xlv2.x"] = x
    out["xlv2.ctx"] = ctx
    out["xlv2.pooled"] = pooled


def golden_generate(llama_mod, gen_mod, qwen_mod, out, dtype=torch.float32, tag="gen"):
    """ContinuousLVLM.generate semantics: the reference *model forward*, *logits processor* and
    *Resampler* are the real classes; the HF-4.34 greedy loop around them is restated here
    (transformers 5.x cannot drive this model) — 'parity unpinned' at that boundary."""
    from transformers import LlamaConfig
    d = LLAMA
    cfg = LlamaConfig(hidden_size=d["hidden"], intermediate_size=d["inter"], num_hidden_layers=d["n_layers"],
                      num_attention_heads=d["n_heads"], vocab_size=d["vocab"], max_position_embeddings=4096,
                      rms_norm_eps=1e-5)
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], dtype=dtype)
    wd.update(synth.resampler_weights(21, "input_resampler.", RES_IN["grid"], RES_IN["embed"], dtype=dtype))
    wd.update(synth.resampler_weights(22, "output_resampler.", RES_OUT["grid"], RES_OUT["embed"], dtype=dtype))
    m = llama_mod.LlamaForCausalLM(cfg).eval()
    m.load_state_dict({k: v for k, v in wd.items() if "resampler" not in k}, strict=False)
    m = m.to(dtype)
    m.use_kv_cache_head = False
    rin = qwen_mod.Resampler(grid_size=RES_IN["grid"], embed_dim=256, num_heads=2, kv_dim=256).eval()
    rin.load_state_dict({k[len("input_resampler."):]: v for k, v in wd.items() if k.startswith("input_resampler.")})
    rout = qwen_mod.Resampler(grid_size=RES_OUT["grid"], embed_dim=256, num_heads=2, kv_dim=256).eval()
    rout.load_state_dict({k[len("output_resampler."):]: v for k, v in wd.items() if k.startswith("output_resampler.")})
    rin, rout = rin.to(dtype), rout.to(dtype)
    proc = gen_mod.AutoImageTokenGenerationProcessor(tokenizer=_FakeTok(), num_img_gen_tokens=64)
    # NOTE tiny config: 16 image-input tokens per image (grid 4), 64 output tokens (as the 7B model).
    n_in = RES_IN["grid"] ** 2
    boi, eoi = IMG_IDS[0], IMG_IDS[-1]
    prompt = [1] + synth.randint(50, (12,), 3, 250).tolist() + [boi] + IMG_IDS[1:1 + n_in] + [eoi]
    input_ids = torch.tensor([prompt])
    ids_cmp_mask = torch.zeros_like(input_ids, dtype=torch.bool)
    ids_cmp_mask[0, 14:14 + n_in] = True
    embeds_cmp_mask = torch.tensor([True])
    image_embeds = synth.normal_like(51, (1, 64, 256), 1.0, dtype=dtype)
    # fp32: 6 teacher-forced caption tokens then <img>, free-running afterwards.  bf16: the whole tail after </img>
    # is forced too (EOS), so a one-ulp argmax flip cannot change the sequence the features are compared on
    forced = synth.randint(52, (6,), 3, 250).tolist() + [boi]
    if dtype != torch.float32:
        forced = forced + IMG_IDS[1:] + [2]
    with torch.no_grad():
        emb = m.get_input_embeddings()(input_ids)
        emb[ids_cmp_mask] = rin(image_embeds)[embeds_cmp_mask].view(-1, 256)
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
            if tok == 2 or len(gen) >= 90:
                break
            r = m(input_ids=torch.tensor([[tok]]), position_ids=torch.tensor([[len(seq) - 1]]),
                  past_key_values=kv, use_cache=True, output_hidden_states=True, return_dict=True)
            kv = r.past_key_values
            hid.append(r.hidden_states[-1][0, -1])
            logits = r.logits[:, -1]
        hidden = torch.stack(hid)
        e = max(i for i, t in enumerate(gen) if t == eoi)
        feat = rout(hidden[e - 64:e].unsqueeze(0))
    dims = O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    mine = O.lvlm_generate(wd, dims, input_ids, image_embeds, embeds_cmp_mask, ids_cmp_mask, IMG_IDS,
                           max_new_tokens=90, forced=forced, n_heads_resampler=2)
    assert mine["generate_ids"] == gen, (mine["generate_ids"], gen)
    tol = 2e-6 if dtype == torch.float32 else 1e-2
    check(tag + " hidden", mine["hidden"], hidden, tol)
    check(tag + " img_gen_feat", mine["img_gen_feat"], feat, tol)
    print("  generate: %d tokens, eoi at %d" % (len(gen), e))
    out[tag + ".input_ids"] = input_ids
    out[tag + ".image_embeds"] = image_embeds.float()
    out[tag + ".forced"] = torch.tensor(forced)
    out[tag + ".generate_ids"] = torch.tensor(gen)
    out[tag + ".hidden"] = hidden.float()
    out[tag + ".img_gen_feat"] = feat.float()


VITBLK = dict(width=1664, heads=16, mlp_width=8192, tokens=1024, row_stride=32)   # one ViT-G block, full width


def golden_vit_block_full(qwen_mod, out):
    """One VisualAttentionBlock at ViT-G width (1664, 16 heads x 104, MLP 8192) on 1024 tokens: the REAL reference
    class (src/models/qwen_visual.py:238-287) in fp32 and in bf16.  Weights and input regenerate from seeds
    (oracle/synth.py), so the fixture holds only every 32nd output row plus whole-tensor norms."""
    from functools import partial
    c = VITBLK
    for dtype, tag in ((torch.float32, "vitblk_f32"), (torch.bfloat16, "vitblk_bf16")):
        wd = synth.vit_block_weights(61, c["width"], c["mlp_width"], dtype=dtype)
        blk = qwen_mod.VisualAttentionBlock(c["width"], c["heads"], c["mlp_width"] / c["width"],
                                            norm_layer=partial(torch.nn.LayerNorm, eps=1e-6)).eval()
        assert blk.mlp.c_fc.weight.shape[0] == c["mlp_width"]
        missing, unexpected = blk.load_state_dict(wd, strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        blk = blk.to(dtype)
        x = synth.normal_like(161, (c["tokens"], 1, c["width"]), 1.0, dtype=dtype)     # [sq, b, h] (reference layout)
        with torch.no_grad():
            ref = blk(x)
        mine = O.vit_block_forward(wd, "", x.transpose(0, 1), c["heads"]).transpose(0, 1)
        check(tag, mine, ref, 2e-6 if dtype == torch.float32 else 1e-2)
        out[tag + ".y_rows"] = ref[::c["row_stride"], 0].float()
        out[tag + ".y_norm"] = ref.float().norm().reshape(1)
        out[tag + ".y_absmean"] = ref.float().abs().mean().reshape(1)


def main():
    torch.set_num_threads(8)
    llama_mod, qwen_mod, gen_mod, ipa_mod = ref_shims.import_reference()
    os.makedirs(GOLD, exist_ok=True)
    out = {}
    print("llama fp32"); golden_llama(llama_mod, torch.float32, "llama_f32", out)
    print("llama bf16"); golden_llama(llama_mod, torch.bfloat16, "llama_bf16", out)
    print("processor"); golden_processor(gen_mod, out)
    print("resamplers"); golden_resampler(qwen_mod, out)
    print("vit"); golden_vit(qwen_mod, out)
    print("xlv2"); golden_xlv2(ipa_mod, out)
    print("generate"); golden_generate(llama_mod, gen_mod, qwen_mod, out)
    print("generate bf16"); golden_generate(llama_mod, gen_mod, qwen_mod, out, torch.bfloat16, "gen_bf16")
    print("vit block, ViT-G width"); golden_vit_block_full(qwen_mod, out)
    out = {k: v.contiguous() for k, v in out.items()}
    save_file(out, os.path.join(GOLD, "hotpath_tiny.safetensors"))
    meta = dict(LLAMA=LLAMA, IMG_IDS=[IMG_IDS[0], IMG_IDS[-1]], RES_IN=RES_IN, RES_OUT=RES_OUT, VIT=VIT, XLV2=XLV2,
                VITBLK=VITBLK,
                source="reference modules under /root/reference run on CPU via oracle/ref_shims.py",
                torch=torch.__version__)
    with open(os.path.join(GOLD, "hotpath_tiny.json"), "w") as f:
        json.dump(meta, f, indent=1)
    sz = os.path.getsize(os.path.join(GOLD, "hotpath_tiny.safetensors"))
    print("wrote %d tensors, %.1f KiB" % (len(out), sz / 1024))


if __name__ == "__main__":
    main()
