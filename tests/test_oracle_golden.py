"""CPU: the oracle restatement (oracle/seedstory_oracle.py) against the golden vectors that
oracle/make_golden.py produced by running the REAL reference modules."""
import os

import torch

import seedstory_oracle as O
import synth


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def _llama(meta, dtype):
    d = meta["LLAMA"]
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], dtype=dtype)
    dims = O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    return wd, dims


def _img_ids(meta):
    lo, hi = meta["IMG_IDS"]
    return list(range(lo, hi + 1))


def test_llama_prefill_continuation_decode_fp32(golden):
    g, meta = golden
    wd, dims = _llama(meta, torch.float32)
    emb = wd["model.embed_tokens.weight"]
    lg, hid, kv = O.llama_forward(wd, dims, emb[g["llama_f32.ids"]], torch.arange(37).unsqueeze(0))
    assert rel(lg, g["llama_f32.prefill_logits"]) < 2e-6
    assert rel(hid, g["llama_f32.prefill_hidden"]) < 2e-6
    assert rel(kv[0][0], g["llama_f32.prefill_k0"]) < 2e-6
    assert rel(kv[1][1], g["llama_f32.prefill_v1"]) < 2e-6
    lg2, hid2, kv2 = O.llama_forward(wd, dims, emb[g["llama_f32.ids2"]], torch.arange(37, 46).unsqueeze(0), kv)
    assert rel(lg2, g["llama_f32.cont_logits"]) < 2e-6
    lg3, hid3, _ = O.llama_forward(wd, dims, emb[g["llama_f32.ids3"]], torch.tensor([[46]]), kv2)
    assert rel(lg3, g["llama_f32.decode_logits"]) < 2e-6
    assert rel(hid3, g["llama_f32.decode_hidden"]) < 2e-6


def test_llama_bf16_rounding_points(golden):
    g, meta = golden
    wd, dims = _llama(meta, torch.bfloat16)
    emb = wd["model.embed_tokens.weight"]
    lg, hid, kv = O.llama_forward(wd, dims, emb[g["llama_bf16.ids"]], torch.arange(37).unsqueeze(0))
    # same ops in the same order as the reference's bf16 CPU path
    assert rel(lg, g["llama_bf16.prefill_logits"]) < 1e-2
    assert rel(kv[0][0], g["llama_bf16.prefill_k0"]) < 1e-2
    x = g["llama_bf16.rmsnorm_in"].bfloat16()
    y = O.rmsnorm(x, wd["model.layers.0.input_layernorm.weight"], 1e-5)
    assert torch.equal(y.float(), g["llama_bf16.rmsnorm_out"])
    q = g["llama_bf16.rope_in"].bfloat16()
    c, s = O.rope_tables(128, 4096, torch.bfloat16)
    assert torch.equal(O.apply_rope(q, c, s, torch.tensor([[3, 9, 10, 40, 63]])).float(), g["llama_bf16.rope_out"])


def test_rmsnorm_rope_fp32_exact(golden):
    g, meta = golden
    wd, dims = _llama(meta, torch.float32)
    y = O.rmsnorm(g["llama_f32.rmsnorm_in"], wd["model.layers.0.input_layernorm.weight"], 1e-5)
    assert rel(y, g["llama_f32.rmsnorm_out"]) < 1e-6
    c, s = O.rope_tables(128, 4096, torch.float32)
    r = O.apply_rope(g["llama_f32.rope_in"], c, s, torch.tensor([[3, 9, 10, 40, 63]]))
    assert rel(r, g["llama_f32.rope_out"]) < 1e-6


def test_logits_processor(golden):
    g, meta = golden
    ids = _img_ids(meta)
    for i, last in enumerate(g["proc.last_ids"].tolist()):
        sc = synth.normal_like(300 + i, (1, meta["LLAMA"]["vocab"]), 2.0)[0]
        out = O.image_token_logits_processor(last, sc.clone(), ids)
        assert torch.equal(out, g["proc.out"][i])


def test_resamplers(golden):
    g, meta = golden
    for tag, key, seed in (("res_in", "RES_IN", 21), ("res_out", "RES_OUT", 22)):
        c = meta[key]
        wd = synth.resampler_weights(seed, "", c["grid"], c["embed"])
        y = O.resampler_forward(wd, "", g[tag + ".x"], c["heads"])
        assert rel(y, g[tag + ".y"]) < 2e-6


def test_vit(golden):
    g, meta = golden
    c = meta["VIT"]
    wd = synth.vit_weights(31, c["width"], c["layers"], c["heads"], c["mlp_width"], c["patch"], c["out_dim"],
                           c["n_queries"])
    y = O.vit_forward(wd, g["vit.x"], width=c["width"], layers=c["layers"], heads=c["heads"], patch=c["patch"],
                      out_dim=c["out_dim"], n_queries=c["n_queries"])
    assert rel(y, g["vit.y"]) < 2e-6


def test_resampler_xlv2(golden):
    g, meta = golden
    c = meta["XLV2"]
    wd = synth.resampler_xlv2_weights(41, **c)
    ctx, pooled = O.resampler_xlv2_forward(wd, g["xlv2.x"], depth=c["depth"], heads=c["heads"],
                                           dim_head=c["dim_head"])
    assert rel(ctx, g["xlv2.ctx"]) < 2e-6
    assert rel(pooled, g["xlv2.pooled"]) < 2e-6


def test_lvlm_generate(golden):
    g, meta = golden
    wd, dims = _llama(meta, torch.float32)
    wd.update(synth.resampler_weights(21, "input_resampler.", meta["RES_IN"]["grid"], 256))
    wd.update(synth.resampler_weights(22, "output_resampler.", meta["RES_OUT"]["grid"], 256))
    input_ids = g["gen.input_ids"]
    n_in = meta["RES_IN"]["grid"] ** 2
    mask = torch.zeros_like(input_ids, dtype=torch.bool)
    mask[0, 14:14 + n_in] = True
    out = O.lvlm_generate(wd, dims, input_ids, g["gen.image_embeds"], torch.tensor([True]), mask, _img_ids(meta),
                          max_new_tokens=90, forced=g["gen.forced"].tolist(), n_heads_resampler=2)
    assert out["generate_ids"] == g["gen.generate_ids"].tolist()
    assert rel(out["hidden"], g["gen.hidden"]) < 2e-6
    assert rel(out["img_gen_feat"], g["gen.img_gen_feat"]) < 2e-6


def test_lora_linear_equals_merged():
    x = synth.normal_like(1, (3, 64), 1.0)
    w = synth.normal_like(2, (32, 64))
    a = synth.normal_like(3, (16, 64))
    b = synth.normal_like(4, (32, 16))
    y = O.lora_linear(x, w, a, b, 2.0)
    ym = torch.nn.functional.linear(x, O.lora_merge(w, a, b, 2.0))
    assert rel(ym, y) < 1e-6


def test_lvlm_generate_bf16(golden):
    """The bf16 reference run of ContinuousLVLM.generate (real forward / processor / Resampler classes in bf16 on
    CPU): the oracle rounds where the reference rounds."""
    g, meta = golden
    wd, dims = _llama(meta, torch.bfloat16)
    wd.update(synth.resampler_weights(21, "input_resampler.", meta["RES_IN"]["grid"], 256, dtype=torch.bfloat16))
    wd.update(synth.resampler_weights(22, "output_resampler.", meta["RES_OUT"]["grid"], 256, dtype=torch.bfloat16))
    input_ids = g["gen_bf16.input_ids"]
    n_in = meta["RES_IN"]["grid"] ** 2
    mask = torch.zeros_like(input_ids, dtype=torch.bool)
    mask[0, 14:14 + n_in] = True
    out = O.lvlm_generate(wd, dims, input_ids, g["gen_bf16.image_embeds"].bfloat16(), torch.tensor([True]), mask,
                          _img_ids(meta), max_new_tokens=90, forced=g["gen_bf16.forced"].tolist(), n_heads_resampler=2)
    assert out["generate_ids"] == g["gen_bf16.generate_ids"].tolist()
    assert rel(out["img_gen_feat"], g["gen_bf16.img_gen_feat"]) < 1e-2
    # the reference's own bf16-vs-fp32 distance on the regressed feature (what "bf16 parity" can mean at best)
    assert rel(g["gen_bf16.img_gen_feat"], g["gen.img_gen_feat"]) < 5e-2


def test_vit_block_full_width(golden):
    """One VisualAttentionBlock at ViT-G width (1664 / 16 heads of 104 / MLP 8192, 1024 tokens) against the rows
    the REAL reference class produced (oracle/make_golden.py::golden_vit_block_full)."""
    g, meta = golden
    c = meta["VITBLK"]
    wd = synth.vit_block_weights(61, c["width"], c["mlp_width"])
    x = synth.normal_like(161, (c["tokens"], 1, c["width"]), 1.0).transpose(0, 1)
    y = O.vit_block_forward(wd, "", x, c["heads"])[0]
    assert rel(y[::c["row_stride"]], g["vitblk_f32.y_rows"]) < 2e-6
    assert abs(float(y.norm()) / float(g["vitblk_f32.y_norm"]) - 1) < 1e-6


# ---- full-dimension fixtures (oracle/make_golden_full.py): the restatement against rows of the REAL reference -------------

def _full():
    import json
    import os
    from safetensors.torch import load_file
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(root, "frontend_full.json")) as f:
        meta = json.load(f)
    return load_file(os.path.join(root, "frontend_full.safetensors")), meta


def _rows(y, stride):
    return y.reshape(-1, y.shape[-1])[::stride]


def test_full_dimension_resamplers_vit_ends_xlv2_fp32():
    """Resampler 4096 / 32 heads (input and output = the regressor), ViT-G ends at 448^2 / 1664 / 4096, ResamplerXLV2 at
    the de-tokenizer's configuration: oracle == REAL reference rows, fp32, 2e-6."""
    g, meta = _full()
    for tag, key in (("res_in", "RES_IN"), ("res_out", "RES_OUT")):
        c = meta[key]
        wd = synth.resampler_weights(c["seed"], "", c["grid"], c["embed"])
        x = synth.normal_like(c["seed"] + 100, (c["batch"], c["n_kv"], c["embed"]), 1.0)
        y = O.resampler_forward(wd, "", x, c["heads"])
        assert rel(_rows(y, c["row_stride"]), g[tag + "_f32.rows"]) < 2e-6
        assert abs(float(y.norm()) - float(g[tag + "_f32.norm"])) < 1e-5 * float(g[tag + "_f32.norm"])
    c = meta["VIT"]
    wd = synth.vit_weights(c["seed"], c["width"], c["layers"], c["heads"], c["mlp_width"], c["patch"], c["out_dim"],
                           c["n_queries"])
    x = synth.normal_like(c["seed"] + 100, (1, 3, c["image"], c["image"]), 1.0)
    y = O.vit_forward(wd, x, width=c["width"], layers=c["layers"], heads=c["heads"], patch=c["patch"],
                      out_dim=c["out_dim"], n_queries=c["n_queries"])
    assert rel(_rows(y, c["row_stride"]), g["vit_f32.rows"]) < 2e-6
    c, r = meta["XLV2"], meta["XLV2_RUN"]
    wd = synth.resampler_xlv2_weights(r["seed"], **c)
    x = synth.normal_like(r["seed"] + 100, (r["batch"], r["tokens"], c["embedding_dim"]), 1.0)
    ctx, pooled = O.resampler_xlv2_forward(wd, x, depth=c["depth"], heads=c["heads"], dim_head=c["dim_head"])
    assert rel(_rows(ctx, r["row_stride"]), g["xlv2_ctx_f32.rows"]) < 2e-6
    assert rel(pooled, g["xlv2_pooled_f32.rows"]) < 2e-6


def test_greedy_loop_pinned_on_hf_generate():
    """The oracle's greedy loop == a REAL ``GenerationMixin.generate`` run (installed transformers 5.15; the reference pins
    4.34 — gap stated in oracle/make_golden_greedy.py) with the reference's real logits processor: ids incl. the 65
    processor-forced tokens and the free tail, EOS stop, last-layer hidden rows."""
    import json
    import os
    from safetensors.torch import load_file
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = load_file(os.path.join(root, "greedy_hf.safetensors"))
    with open(os.path.join(root, "greedy_hf.json")) as f:
        meta = json.load(f)
    d = meta["LLAMA"]
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    dims = O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    img_ids = list(range(meta["IMG_IDS"][0], meta["IMG_IDS"][1] + 1))
    for tag, eos in (("free", 2), ("eos", meta["eos_case_id"])):
        gen, hid, _, _ = O.greedy_generate(wd, dims, g["input_ids"], g["inputs_embeds"].clone(), img_ids, meta["MAX_NEW"],
                                           eos_id=eos)
        assert gen == g[tag + ".generate_ids"].tolist()
        assert rel(hid, g[tag + ".hidden"]) < 2e-6
    assert g["eos.generate_ids"].tolist()[-1] == meta["eos_case_id"] and len(g["eos.generate_ids"]) < meta["MAX_NEW"]


def test_full_dimension_generate_fp32():
    """``ContinuousLVLM.generate`` semantics at hidden 4096 / 32 heads / inter 11008 / vocab 32066 (2 layers) with the
    full-size resamplers: the oracle reproduces the ids, the 64 regressor input rows and ``img_gen_feat`` that the REAL
    reference modules produced (oracle/make_golden_full.py), fp32, 2e-6.  ~1.5 min and 6 GB on 8 cores."""
    g, meta = _full()
    d, gen = meta["LLAMA"], meta["GEN"]
    E = d["hidden"]
    lo, hi = meta["IMG_IDS"]
    img_ids = list(range(lo, hi + 1))
    wd = synth.llama_weights(gen["seed"], d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    wd.update(synth.resampler_weights(meta["RES_IN"]["seed"], "input_resampler.", meta["RES_IN"]["grid"], E))
    wd.update(synth.resampler_weights(meta["RES_OUT"]["seed"], "output_resampler.", meta["RES_OUT"]["grid"], E))
    n_text = gen["n_text"]
    prompt = [1] + synth.randint(50, (n_text,), 3, 32000).tolist() + [img_ids[0]] + img_ids[1:65] + [img_ids[-1]]
    input_ids = torch.tensor([prompt])
    mask = torch.zeros_like(input_ids, dtype=torch.bool)
    mask[0, n_text + 2:n_text + 2 + 64] = True
    image_embeds = synth.normal_like(51, (1, 256, E), 1.0)
    forced = synth.randint(52, (6,), 3, 32000).tolist() + [img_ids[0]]
    dims = O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    with torch.no_grad():
        out = O.lvlm_generate(wd, dims, input_ids, image_embeds, torch.tensor([True]), mask, img_ids, max_new_tokens=gen["max_new"],
                              forced=forced, n_heads_resampler=32)
    ids = out["generate_ids"]
    assert ids == g["gen_f32.generate_ids"].tolist()
    e = max(i for i, t in enumerate(ids) if t == img_ids[-1])
    assert rel(_rows(out["hidden"][e - 64:e], gen["hidden_stride"]), g["gen_f32.feed.rows"]) < 2e-6
    assert rel(_rows(out["img_gen_feat"], gen["feat_stride"]), g["gen_f32.img_gen_feat.rows"]) < 2e-6


def test_truth_cache_roundtrip_and_key_mismatch(tmp_path, monkeypatch):
    """tests/truth_cache.py (the cache of full-size CPU-oracle truths used by the GPU tests): a file written under
    SS_WRITE_GOLDEN_DIR loads back bit for bit when the key matches, and is ignored (recomputed) when any input tensor differs."""
    import truth_cache as TC
    g = torch.Generator().manual_seed(5)
    w = {"a": torch.randn(300, 70, generator=g).to(torch.bfloat16), "b": torch.randn(9, generator=g)}
    ids = [3, 1, 4, 1, 5]
    key = TC.tensors_key(w, ids, 8)
    assert key == TC.tensors_key(dict(reversed(list(w.items()))), ids, 8)        # dict order does not matter
    w2 = {k: v.clone() for k, v in w.items()}
    w2["a"][0, 0] += 1
    assert TC.tensors_key(w2, ids, 8) != key and TC.tensors_key(w, ids + [9], 8) != key and TC.tensors_key(w, ids, 4) != key
    calls = []

    def compute():
        calls.append(1)
        return {"f32.x": torch.arange(12.0).reshape(3, 4), "bf16.x": torch.arange(12.0).reshape(3, 4).to(torch.bfloat16)}
    monkeypatch.setattr(TC, "GOLDEN", str(tmp_path))
    monkeypatch.setenv("SS_WRITE_GOLDEN_DIR", str(tmp_path))
    monkeypatch.delenv("SS_IGNORE_TRUTH_CACHE", raising=False)
    R, how = TC.load_or_compute("unit_truth", key, compute)
    assert how == "computed" and len(calls) == 1
    R2, how2 = TC.load_or_compute("unit_truth", key, compute)
    assert how2 == "loaded" and len(calls) == 1
    assert all(torch.equal(R[k], R2[k]) and R[k].dtype == R2[k].dtype for k in R)
    R3, how3 = TC.load_or_compute("unit_truth", TC.tensors_key(w2, ids, 8), compute)
    assert how3 == "computed" and len(calls) == 2


def test_cached_full_size_truth_files_are_well_formed():
    """The committed truth files of the full-size MLLM-half tests carry a key and the tensors the tests read (shapes, dtypes, finite values).
    (Whether a box LOADS them is decided by the key over the weights it draws: oracle/synth.py is platform-independent by construction,
    the torch CPU generator of the 7B weights was checked to draw the same stream on the GPU box's host, tools/randn_fingerprint.py.)"""
    from safetensors import safe_open
    import truth_cache as TC
    want = {"vitg48_truth": {"f32": (32, 4096), "bf16": (32, 4096)},
            "llama7b_truth": {"f32.prefill_hidden": (64, 4096), "bf16.cont_hidden": (66, 4096), "f32.decode_hidden": (3, 4096),
                              "f32.prefill_logits": (32066,), "bf16.cont_logits": (32066,)},
            "story3_truth": {"f32.step0": (32, 4096), "bf16.step2": (32, 4096)}}
    for name, tensors in want.items():
        path = os.path.join(TC.GOLDEN, name + ".safetensors")
        assert os.path.exists(path), path
        with safe_open(path, "pt") as f:
            meta = f.metadata() or {}
            assert len(meta.get("key", "")) == 64 and "make_golden_mllm_full" in meta.get("generator", "")
            for k, shape in tensors.items():
                t = f.get_tensor(k)
                assert tuple(t.shape) == shape and bool(torch.isfinite(t.float()).all()), (name, k, t.shape)
                assert t.dtype == (torch.bfloat16 if k.startswith("bf16") else torch.float32), (name, k, t.dtype)
