"""GPU parity of the native engines (LLaMA prefill / continuation / hipGraph decode, Resampler,
ViT) and of the reference-API mirror (ContinuousLVLM.generate) against the committed golden
vectors, which were produced by the REAL reference modules (oracle/make_golden.py).

fp32 is the "matches the reference CPU path" gate: logits / hidden states / regressed image
features within 1e-4 relative (north-star tolerance: 1e-3 on img_gen_feat).  bf16 is checked
against the reference's own bf16 CPU run: within 2e-2 relative (one-ulp rounding flips
accumulate through the layers; LoRA-free weights here)."""
import pytest
import torch

import seedstory_oracle as O
import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _img_ids(meta):
    lo, hi = meta["IMG_IDS"]
    return list(range(lo, hi + 1))


def _engine(meta, dtype, **kw):
    from seedstory.llama import LlamaEngine
    d = meta["LLAMA"]
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], dtype=dtype)
    eng = LlamaEngine(wd, hidden=d["hidden"], n_heads=d["n_heads"], n_layers=d["n_layers"], inter=d["inter"],
                      vocab=d["vocab"], dtype=dtype, device=DEV, cache_cap=256, max_new=128, max_prefill_rows=64,
                      img_ids=_img_ids(meta), **kw)
    return eng, wd


@pytest.mark.parametrize("dtype,tag,tol", [(torch.float32, "llama_f32", 1e-4), (torch.bfloat16, "llama_bf16", 2e-2)])
def test_llama_prefill_continuation_decode(golden, dtype, tag, tol):
    g, meta = golden
    eng, wd = _engine(meta, dtype)
    emb = wd["model.embed_tokens.weight"]
    hid = eng.prefill(emb[g[tag + ".ids"][0]], want_hidden=True)
    assert eng.lengths() == (37, 37)
    assert rel(hid, g[tag + ".prefill_hidden"][0]) < tol
    assert rel(eng.logits, g[tag + ".prefill_logits"][0, -1]) < tol
    pkv = eng.past_key_values()
    assert pkv[0][0].shape == (1, 2, 37, 128)
    assert rel(pkv[0][0], g[tag + ".prefill_k0"]) < tol
    assert rel(pkv[1][1], g[tag + ".prefill_v1"]) < tol
    # continuation: 9 new rows against the cached prefix (bottom-right causal mask)
    hid2 = eng.prefill(emb[g[tag + ".ids2"][0]], want_hidden=True)
    assert eng.lengths() == (46, 46)
    assert rel(hid2, g[tag + ".cont_hidden"][0]) < tol
    assert rel(eng.logits, g[tag + ".cont_logits"][0, -1]) < tol
    # single-token decode through the captured graph: force the golden token id
    tok = int(g[tag + ".ids3"][0, 0])
    n = eng.generate(2, last_prompt_id=5, forced=[tok, 3])
    assert n == 2 and eng.gen_ids[:2].tolist() == [tok, 3]
    assert rel(eng.hidden_rows[0], g[tag + ".decode_hidden"][0, 0]) < tol
    assert eng.lengths() == (47, 47)


def test_llama_graph_equals_eager(golden):
    from seedstory import _lib
    g, meta = golden
    outs = []
    for use_graph in (1, 0):
        _lib.set_tuning("llama_graph", use_graph)
        try:
            eng, wd = _engine(meta, torch.bfloat16)
            eng.prefill(wd["model.embed_tokens.weight"][g["llama_bf16.ids"][0]])
            n = eng.generate(20, last_prompt_id=7)
            outs.append((n, eng.gen_ids[:n].tolist(), eng.hidden_rows[:n - 1].clone()))
        finally:
            _lib.set_tuning("llama_graph", 1)
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    assert torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("n_seq", [2, 3, 4, 6, 8])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_llama_slot_batched_decode_equals_single(golden, n_seq, dtype):
    """n_seq story slots decoded in lock-step (one sweep of the weights per token) must reproduce,
    slot by slot, the batch-1 engine: token ids, hidden-state rows and the KV cache.  Slots get
    different prompts, different forced prefixes, different stop points and one inactive slot."""
    g, meta = golden
    d = meta["LLAMA"]
    tag = "llama_f32" if dtype == torch.float32 else "llama_bf16"
    prompts = [synth.randint(40 + b, (19 + 5 * b,), 3, 250) for b in range(n_seq)]
    prompts[0] = g[tag + ".ids"][0]
    forced = [[7 + b, 9, 11 + b][: (b % 3) + 1] for b in range(n_seq)]
    forced[-1] = [5, 2, 8]  # EOS (2) forced at step 2: this slot stops early
    steps = 12
    single, wd = [], None
    for b in range(n_seq):
        eng, wd = _engine(meta, dtype)
        eng.prefill(wd["model.embed_tokens.weight"][prompts[b]])
        n = eng.generate(steps, last_prompt_id=int(prompts[b][-1]), forced=forced[b])
        kv = eng.lengths()[0]
        single.append((n, eng.gen_ids[:n].tolist(), eng.hidden_rows[:max(n - 1, 0)].clone(),
                       eng.k_cache[:, :, :kv].clone(), eng.v_cache[:, :, :kv].clone(), eng.lengths()))
        del eng
    eng, wd = _engine(meta, dtype, n_seq=n_seq)
    for b in range(n_seq):
        eng.select(b).prefill(wd["model.embed_tokens.weight"][prompts[b]])
    ns = eng.generate_batch(steps, [int(p[-1]) for p in prompts], forced=forced)
    assert ns[-1] == 2 and ns[0] == steps
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for b in range(n_seq):
        n, ids, hid, k, v, lens = single[b]
        eng.select(b)
        assert ns[b] == n and eng.lengths() == lens, b
        assert eng.gen_ids[:n].tolist() == ids, b
        kv = lens[0]
        assert rel(eng.hidden_rows[:max(n - 1, 0)], hid) < tol, b
        assert rel(eng.k_cache[:, :, :kv], k) < tol and rel(eng.v_cache[:, :, :kv], v) < tol, b
    # a second batch call with slot 0 inactive leaves slot 0 untouched and continues slot 1
    before = (eng.select(0).lengths(), eng.gen_ids[:4].clone())
    ns2 = eng.generate_batch(3, [1] * n_seq, forced=[[4, 4, 4]] * n_seq, active=[0] + [1] * (n_seq - 1))
    assert ns2[0] == 0 and ns2[1] == 3
    assert eng.select(0).lengths() == before[0] and torch.equal(eng.gen_ids[:4], before[1])
    # 3 steps = 2 forwarded tokens (the step that hits the limit is sampled but not forwarded)
    assert eng.select(1).lengths()[0] == single[1][5][0] + 2


def test_llama_lora_merge(golden):
    """LoRA factors present: engine (merged once, fp32) vs the oracle's unmerged peft formula."""
    g, meta = golden
    d = meta["LLAMA"]
    wd = synth.llama_weights(12, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], lora_r=16)
    from seedstory.llama import LlamaEngine
    eng = LlamaEngine(wd, hidden=d["hidden"], n_heads=d["n_heads"], n_layers=d["n_layers"], inter=d["inter"],
                      vocab=d["vocab"], dtype=torch.float32, device=DEV, cache_cap=128, max_new=16,
                      max_prefill_rows=64, img_ids=_img_ids(meta), lora_scaling=2.0)
    ids = synth.randint(5, (1, 21), 3, 250)
    emb = wd["model.embed_tokens.weight"][ids]
    dims = O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    lg, hid, _ = O.llama_forward(wd, dims, emb, torch.arange(21).unsqueeze(0))
    h = eng.prefill(emb[0], want_hidden=True)
    assert rel(h, hid[0]) < 1e-4
    assert rel(eng.logits, lg[0, -1]) < 1e-4


def test_kv_truncate_and_sink_gather(golden):
    """vis_george_sink.py:243 (truncate) and :266-295 (sink re-pack) on the slab."""
    g, meta = golden
    eng, wd = _engine(meta, torch.float32)
    emb = wd["model.embed_tokens.weight"]
    eng.prefill(emb[g["llama_f32.ids"][0]])
    k_before = eng.k_cache[:, :, :37].clone()
    keep, new_sink = O.sink_evict_indices(37, boi=10, eoi=27, sink_len=0, first=True)
    eng.kv_gather(keep)
    assert eng.lengths()[0] == len(keep)
    assert torch.equal(eng.k_cache[:, :, :len(keep)].cpu(), k_before[:, :, keep].cpu())
    eng.set_lengths(20, 20)
    assert eng.past_key_values()[0][0].shape[2] == 20


def test_resamplers(golden):
    from src.models.qwen_visual import Resampler
    g, meta = golden
    for tag, key, seed in (("res_in", "RES_IN", 21), ("res_out", "RES_OUT", 22)):
        c = meta[key]
        wd = synth.resampler_weights(seed, "", c["grid"], c["embed"])
        m = Resampler(grid_size=c["grid"], embed_dim=c["embed"], num_heads=c["heads"], kv_dim=c["embed"])
        missing, unexpected = m.load_state_dict(wd, strict=False)
        assert not missing and not unexpected
        m = m.to(DEV)
        y = m(g[tag + ".x"].to(DEV))
        assert rel(y, g[tag + ".y"]) < 1e-4
        yb = m.to(torch.bfloat16)(g[tag + ".x"].to(DEV, torch.bfloat16))
        assert rel(yb, g[tag + ".y"]) < 2e-2


def test_vit(golden):
    from src.models.qwen_visual import VisionTransformerWithAttnPool
    g, meta = golden
    c = meta["VIT"]
    wd = synth.vit_weights(31, c["width"], c["layers"], c["heads"], c["mlp_width"], c["patch"], c["out_dim"],
                           c["n_queries"])
    m = VisionTransformerWithAttnPool(image_size=c["image"], patch_size=c["patch"], width=c["width"],
                                      layers=c["layers"], heads=c["heads"], mlp_ratio=c["mlp_width"] / c["width"],
                                      n_queries=c["n_queries"], output_dim=c["out_dim"])
    missing, unexpected = m.load_state_dict(wd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    m = m.to(DEV)
    y = m(g["vit.x"].to(DEV))
    assert rel(y, g["vit.y"]) < 1e-4
    yb = m.to(torch.bfloat16)(g["vit.x"].to(DEV))
    assert rel(yb, g["vit.y"]) < 3e-2


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


def test_continuous_lvlm_generate_matches_reference(golden):
    """End to end through the reference API surface: ContinuousLVLM.generate -> img_gen_feat."""
    from src.models.qwen_visual import Resampler
    from src.models_clm.modeling_llama_xformer import LlamaConfig, LlamaForCausalLM
    from src.models_clm.models import ContinuousLVLM
    g, meta = golden
    d = meta["LLAMA"]
    cfg = LlamaConfig(hidden_size=d["hidden"], intermediate_size=d["inter"], num_hidden_layers=d["n_layers"],
                      num_attention_heads=d["n_heads"], vocab_size=d["vocab"])
    llm = LlamaForCausalLM(cfg)
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    missing, unexpected = llm.load_state_dict(wd, strict=False)
    assert not missing and not unexpected
    llm.cache_cap, llm.max_new, llm.max_prefill_rows = 256, 128, 64
    llm.use_kv_cache_head = False
    rin = Resampler(grid_size=meta["RES_IN"]["grid"], embed_dim=256, num_heads=2, kv_dim=256)
    rin.load_state_dict(synth.resampler_weights(21, "", meta["RES_IN"]["grid"], 256))
    rout = Resampler(grid_size=meta["RES_OUT"]["grid"], embed_dim=256, num_heads=2, kv_dim=256)
    rout.load_state_dict(synth.resampler_weights(22, "", meta["RES_OUT"]["grid"], 256))
    agent = ContinuousLVLM(llm, rin, rout).eval().to(DEV)
    input_ids = g["gen.input_ids"]
    n_in = meta["RES_IN"]["grid"] ** 2
    mask = torch.zeros_like(input_ids, dtype=torch.bool)
    mask[0, 14:14 + n_in] = True
    out = agent.generate(tokenizer=_Tok(_img_ids(meta)), input_ids=input_ids, image_embeds=g["gen.image_embeds"].to(DEV),
                         embeds_cmp_mask=torch.tensor([True]), ids_cmp_mask=mask, max_new_tokens=90,
                         num_img_gen_tokens=64, forced_tokens=g["gen.forced"].tolist())
    assert out["generate_ids"].tolist() == g["gen.generate_ids"].tolist()
    assert out["has_img_output"] and out["num_gen_imgs"] == 1
    r = rel(out["img_gen_feat"], g["gen.img_gen_feat"])
    assert r < 1e-4, r  # north-star gate is 1e-3 relative on the regressed image feature
    pkv = out["past_key_values"]
    assert len(pkv) == d["n_layers"] and pkv[0][0].shape[2] == input_ids.shape[1] + 90 - 1


def test_full_width_layer_properties():
    """BASELINE full sizes (hidden 4096 / inter 11008 / 32 heads), one layer: size-independent
    properties — (a) prefill-then-decode == prefill of the longer sequence (KV-cache consistency),
    (b) the decode graph is deterministic across replays."""
    from seedstory.llama import LlamaEngine
    torch.manual_seed(0)
    H, I, V = 4096, 11008, 32066
    wd = synth.llama_weights(3, H, 32, 1, I, 1024, dtype=torch.bfloat16)
    kw = dict(hidden=H, n_heads=32, n_layers=1, inter=I, vocab=1024, dtype=torch.bfloat16, device=DEV, cache_cap=512,
              max_new=32, max_prefill_rows=256, img_ids=list(range(900, 966)))
    emb = wd["model.embed_tokens.weight"]
    ids = synth.randint(9, (120,), 3, 800)
    a = LlamaEngine(wd, **kw)
    ha = a.prefill(emb[ids], want_hidden=True)
    b = LlamaEngine(wd, **kw)
    b.prefill(emb[ids[:100]])
    n = b.generate(21, last_prompt_id=int(ids[99]), forced=ids[100:].tolist() + [5])
    assert n == 21
    # rows 100..119 of the long prefill == decode rows 0..19 (same inputs, cached prefix)
    r = rel(b.hidden_rows[:20], ha[100:120])
    assert r < 2e-2, r
    assert rel(b.k_cache[0, :, :120], a.k_cache[0, :, :120]) < 1e-2
    c = LlamaEngine(wd, **kw)
    c.prefill(emb[ids[:100]])
    c.generate(21, last_prompt_id=int(ids[99]), forced=ids[100:].tolist() + [5])
    assert torch.equal(c.hidden_rows[:20], b.hidden_rows[:20])


def test_attention_sink_continuation_matches_oracle(golden):
    """Multimodal attention sink (clean spec of vis_george_sink.py:266-295): evict the oldest image on the KV
    slab, then continue with window-relative positions; hidden states must equal the oracle run on the gathered
    cache (keys keep their original rotary phase)."""
    from seedstory.llama import LlamaEngine
    from seedstory.story import StoryContext
    g, meta = golden
    d = meta["LLAMA"]
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    eng = LlamaEngine(wd, hidden=d["hidden"], n_heads=d["n_heads"], n_layers=d["n_layers"], inter=d["inter"],
                      vocab=d["vocab"], dtype=torch.float32, device=DEV, cache_cap=512, max_new=64, max_prefill_rows=256,
                      img_ids=_img_ids(meta))
    emb = wd["model.embed_tokens.weight"]
    dims = O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    ids_img = _img_ids(meta)
    ctx = StoryContext(bos_id=1, boi_id=ids_img[0], eoi_id=ids_img[-1], img_placeholder_ids=ids_img[1:-1], window=1)
    ctx.start(synth.randint(70, (6,), 3, 250).tolist(), torch.zeros(1, 4, 8))
    ctx.append_step(synth.randint(71, (5,), 3, 250).tolist(), torch.zeros(1, 4, 8))
    ids = torch.tensor(ctx.ids)
    S = len(ctx.ids)                                              # 1 + 6 + 66 + 5 + 66 = 144
    eng.reset()
    eng.prefill(emb[ids])
    _, _, kv = O.llama_forward(wd, dims, emb[ids].unsqueeze(0), torch.arange(S).unsqueeze(0))
    b, e = ctx.ids.index(ids_img[0]), ctx.ids.index(ids_img[-1])
    keep, new_sink = O.sink_evict_indices(S, b, e, 0, True)
    kv_len = ctx.evict_sink(eng, S)
    assert kv_len == len(keep) == eng.lengths()[0] and ctx.sink_len == new_sink
    window_len = len(ctx.ids)
    eng.set_lengths(kv_len, window_len)                           # new queries: window-relative positions
    new_ids = synth.randint(72, (9,), 3, 250)
    hid = eng.prefill(emb[new_ids], want_hidden=True)
    past = [(k[:, :, keep], v[:, :, keep]) for k, v in kv]
    pos = torch.arange(window_len, window_len + 9).unsqueeze(0)
    _, ref_hid, _ = O.llama_forward(wd, dims, emb[new_ids].unsqueeze(0), pos, past)
    assert rel(hid, ref_hid[0]) < 1e-4


@pytest.mark.parametrize("parity", [False, True])
def test_gen_george_driver_synthetic_tiny(tmp_path, parity):
    """The story driver end to end (ViT -> agent.generate -> adapter.generate -> JPEG, window eviction) on
    tiny random-weight models: 4 steps with a 2-image window (forces two recompute evictions); ``--parity`` = the
    reference's string-level prompt bookkeeping (decode -> scrub -> re-tokenise, '[INST]' skip) instead of ids."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "seed-story_amd"))
    out = subprocess.run([sys.executable, "-m", "src.inference.gen_george", "--synthetic", "--tiny", "--steps", "4",
                          "--window", "2", "--diffusion-steps", "2", "--image-size", "64", "--caption-tokens", "5",
                          "--out", str(tmp_path)] + (["--parity"] if parity else []),
                         env=env, capture_output=True, text=True, timeout=600,
                         cwd=os.path.join(root, "seed-story_amd"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    folder = tmp_path / "val_0"
    assert sorted(p.name for p in folder.glob("ori_*.jpg")) == ["ori_01.jpg", "ori_02.jpg", "ori_03.jpg", "ori_04.jpg"]
    lens = [int(l.split(",")[1].strip(" )\n")) for l in open(folder / "token.txt")]
    # id level: the 5-token caption in front of the generated <img> is kept.  --parity: like the reference, the whole
    # decoded text minus the <...> tokens is appended (random weights never emit EOS: 500 generated - 66 image tokens)
    # (the tiny random-weight model may or may not emit EOS inside the 500 tokens — 1 of 320 ids per step: the caption length
    # is then whatever was generated before it, at most 500 - 66)
    assert lens[0] == 1 + 6 + 66
    if parity:
        cap = lens[1] - lens[0] - 66
        assert 0 <= cap <= 500 - 66, lens
    else:
        cap = 5
        assert lens[1] == lens[0] + cap + 66, lens
    assert lens[3] <= lens[1] + (500 - 66 if parity else cap) + 66, lens                              # window holds


def test_vis_george_sink_driver_synthetic_tiny(tmp_path):
    """Story-visualisation driver with the multimodal attention sink live on the KV slab: 5 steps, 2-image window
    (three sink evictions), continuation through past_key_values / kv_cache_head."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "seed-story_amd"))
    out = subprocess.run([sys.executable, "-m", "src.inference.vis_george_sink", "--synthetic", "--tiny", "--steps", "5",
                          "--window", "2", "--diffusion-steps", "2", "--image-size", "64", "--caption-tokens", "7",
                          "--out", str(tmp_path)], env=env, capture_output=True, text=True, timeout=600,
                         cwd=os.path.join(root, "seed-story_amd"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = open(tmp_path / "val_0" / "token.txt").read().strip().split("\n")
    assert len(lines) == 5
    sinks = [int(l.rsplit("sink:", 1)[1]) for l in lines]
    assert sinks[:2] == [0, 0] and sinks[2] == 28 and sinks[3] == 52 and sinks[4] == 76     # 4 + 24 per evicted image
    assert len(list((tmp_path / "val_0").glob("ori_*.jpg"))) == 5


def test_generate_past_key_values_kv_cache_head_numeric(golden):
    """``LlamaForCausalLM.generate(past_key_values=…)`` with ``use_kv_cache_head`` / ``kv_cache_head``
    (reference modeling_llama_xformer.py:676-678 attributes, :804-826 ``prepare_inputs_for_generation``: rows
    ``[kv_cache_head:]`` of the new ``input_ids`` are fed against the cached prefix, ``position_ids`` =
    ``cumsum(mask)-1`` sliced the same way).

    (1) cached continuation of a second round == the oracle's from-scratch greedy run over the whole sequence
        (ids exactly; hidden rows, the regressed positions and the final cache within fp32 tolerance);
    (2) continuation from a SLICED cache (the multimodal attention sink: keys keep the RoPE phase they were cached
        with, new rows are numbered from the trimmed length) == the oracle's ``llama_forward`` on that same past."""
    from src.models_clm.generation import AutoImageTokenGenerationProcessor
    from src.models_clm.modeling_llama_xformer import LlamaConfig, LlamaForCausalLM
    g, meta = golden
    d = meta["LLAMA"]
    dims = O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    cfg = LlamaConfig(hidden_size=d["hidden"], intermediate_size=d["inter"], num_hidden_layers=d["n_layers"],
                      num_attention_heads=d["n_heads"], vocab_size=d["vocab"])
    llm = LlamaForCausalLM(cfg)
    llm.load_state_dict(wd, strict=False)
    llm = llm.to(DEV)
    llm.cache_cap, llm.max_new, llm.max_prefill_rows = 256, 64, 64
    img_ids = _img_ids(meta)
    proc = [AutoImageTokenGenerationProcessor(tokenizer=_Tok(img_ids))]
    emb = wd["model.embed_tokens.weight"]

    ids1 = synth.randint(41, (1, 23), 3, img_ids[0] - 1)
    forced1 = synth.randint(42, (6,), 3, img_ids[0] - 1).tolist()
    llm.use_kv_cache_head, llm.kv_cache_head, llm.past_key_values = True, None, None
    out1 = llm.generate(input_ids=ids1, inputs_embeds=emb[ids1].to(DEV), logits_processor=proc, max_new_tokens=6,
                        forced_tokens=forced1)
    assert out1.sequences[0, 23:].tolist() == forced1
    assert llm.kv_cache_head == 23 + 5 and llm.past_key_values[0][0].shape[2] == 28      # the last token is not cached

    # round 2: previous sequence + 9 new prompt tokens; 4 forced then 5 free-running greedy tokens
    extra = synth.randint(43, (1, 9), 3, img_ids[0] - 1)
    ids2 = torch.cat([out1.sequences.cpu(), extra], dim=1)
    forced2 = synth.randint(44, (4,), 3, img_ids[0] - 1).tolist()
    S2 = ids2.shape[1]
    out2 = llm.generate(input_ids=ids2, inputs_embeds=emb[ids2].to(DEV), logits_processor=proc, max_new_tokens=9,
                        past_key_values=llm.past_key_values, forced_tokens=forced2)
    gen_o, hid_o, _, kv_o = O.greedy_generate(wd, dims, ids2, emb[ids2], img_ids, 9, forced=forced2)
    assert out2.sequences[0, S2:].tolist() == gen_o
    assert out2.hidden_states[0][0].shape == (1, S2 - 28, d["hidden"])                    # only rows [kv_cache_head:] were fed
    hid_h = torch.cat([h[0].reshape(1, -1) for h in out2.hidden_states[1:]])
    assert rel(hid_h, hid_o) < 1e-4
    n_kv = S2 + len(gen_o) - 1
    assert llm.kv_cache_head == n_kv and llm.past_key_values[0][0].shape[2] == n_kv
    for l in range(d["n_layers"]):
        assert rel(llm.past_key_values[l][0], kv_o[l][0]) < 1e-4
        assert rel(llm.past_key_values[l][1], kv_o[l][1]) < 1e-4

    # (2) sliced cache: drop cached rows [5, 17) (an evicted image span), continue with 7 new rows
    keep = torch.cat([torch.arange(0, 5), torch.arange(17, n_kv)])
    past_sl_o = [(k[:, :, keep].clone(), v[:, :, keep].clone()) for (k, v) in kv_o]
    past_sl = tuple((k[:, :, keep.to(DEV)].clone(), v[:, :, keep.to(DEV)].clone()) for (k, v) in llm.past_key_values)
    head = keep.numel()
    seq_all = torch.cat([ids2[0], torch.tensor(gen_o)])
    ids3 = torch.cat([seq_all[keep], seq_all[n_kv:n_kv + 1], synth.randint(45, (6,), 3, img_ids[0] - 1)]).unsqueeze(0)
    S3 = ids3.shape[1]
    llm.kv_cache_head = head
    forced3 = synth.randint(46, (3,), 3, img_ids[0] - 1).tolist()
    out3 = llm.generate(input_ids=ids3, inputs_embeds=emb[ids3].to(DEV), logits_processor=proc, max_new_tokens=3,
                        past_key_values=past_sl, forced_tokens=forced3)
    pos = torch.arange(head, S3).unsqueeze(0)
    _, last_o, kv3 = O.llama_forward(wd, dims, emb[ids3[:, head:]], pos, past_sl_o)
    assert rel(out3.hidden_states[0][0][0], last_o[0]) < 1e-4
    x = emb[torch.tensor([[forced3[0]]])]
    _, last_o2, kv3 = O.llama_forward(wd, dims, x, torch.tensor([[S3]]), kv3)
    assert rel(out3.hidden_states[1][0].reshape(-1), last_o2[0, -1]) < 1e-4
    assert llm.kv_cache_head == S3 + 2


def test_forward_call_and_manual_greedy_loop(golden):
    """``LlamaForCausalLM.forward`` (reference :703-794) + ``prepare_inputs_for_generation`` (:796-852) driven the way HF's
    greedy search drives them: logits of ALL rows vs the oracle, cache growth, ``kv_cache_head``
    bookkeeping, and a hand-rolled loop (prepare -> forward -> image-token processor -> argmax) that reproduces both
    ``generate()`` and the oracle's greedy ids."""
    from src.models_clm.generation import AutoImageTokenGenerationProcessor
    from src.models_clm.modeling_llama_xformer import LlamaConfig, LlamaForCausalLM
    g, meta = golden
    d = meta["LLAMA"]
    dims = O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    llm = LlamaForCausalLM(LlamaConfig(hidden_size=d["hidden"], intermediate_size=d["inter"], num_hidden_layers=d["n_layers"],
                                       num_attention_heads=d["n_heads"], vocab_size=d["vocab"]))
    llm.load_state_dict(wd, strict=False)
    llm = llm.to(DEV).eval()
    llm.cache_cap, llm.max_new, llm.max_prefill_rows = 256, 32, 64
    img_ids = _img_ids(meta)
    emb = wd["model.embed_tokens.weight"]
    ids = synth.randint(61, (1, 21), 3, img_ids[0] - 1)
    # (1) one call, all-row logits; then a 4-row continuation against the returned cache
    llm.use_kv_cache_head, llm.kv_cache_head = True, None
    out = llm(input_ids=ids.to(DEV), inputs_embeds=emb[ids].to(DEV), output_hidden_states=True)
    lo, last_o, kv_o = O.llama_forward(wd, dims, emb[ids], torch.arange(21).unsqueeze(0), None)
    assert out.logits.shape == (1, 21, d["vocab"]) and rel(out.logits, lo) < 1e-4
    assert rel(out.hidden_states[-1], last_o) < 1e-4 and out.past_key_values[0][0].shape[2] == 21
    assert llm.kv_cache_head == 21 and llm.past_key_values is out.past_key_values
    more = synth.randint(62, (1, 4), 3, img_ids[0] - 1)
    out2 = llm(input_ids=more.to(DEV), past_key_values=out.past_key_values, position_ids=torch.arange(21, 25).unsqueeze(0))
    lo2, _, _ = O.llama_forward(wd, dims, emb[more], torch.arange(21, 25).unsqueeze(0), kv_o)
    assert rel(out2.logits, lo2) < 1e-4 and llm.kv_cache_head == 25 and out2.past_key_values[1][1].shape[2] == 25
    # labels: the training-side call (modeling_llama_xformer.py:761-772), forward only
    lab = ids.clone()
    lab[0, :5] = -100
    ol = llm(input_ids=ids.to(DEV), labels=lab.to(DEV), output_hidden_states=True)
    ref_loss = torch.nn.functional.cross_entropy(lo[0, :-1], lab[0, 1:])
    assert abs(float(ol.loss) - float(ref_loss)) < 1e-4 * float(ref_loss) and rel(ol.logits, lo) < 1e-4
    assert ol.past_key_values is None and ol["loss"] is ol.loss
    # (2) the greedy loop by hand (use_kv_cache_head = False, gen_george.py:165)
    llm.use_kv_cache_head, llm.kv_cache_head = False, None
    proc = AutoImageTokenGenerationProcessor(tokenizer=_Tok(img_ids))
    seq = ids.to(DEV)
    mask = torch.ones_like(seq)
    past, gen = None, []
    for step in range(8):
        inp = llm.prepare_inputs_for_generation(seq, past_key_values=past, attention_mask=mask,
                                                inputs_embeds=emb[ids].to(DEV) if step == 0 else None, use_cache=True)
        kw = {k: v for k, v in inp.items() if k in ("input_ids", "inputs_embeds", "position_ids", "past_key_values")}
        if kw.get("inputs_embeds") is None:
            kw.pop("inputs_embeds", None)
        o = llm(**kw)
        scores = proc(seq, o.logits[:, -1, :].float().contiguous())
        tok = int(scores.argmax(-1))
        gen.append(tok)
        seq = torch.cat([seq, torch.tensor([[tok]], device=DEV)], dim=1)
        mask = torch.ones_like(seq)
        past = o.past_key_values
    gen_o, _, _, _ = O.greedy_generate(wd, dims, ids, emb[ids], img_ids, 8)
    assert gen == gen_o[:8]
    g2 = llm.generate(input_ids=ids, inputs_embeds=emb[ids].to(DEV), logits_processor=[proc], max_new_tokens=8)
    assert g2.sequences[0, 21:].tolist() == gen_o[:8]
    with pytest.raises(ValueError):
        llm.generate(input_ids=ids, inputs_embeds=emb[ids].to(DEV), logits_processor=[proc], max_new_tokens=33)   # > max_new


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_img_block_decode_equals_token_by_token(golden, dtype, tol):
    """The forced image-token run behind ``<img>`` fed as one batched continuation (``generate_img_block`` /
    ``generate_batch_img_block``, ss_llama_set_stop_id) vs the token-by-token loop: identical ids, hidden rows and KV cache
    within accumulation-order noise — single slot (two images in one call, budget ending inside a block) and four slots
    that reach ``<img>`` at different times."""
    from seedstory.llama import LlamaEngine
    g, meta = golden
    img = _img_ids(meta)
    d = meta["LLAMA"]
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], dtype=dtype)
    eng = LlamaEngine(wd, hidden=d["hidden"], n_heads=d["n_heads"], n_layers=d["n_layers"], inter=d["inter"], vocab=d["vocab"],
                      dtype=dtype, device=DEV, cache_cap=256, max_new=192, max_prefill_rows=64, img_ids=img)   # ring >= every n_steps below
    emb = wd["model.embed_tokens.weight"]
    prompt = synth.randint(71, (19,), 3, img[0] - 1)

    def run(block, n_steps, forced):
        eng.reset()
        eng.prefill(emb[prompt])
        if block:
            ids, hid = eng.generate_img_block(n_steps, int(prompt[-1]), forced)
        else:
            n = eng.generate(n_steps, int(prompt[-1]), forced)
            ids, hid = eng.gen_ids[:n].tolist(), eng.hidden_rows[:n - 1].clone()
        return ids, hid.float().cpu(), eng.lengths(), eng.k_cache[:, :, :eng.lengths()[0]].float().cpu()

    cap = synth.randint(72, (5,), 3, img[0] - 1).tolist()
    for n_steps, forced in ((100, cap + [img[0]]),                                   # caption, <img>, block, free tail
                            (5 + 66 + 4 + 66 + 3, cap + img + cap[:3] + [img[0]]),   # two images in one call
                            (5 + 1 + 30, cap + [img[0]])):                           # budget ends inside the block
        a, b = run(False, n_steps, forced), run(True, n_steps, forced)
        if dtype == torch.float32:
            assert a[0] == b[0]
        else:                       # bf16: the forced part is exact; a free-running tail may flip on a near-tie
            k = 5 + 66 if n_steps >= 5 + 66 else len(a[0])
            assert a[0][:k] == b[0][:k]
        n = min(len(a[0]), len(b[0])) if dtype != torch.float32 else len(a[0])
        if a[0] == b[0]:
            assert a[2] == b[2] and a[1].shape == b[1].shape
            assert rel(b[1], a[1]) < tol and rel(b[3], a[3]) < tol
    # four slots, captions of different lengths (slot 2 never opens an image, slot 3 is forced through EOS)
    e4 = LlamaEngine(wd, hidden=d["hidden"], n_heads=d["n_heads"], n_layers=d["n_layers"], inter=d["inter"], vocab=d["vocab"],
                     dtype=dtype, device=DEV, cache_cap=256, max_new=128, max_prefill_rows=64, img_ids=img, n_seq=4)
    prompts = [synth.randint(80 + b, (11 + 3 * b,), 3, img[0] - 1) for b in range(4)]
    forced = [cap[:2] + [img[0]], cap + cap + [img[0]], cap + cap + cap, cap[:3] + [2]]
    lasts = [int(p[-1]) for p in prompts]

    def prefill_all():
        for b in range(4):
            e4.select(b).reset()
            e4.select(b).prefill(emb[prompts[b]])
    prefill_all()
    ns = e4.generate_batch(80, lasts, forced)
    seq = [(e4.select(b).gen_ids[:ns[b]].tolist(), e4.select(b).hidden_rows[:max(ns[b] - 1, 0)].float().cpu()) for b in range(4)]
    prefill_all()
    ids, hids = e4.generate_batch_img_block(80, lasts, forced)
    for b in range(4):
        fixed = len(forced[b]) + (65 if forced[b][-1] == img[0] else 0)
        assert ids[b][:fixed] == seq[b][0][:fixed], b
        if ids[b] == seq[b][0]:
            assert rel(hids[b], seq[b][1]) < tol, b
        else:
            assert dtype != torch.float32, b
    assert ids[3] == forced[3] and len(ids[0]) == 80 and len(ids[2]) == 80


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("stack_rows", [256, 64])
def test_prefill_batch_equals_per_slot_prefill(golden, dtype, tol, stack_rows):
    """``ss_llama_prefill_batch`` (the slots' rows stacked, every projection once: weights streamed once per call) == one
    ``ss_llama_prefill`` per slot: prompt prefill with ragged lengths, then a stacked continuation against the caches (one
    slot sitting out), hidden rows / last-row logits / KV / lengths per slot.  ``stack_rows`` = the engine's
    max_prefill_rows: 256 takes the native stacked path (97 and 83 rows), 64 the slot-by-slot fallback of the wrapper."""
    from seedstory.llama import LlamaEngine
    g, meta = golden
    d = meta["LLAMA"]
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], dtype=dtype)
    kw = dict(hidden=d["hidden"], n_heads=d["n_heads"], n_layers=d["n_layers"], inter=d["inter"], vocab=d["vocab"], dtype=dtype,
              device=DEV, cache_cap=256, max_new=128, img_ids=_img_ids(meta))
    e4 = LlamaEngine(wd, max_prefill_rows=stack_rows, n_seq=4, **kw)
    e1s = [LlamaEngine(wd, max_prefill_rows=128, **kw) for _ in range(4)]
    emb = wd["model.embed_tokens.weight"]
    lens = [37, 21, 30, 9]
    prompts = [synth.randint(900 + b, (lens[b],), 3, 250) for b in range(4)]
    for b in range(4):
        e4.select(b).reset()
    hb = e4.prefill_batch([emb[p] for p in prompts], want_hidden=True)
    conts = [synth.randint(910 + b, (n,), 3, 250) if n else None for b, n in enumerate([12, 0, 66, 5])]
    hc = e4.prefill_batch([None if c is None else emb[c] for c in conts], want_hidden=True)
    for b in range(4):
        ref = e1s[b]
        h1 = ref.prefill(emb[prompts[b]], want_hidden=True)
        assert rel(hb[b], h1) < tol
        n = lens[b]
        if conts[b] is not None:
            h2 = ref.prefill(emb[conts[b]], want_hidden=True)
            assert rel(hc[b], h2) < tol
            n += len(conts[b])
        else:
            assert hc[b] is None
        e4.select(b)
        assert e4.lengths() == ref.lengths() == (n, n)
        assert rel(e4.logits, ref.logits) < tol
        assert rel(e4.k_cache[:, :, :n], ref.k_cache[:, :, :n]) < tol and rel(e4.v_cache[:, :, :n], ref.v_cache[:, :, :n]) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("n_seq", [3, 8])
def test_prefill_batch_uniform_rows_one_attention_launch(golden, dtype, tol, n_seq):
    """The lock-step image-token block: every slot feeds the SAME number of rows against caches of DIFFERENT lengths — the
    stacked forward then issues one ragged attention launch per layer (ss_attention_ragged) instead of one per slot; results
    equal the per-slot engine, and equal the per-slot-launch path of the same engine bit for bit (knob llama_batched_attn)."""
    from seedstory import _lib
    from seedstory.llama import LlamaEngine
    g, meta = golden
    d = meta["LLAMA"]
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], dtype=dtype)
    kw = dict(hidden=d["hidden"], n_heads=d["n_heads"], n_layers=d["n_layers"], inter=d["inter"], vocab=d["vocab"], dtype=dtype,
              device=DEV, cache_cap=256, max_new=128, img_ids=_img_ids(meta))
    emb = wd["model.embed_tokens.weight"]
    lens = [37, 21, 30, 9, 50, 44, 12, 26][:n_seq]
    prompts = [synth.randint(930 + b, (lens[b],), 3, 250) for b in range(n_seq)]
    conts = [synth.randint(940 + b, (66,), 3, 250) for b in range(n_seq)]
    outs = []
    for batched in (1, 0):
        _lib.set_tuning("llama_batched_attn", batched)
        try:
            e = LlamaEngine(wd, max_prefill_rows=66 * n_seq, n_seq=n_seq, **kw)
            e.prefill_batch([emb[p] for p in prompts])
            hc = e.prefill_batch([emb[c] for c in conts], want_hidden=True)
            outs.append((e, hc))
        finally:
            _lib.set_tuning("llama_batched_attn", 1)
    for b in range(n_seq):
        ref = LlamaEngine(wd, max_prefill_rows=128, **kw)
        ref.prefill(emb[prompts[b]])
        h2 = ref.prefill(emb[conts[b]], want_hidden=True)
        n = lens[b] + 66
        for e, hc in outs:
            e.select(b)
            assert e.lengths() == ref.lengths() == (n, n)
            assert rel(hc[b], h2) < tol and rel(e.logits, ref.logits) < tol
            assert rel(e.k_cache[:, :, :n], ref.k_cache[:, :, :n]) < tol
        assert torch.equal(outs[0][1][b], outs[1][1][b])


class _ProcStub:
    """Carries the 66 image-token ids the way AutoImageTokenGenerationProcessor does (generation.py:17)."""

    def __init__(self, ids):
        self.img_ids_list = list(ids)


@pytest.mark.parametrize("case", ["free", "eos"])
def test_llm_generate_vs_real_hf_generate(case):
    """``LlamaForCausalLM.generate`` (device greedy loop + image-token block decode) against a REAL
    ``GenerationMixin.generate`` run with the reference's logits processor (tests/golden/greedy_hf.*, written by
    oracle/make_golden_greedy.py with the installed transformers 5.15; the reference pins 4.34): generated ids — the 65
    processor-forced tokens and the free-running tail — the EOS stop, and the last-layer hidden rows."""
    import json
    import os
    from safetensors.torch import load_file
    from src.models_clm.modeling_llama_xformer import LlamaConfig, LlamaForCausalLM
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = load_file(os.path.join(root, "greedy_hf.safetensors"))
    with open(os.path.join(root, "greedy_hf.json")) as f:
        meta = json.load(f)
    d = meta["LLAMA"]
    eos = 2 if case == "free" else meta["eos_case_id"]
    cfg = LlamaConfig(hidden_size=d["hidden"], intermediate_size=d["inter"], num_hidden_layers=d["n_layers"],
                      num_attention_heads=d["n_heads"], vocab_size=d["vocab"], eos_token_id=eos)
    llm = LlamaForCausalLM(cfg)
    missing, unexpected = llm.load_state_dict(synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"],
                                                                  d["vocab"]), strict=False)
    assert not missing and not unexpected
    llm.cache_cap, llm.max_new, llm.max_prefill_rows = 256, 128, 128
    llm.use_kv_cache_head = False
    llm = llm.to(DEV)
    img_ids = list(range(meta["IMG_IDS"][0], meta["IMG_IDS"][1] + 1))
    S = g["input_ids"].shape[1]
    out = llm.generate(input_ids=g["input_ids"], inputs_embeds=g["inputs_embeds"].to(DEV), logits_processor=[_ProcStub(img_ids)],
                       max_new_tokens=meta["MAX_NEW"], output_hidden_states=True, return_dict_in_generate=True)
    gen = out.sequences[0][S:].tolist()
    assert gen == g[case + ".generate_ids"].tolist()
    assert len(out.hidden_states) == len(gen) and out.hidden_states[0][-1].shape == (1, S, d["hidden"])
    last = torch.cat([hs[-1] for hs in out.hidden_states], dim=1)[0, S:]                  # models.py:182-184
    assert rel(last, g[case + ".hidden"]) < 1e-4
