"""TEST INFRASTRUCTURE ONLY — goldens of the TRAINING-SIDE FORWARD (forward only) produced by the REAL
reference class ``ContinuousLVLM`` (``/root/reference/src/models_clm/models.py:20-96``) around the real
``LlamaForCausalLM`` (``modeling_llama_xformer.py``) and the real ``Resampler`` (``src/models/qwen_visual.py``), imported
behind ``oracle/ref_shims.py`` and loaded with the seeded synthetic weights of ``oracle/synth.py``.

    python oracle/make_golden_forward.py          (build container; needs /root/reference; ~3 minutes on 8 cores)

Two configurations, fp32 and bf16 each:

  tiny   hidden 256 / 2 heads / 2 layers / inter 512 / vocab 320, input resampler grid 4 (16 LLM tokens per image), output
         resampler grid 8 (64 queries): batch of 3 right-padded sequences, one comprehension image, two generation targets,
         one sequence without any image -> every tensor of the result is stored;
  full   hidden 4096 / 32 heads / inter 11008 / vocab 32066 (2 layers) with the real-size resamplers (256 -> 64 in,
         64 -> 256 out): batch of 2 -> the three losses + every 16th row of ``recon_image_embeds``.

The oracle restatement ``seedstory_oracle.lvlm_forward`` is asserted against the reference output here (fp32 <= 5e-6
relative on the losses / features; bf16 within the reference's own bf16-vs-fp32 distance), i.e. the row is PINNED.
Also stored: the no-image branch (reference :41-47, 58-62, 82-90 — placeholder tensors times 0.0) on the tiny batch.
Fixture: ``tests/golden/forward_f4.safetensors``.
"""
import importlib
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

TINY = dict(hidden=256, n_heads=2, n_layers=2, inter=512, vocab=320, grid_in=4, grid_out=8, res_heads=2,
            img_ids=list(range(254, 320)), seed=11, sq=120, text_hi=250)
FULL = dict(hidden=4096, n_heads=32, n_layers=2, inter=11008, vocab=32066, grid_in=8, grid_out=16, res_heads=32,
            img_ids=list(range(32000, 32066)), seed=11, sq=160, text_hi=32000)


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def weights(c, dtype):
    wd = synth.llama_weights(c["seed"], c["hidden"], c["n_heads"], c["n_layers"], c["inter"], c["vocab"], dtype=dtype)
    wd.update(synth.resampler_weights(21, "input_resampler.", c["grid_in"], c["hidden"], dtype=dtype))
    wd.update(synth.resampler_weights(22, "output_resampler.", c["grid_out"], c["hidden"], dtype=dtype))
    return wd


def batch(c, dtype, bz):
    """Right-padded training batch in the reference data pipeline's conventions: ``<img>`` + placeholders + ``</img>`` per image;
    comprehension images are spliced at ``ids_cmp_mask``, generation targets read at ``ids_gen_mask``; labels are the ids
    with -100 on the first segment, the comprehension placeholders and the padding."""
    ids = c["img_ids"]
    boi, eoi = ids[0], ids[-1]
    nq_in, nq_out = c["grid_in"] ** 2, 64
    sq, H = c["sq"], c["hidden"]
    n_tok_img = c["grid_out"] ** 2            # tokens of a ViT-space image feature = output resampler queries
    text = lambda seed, n: synth.randint(seed, (n,), 3, c["text_hi"]).tolist()  # noqa: E731
    rows, cmp_spans, gen_spans, n_valid = [], [], [], []
    # sequence 0: text, comprehension image, text, generated image
    r = [1] + text(70, 9)
    cmp_spans.append((0, len(r) + 1, nq_in))
    r += [boi] + ids[1:1 + nq_in] + [eoi] + text(71, 7)
    gen_spans.append((0, len(r) + 1, nq_out))
    r += [boi] + ids[1:65] + [eoi, 2]
    rows.append(r)
    if bz > 1:      # sequence 1: text, generated image
        r = [1] + text(72, 14)
        gen_spans.append((1, len(r) + 1, nq_out))
        r += [boi] + ids[1:65] + [eoi] + text(73, 3) + [2]
        rows.append(r)
    if bz > 2:      # sequence 2: text only
        rows.append([1] + text(74, 30) + [2])
    input_ids = torch.zeros(bz, sq, dtype=torch.long)
    attention_mask = torch.zeros(bz, sq, dtype=torch.long)
    labels = torch.full((bz, sq), -100, dtype=torch.long)
    for b, r in enumerate(rows):
        assert len(r) <= sq
        input_ids[b, :len(r)] = torch.tensor(r)
        attention_mask[b, :len(r)] = 1
        labels[b, 6:len(r)] = torch.tensor(r[6:])                      # the first 6 tokens are "prompt": not scored
    ids_cmp_mask = torch.zeros(bz, sq, dtype=torch.bool)
    ids_gen_mask = torch.zeros(bz, sq, dtype=torch.bool)
    for b, s0, n in cmp_spans:
        ids_cmp_mask[b, s0:s0 + n] = True
        labels[b, s0:s0 + n] = -100
    for b, s0, n in gen_spans:
        ids_gen_mask[b, s0:s0 + n] = True
    n_img = len(cmp_spans) + len(gen_spans)
    image_embeds = synth.normal_like(51, (n_img, n_tok_img, H), 1.0, dtype=dtype)
    embeds_cmp_mask = torch.tensor([True] + [False] * len(gen_spans))
    embeds_gen_mask = torch.tensor([False] + [True] * len(gen_spans))
    return dict(input_ids=input_ids, attention_mask=attention_mask, labels=labels, image_embeds=image_embeds,
                embeds_gen_mask=embeds_gen_mask, embeds_cmp_mask=embeds_cmp_mask, ids_gen_mask=ids_gen_mask,
                ids_cmp_mask=ids_cmp_mask)


def build_reference(c, wd, dtype, llama_mod, qwen_mod, models_mod):
    from transformers import LlamaConfig
    cfg = LlamaConfig(hidden_size=c["hidden"], intermediate_size=c["inter"], num_hidden_layers=c["n_layers"],
                      num_attention_heads=c["n_heads"], vocab_size=c["vocab"], max_position_embeddings=4096, rms_norm_eps=1e-5)
    m = llama_mod.LlamaForCausalLM(cfg).eval()
    m.load_state_dict({k: v for k, v in wd.items() if "resampler" not in k}, strict=False)
    m = m.to(dtype)
    m.use_kv_cache_head = False
    H = c["hidden"]
    rin = qwen_mod.Resampler(grid_size=c["grid_in"], embed_dim=H, num_heads=c["res_heads"], kv_dim=H).eval()
    rin.load_state_dict({k[len("input_resampler."):]: v for k, v in wd.items() if k.startswith("input_resampler.")})
    rout = qwen_mod.Resampler(grid_size=c["grid_out"], embed_dim=H, num_heads=c["res_heads"], kv_dim=H).eval()
    rout.load_state_dict({k[len("output_resampler."):]: v for k, v in wd.items() if k.startswith("output_resampler.")})
    return models_mod.ContinuousLVLM(llm=m, input_resampler=rin.to(dtype), output_resampler=rout.to(dtype),
                                     lm_loss_scale=1.0, rec_loss_scale=1.0).eval()


def run(c, tag, bz, row_stride, out, mods, report):
    llama_mod, qwen_mod, models_mod = mods
    dims = O.LlamaDims(c["hidden"], c["n_heads"], c["n_layers"], c["inter"], c["vocab"])
    ref32 = None
    for dtype, dtag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        wd = weights(c, dtype)
        agent = build_reference(c, wd, dtype, llama_mod, qwen_mod, models_mod)
        b = batch(c, dtype, bz)
        with torch.no_grad():
            ref = agent(b["input_ids"], b["attention_mask"], b["labels"], b["image_embeds"], b["embeds_gen_mask"],
                        b["embeds_cmp_mask"], b["ids_gen_mask"], b["ids_cmp_mask"], return_recon_image_embeds=True)
            mine = O.lvlm_forward(wd, dims, b["input_ids"], b["labels"], b["image_embeds"], b["embeds_gen_mask"],
                                  b["embeds_cmp_mask"], b["ids_gen_mask"], b["ids_cmp_mask"], n_heads_resampler=c["res_heads"])
        errs = {k: abs(float(mine[k]) - float(ref[k])) / (abs(float(ref[k])) + 1e-30) for k in ("total_loss", "lm_loss", "rec_loss")}
        errs["recon_image_embeds"] = rel(mine["recon_image_embeds"], ref["recon_image_embeds"])
        if dtype == torch.float32:
            ref32 = ref
            assert max(errs.values()) <= 5e-6, errs          # fp32 summation-order noise at hidden 4096: 2.5e-6
        else:
            gap = rel(ref["recon_image_embeds"], ref32["recon_image_embeds"])
            assert errs["recon_image_embeds"] <= 1.5 * gap + 2e-3, (errs, gap)
            for k in ("total_loss", "lm_loss", "rec_loss"):
                g = abs(float(ref[k]) - float(ref32[k])) / abs(float(ref32[k]))
                assert errs[k] <= 1.5 * g + 2e-2, (k, errs[k], g)
            report[tag + "_bf16_vs_f32_reference"] = {"recon": gap, **{k: abs(float(ref[k]) - float(ref32[k])) / abs(float(ref32[k]))
                                                                       for k in ("total_loss", "lm_loss", "rec_loss")}}
        report["%s_%s_oracle_vs_reference" % (tag, dtag)] = errs
        print("  %s %s: lm %.6f rec %.6f total %.6f | oracle vs reference %s" % (
            tag, dtag, float(ref["lm_loss"]), float(ref["rec_loss"]), float(ref["total_loss"]),
            {k: "%.1e" % v for k, v in errs.items()}), flush=True)
        t = "%s_%s." % (tag, dtag)
        for k in ("total_loss", "lm_loss", "rec_loss"):
            out[t + k] = ref[k].float().reshape(1)
        rec = ref["recon_image_embeds"].float()
        out[t + "recon_rows"] = rec.reshape(-1, rec.shape[-1])[::row_stride].contiguous()
        out[t + "recon_norm"] = rec.norm().reshape(1)
        if tag == "tiny":
            for k, v in b.items():
                out[t + "in." + k] = v.float() if v.dtype in (torch.bfloat16,) else (v.to(torch.int64) if v.dtype == torch.bool else v)
            # the branch without images: every placeholder term is multiplied by 0.0 in the reference
            torch.manual_seed(0)
            with torch.no_grad():
                r0 = agent(b["input_ids"], b["attention_mask"], b["labels"], None, None, None, None, None)
                m0 = O.lvlm_forward(wd, dims, b["input_ids"], b["labels"], None, None, None, None, None, n_heads_resampler=c["res_heads"])
            assert float(r0["rec_loss"]) == 0.0
            e0 = abs(float(m0["lm_loss"]) - float(r0["lm_loss"])) / abs(float(r0["lm_loss"]))
            assert e0 <= (2e-6 if dtype == torch.float32 else 2e-2), e0
            out[t + "noimg.lm_loss"] = r0["lm_loss"].float().reshape(1)
            out[t + "noimg.total_loss"] = r0["total_loss"].float().reshape(1)
        del agent, wd


def main():
    torch.set_num_threads(8)
    llama_mod, qwen_mod, gen_mod, ipa_mod = ref_shims.import_reference()
    models_mod = importlib.import_module("src.models_clm.models")
    assert ref_shims.REFERENCE_ROOT in models_mod.__file__
    os.makedirs(GOLD, exist_ok=True)
    out, report = {}, {}
    run(TINY, "tiny", 3, 1, out, (llama_mod, qwen_mod, models_mod), report)
    run(FULL, "full", 2, 16, out, (llama_mod, qwen_mod, models_mod), report)
    meta = {"generator": "oracle/make_golden_forward.py", "tiny": json.dumps({k: v for k, v in TINY.items() if k != "img_ids"}),
            "full": json.dumps({k: v for k, v in FULL.items() if k != "img_ids"}), "report": json.dumps(report)}
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLD, "forward_f4.safetensors"), metadata=meta)
    print("wrote", os.path.join(GOLD, "forward_f4.safetensors"), sum(v.numel() * v.element_size() for v in out.values()), "bytes")


if __name__ == "__main__":
    main()
