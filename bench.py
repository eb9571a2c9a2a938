#!/usr/bin/env python
"""bench.py — story-steps/sec of the MI355X-native SEED-Story hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (default = the configuration BASELINE.json's metric is quoted on: the 10-step StoryStream chunk of
configs[3] — full pipeline, story length 10, 8-image context window — run on ONE GPU per replica; SURVEY.md §8d
synthetic schedule):
LLaMA-2-7B-shaped MLLM (random N(0,0.02) weights, bf16) + Qwen ViT-G encode + learnable-query
image-feature regression + SDXL de-tokenizer (ResamplerXLV2 -> 30-step Euler, CFG 7.5, 1024x1024,
random-init SDXL-base UNet/VAE, bf16), stories of 10 steps; from step 8 on the context holds more than 8 images and
the oldest image-text pair is evicted "as released" (gen_george.py:235-239: prompt cut through the first </img>,
whole window re-prefilled), so prompts grow S = 115, 229, ..., 913, 913, 913.  ``--story-len 5`` is configs[2];
``--mllm-only`` runs configs[1] (3 pairs, no SDXL).  One *step* = one ``agent.generate`` of the reference
(gen_george.py:189/257): embed + splice the window's image features (input resampler over every
image in context) -> prefill of the whole prompt (S = 115 ... 913; "as released", no KV reuse;
``--kv-reuse`` switches to the 65-row continuation) -> 115 greedy decode iterations under the forced
token schedule (48 caption ids, ``<img>``, 64 image tokens + ``</img>`` forced by the reference's
logits processor, EOS) -> output resampler regression of the 64 hidden states to the 256x4096 image
feature -> ``adapter.generate`` (gen_george.py:210): ResamplerXLV2 conditioning + 30 x {UNet (batch 2,
CFG) + Euler update} + VAE decode to a uint8 1024x1024 image.  The first step of every story also
encodes the 448x448 input image with ViT-G (and the constant all-zeros negative image once).

``--stories-per-gpu S`` (default 8): S independent stories are resident on each GPU and advance in
lock-step — their decode iterations share ONE sweep of the 13.2 GB of LLaMA weights per token
(ss_llama_generate_batch, HBM bytes per generated token / S) and their S images are denoised together
(UNet batch 2S).  One bench *step* is then one lock-step round = S story-steps; every story still
computes exactly what a batch-1 run computes (tests/test_engine_gpu.py::
test_llama_slot_batched_decode_equals_single).  ``--stories-per-gpu 1`` is the reference's batch-1
latency configuration.

N > 1: one process per GPU, independent stories per rank (SURVEY §8e story-level replicas, no
data-path collective), weak scaling; value = story-steps of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "seed-story_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

H, NH, NL, INTER, VOCAB = 4096, 32, 32, 11008, 32066
IMG_IDS = list(range(32000, 32066))     # <img>, <img_00000..63>, </img>  (66 added tokens)
BOS, EOS = 1, 2
CAPTION = 48
STORY_LEN = 10
WINDOW = 8                               # images kept in context (gen_george.py:203 window_size)
T_GEN = CAPTION + 66 + 1                 # caption + image tokens + EOS = 115 decode iterations
CACHE_CAP = 1152                         # KV rows per story slot (--sink: + the sink prefix of every eviction of the story)
SINK = False                             # --sink: multimodal attention sink on the KV slab instead of evict-and-re-prefill


def prompt_len(step):
    """Prompt length at story step `step`: BOS + (48 caption + 66 image tokens) per pair in context, at most WINDOW pairs."""
    return 1 + 114 * min(step + 1, WINDOW)


def build_detokenizer(device, dtype, vit):
    from seedstory.diffusion import AutoencoderKL, EulerDiscreteScheduler, UNet2DConditionModel
    from src.models.discrete_models import DiscreteModleIdentity
    from src.models_ipa.adapter_modules import SDXLAdapter
    from src.models_ipa.resampler import ResamplerXLV2
    unet = UNet2DConditionModel().to(device=device, dtype=dtype).init_synthetic(4)
    vae = AutoencoderKL().to(device=device, dtype=dtype).init_synthetic(5)
    rs = ResamplerXLV2(dim=1024, depth=4, dim_head=64, heads=16, num_queries=64, embedding_dim=4096, output1_dim=768,
                       output2_dim=1280, ff_mult=4).to(device=device, dtype=dtype).init_synthetic(6)
    adapter = SDXLAdapter.from_pretrained(unet=unet, resampler=rs).eval()
    adapter.init_pipe(vae=vae, scheduler=EulerDiscreteScheduler(), visual_encoder=vit, image_transform=None,
                      discrete_model=DiscreteModleIdentity(), dtype=dtype, device=device)
    return adapter


def build_frontend(device, dtype):
    from src.models.qwen_visual import Resampler, VisionTransformerWithAttnPool
    rin = Resampler(grid_size=8, embed_dim=H, num_heads=32, kv_dim=H).to(device=device, dtype=dtype).init_synthetic(1)
    rout = Resampler(grid_size=16, embed_dim=H, num_heads=32, kv_dim=H).to(device=device, dtype=dtype).init_synthetic(2)
    vit = VisionTransformerWithAttnPool(image_size=448, patch_size=14, width=1664, layers=48, heads=16,
                                        mlp_ratio=4.9231, output_dim=4096).to(device=device, dtype=dtype)
    vit.init_synthetic(3)
    return rin, rout, vit


class Story:
    """Id-level context manager of one synthetic story (the string-level prompt surgery of
    gen_george.py:168-255 restated on token ids)."""

    def __init__(self, seed, device):
        g = torch.Generator().manual_seed(seed)
        self.g = g
        self.device = device
        self.image = torch.rand(1, 3, 448, 448, generator=g)
        mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
        std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)
        self.image = ((self.image - mean) / std).to(device)
        self.ids = [BOS] + torch.randint(3, 32000, (CAPTION,), generator=g).tolist() + IMG_IDS
        self.image_embeds = None
        self.step = 0
        self.evicted_last = False          # the previous step evicted an image: cached KV positions are stale
        self.sink_len = 0                  # --sink: KV entries in front of the live window (seedstory/story.py)
        self.sink_log = []                 # --sink: (rows kept, rows dropped) per eviction

    def forced(self):
        cap = torch.randint(3, 32000, (CAPTION,), generator=self.g).tolist()
        return cap + IMG_IDS + [EOS]


def advance_context(st, forced_ids, new_feat):
    """Context bookkeeping of one finished step on token ids (gen_george.py:224-243 does it on strings)."""
    st.image_embeds = torch.cat([st.image_embeds, new_feat], dim=0)       # gen_george.py:224
    st.ids = st.ids + list(forced_ids[:CAPTION]) + IMG_IDS                # prompt + text + image_tokens (:231)
    st.evicted_last = False
    while not SINK and st.image_embeds.shape[0] > WINDOW:                 # :235-239: cut through the first </img>,
        e = st.ids.index(IMG_IDS[-1])                                     # drop the oldest image, re-add BOS (:243)
        st.ids = [BOS] + st.ids[e + 1:]
        st.image_embeds = st.image_embeds[1:]
        st.evicted_last = True
    st.step += 1


def mllm_part(sts, eng, rin, rout, vit, kv_reuse):
    """The MLLM half of one multimodal step of every resident story (slot b of the engine = story sts[b]; all
    stories of a round are at the same step index): ``agent.generate`` of gen_george.py:189/257.  Advances the
    stories' context (ids, image features) and returns img_gen_feat [S,256,4096].
    ``eng`` may be a list of engines (groups of <= 4 lock-step slots over the SAME weight tensors — the decode GEMV
    sweeps the weights once per group of up to 4 stories): the groups run one after the other."""
    from seedstory import ops
    if isinstance(eng, (list, tuple)):
        feats, i = [], 0
        for e in eng:
            feats.append(mllm_part(sts[i:i + e.n_seq], e, rin, rout, vit, kv_reuse))
            i += e.n_seq
        assert i == len(sts)
        return feats[0] if len(feats) == 1 else torch.cat(feats, dim=0)
    dev = sts[0].device
    embs = [None] * eng.n_seq
    for b, st in enumerate(sts):
        eng.select(b)
        if st.step == 0:
            st.image_embeds = vit(st.image)                               # [1,256,4096]  gen_george.py:187-188
        if SINK and st.step > 0:
            # vis_george_sink.py:243-295 as intended (seedstory/story.py): the slab keeps [sink prefix | window]; the prompt
            # through the newest <img> is cached (previous prompt + the decoded caption + <img>), the 64 image rows + </img>
            # are re-fed with the regressed feature spliced in (65-row continuation); when more than WINDOW images are live
            # the oldest one is evicted ON THE SLAB: the first 4 positions + 12 rows around its <img> + 12 around its </img>
            # join the sink prefix, everything behind its </img> slides down (ss_llama_kv_gather) — no re-prefill.
            from seedstory.story import sink_keep_indices
            keep = len(st.ids) - 65
            kv_len = st.sink_len + keep
            eng.set_lengths(kv_len, keep)
            while st.image_embeds.shape[0] > WINDOW:
                bi = st.ids.index(IMG_IDS[0]) + st.sink_len
                ei = st.ids.index(IMG_IDS[-1]) + st.sink_len
                idx, new_sink = sink_keep_indices(kv_len, bi, ei, st.sink_len, st.sink_len == 0)
                eng.kv_gather(idx)
                st.sink_log.append((len(idx), kv_len - len(idx)))
                cut = st.ids.index(IMG_IDS[-1]) + 1
                st.ids = st.ids[cut:]
                st.image_embeds = st.image_embeds[1:]
                kv_len, st.sink_len = len(idx), new_sink
            keep = len(st.ids) - 65
            eng.set_lengths(kv_len, keep)                                 # new queries: window-relative positions
            ids = torch.tensor(st.ids[keep:], dtype=torch.int32, device=dev)
            emb = ops.gather_rows(eng.embed, ids)
            lm = rin(st.image_embeds[-1:])                                # only the newest image's rows are re-fed
            ops.scatter_rows_(emb, torch.arange(64, dtype=torch.int32, device=dev), lm.reshape(-1, H))
            embs[b] = emb
            continue
        ids = torch.tensor(st.ids, dtype=torch.int32, device=dev)
        emb = ops.gather_rows(eng.embed, ids)                             # models.py:127
        lm = rin(st.image_embeds)                                         # [Nimg,64,H]   models.py:133
        pos = [i + 1 for i, t in enumerate(st.ids) if t == IMG_IDS[0]]
        idx = torch.tensor([p + j for p in pos for j in range(64)], dtype=torch.int32, device=dev)
        ops.scatter_rows_(emb, idx, lm.reshape(-1, H))                    # models.py:135
        S = len(st.ids)
        if kv_reuse and st.step > 0 and not st.evicted_last:
            keep = S - 65                                                 # ... caption + <img> stay cached
            eng.set_lengths(keep, keep)
            embs[b] = emb[keep:]
        else:
            eng.reset()
            embs[b] = emb
    if len(sts) == 1:
        eng.select(0).prefill(embs[0])
    else:       # the prompts of the lock-step stories as ONE stacked prefill: layer weights streamed once per round
        eng.prefill_batch(embs)
    forced = [st.forced() for st in sts]
    for st, f in zip(sts, forced):
        st.last_forced = f                                                # (the slot ring broadcasts them: seedstory/parallel.py)
    e = CAPTION + 65                                                      # index of </img> in the generated ids
    if eng.img_block_enabled():
        # the decode loop stops at <img>; the 65 tokens the logits processor forces behind it are fed as ONE batched
        # continuation per story (same layers / attention / hidden rows / KV entries, weights streamed once), then the
        # loop resumes (seedstory/llama.py: generate_img_block).  T_GEN tokens per story either way.
        if len(sts) == 1:
            g1, h1 = eng.generate_img_block(T_GEN, sts[0].ids[-1], forced[0])
            gens, hids = [g1], [h1]
        else:
            gens, hids = eng.generate_batch_img_block(T_GEN, [st.ids[-1] for st in sts], forced)
        assert all(len(g) == T_GEN and g[e] == IMG_IDS[-1] for g in gens), [len(g) for g in gens]
        feats = torch.stack([h[e - 64:e] for h in hids]).contiguous()
    else:
        if len(sts) == 1:
            ns = [eng.generate(500, sts[0].ids[-1], forced[0])]               # max_new_tokens=500 (gen_george.py:194)
        else:
            ns = eng.generate_batch(500, [st.ids[-1] for st in sts], forced)  # the same loop, all slots per weight sweep
        assert all(n == T_GEN for n in ns), ns
        feats = torch.stack([eng.select(b).hidden_rows[e - 64:e] for b in range(len(sts))]).contiguous()   # models.py:197
    img_gen_feat = rout(feats)                                            # models.py:205  [S,256,4096]
    for b, st in enumerate(sts):
        advance_context(st, forced[b], img_gen_feat[b:b + 1])
    return img_gen_feat


def sdxl_part(sts, adapter, img_gen_feat, steps):
    """The de-tokenizer half: ``adapter.generate`` (gen_george.py:210) for the S images of the round.
    ``adapter`` may be a list of G de-tokenizer replicas (same weights values, separate activation / hipGraph state): the
    round's images are rendered as G independent batches on G HIP streams at once (one host thread each).  Independent
    kernels of the G forwards fill each other's prologue / epilogue / tail phases — two concurrent batch-8 UNet forwards
    cost 59.5 ms per forward against 63.5 ms alone (tools/unet_concurrent.py) — and the work per image is unchanged."""
    if isinstance(adapter, (list, tuple)):
        if len(adapter) == 1:
            return sdxl_part(sts, adapter[0], img_gen_feat, steps)
        import threading
        G = len(adapter)
        assert len(sts) % G == 0
        per = len(sts) // G
        if not RENDER_CONCURRENT:     # preparation round: one group after the other (each captures its UNet hipGraph, and a
            # capture must not coincide with another thread's device-wide synchronisation)
            return torch.cat([sdxl_part(sts[k * per:(k + 1) * per], adapter[k], img_gen_feat[k * per:(k + 1) * per], steps)
                              for k in range(G)], dim=0)
        dev = img_gen_feat.device
        cur = torch.cuda.current_stream(dev)
        streams = _render_streams(G, dev)
        outs, errs = [None] * G, []

        def work(k):
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(streams[k]):
                    outs[k] = sdxl_part(sts[k * per:(k + 1) * per], adapter[k], img_gen_feat[k * per:(k + 1) * per], steps)
                streams[k].synchronize()
            except BaseException as ex:       # surfaced by the caller
                errs.append(ex)
        for s_ in streams:
            s_.wait_stream(cur)
        ths = [threading.Thread(target=work, args=(k,)) for k in range(G)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]
        return torch.cat(outs, dim=0)
    imgs = adapter.generate(image_embeds=img_gen_feat, num_inference_steps=steps, output_type="pt")
    imgs = imgs.unsqueeze(0) if len(sts) == 1 else imgs
    for b, st in enumerate(sts):
        st.last_image = imgs[b]
    return imgs


_RENDER_STREAMS = {}
RENDER_CONCURRENT = True


def _render_streams(n, device):
    key = (n, str(device))
    if key not in _RENDER_STREAMS:
        _RENDER_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _RENDER_STREAMS[key]


def run_round(sts, eng, rin, rout, vit, kv_reuse, adapter=None, steps=30):
    """One multimodal step of every resident story, sequentially (MLLM half, then the render)."""
    img_gen_feat = mllm_part(sts, eng, rin, rout, vit, kv_reuse)
    if adapter is not None:
        sdxl_part(sts, adapter, img_gen_feat, steps)
    return img_gen_feat


def run_step(st, eng, rin, rout, vit, kv_reuse, adapter=None, steps=30):
    return run_round([st], eng, rin, rout, vit, kv_reuse, adapter, steps)


def cpu_baseline(with_sdxl=True, diffusion_steps=30, budget_s=240.0):
    """The oracle (CPU restatement of the reference, kind "port") timed on this box's host cores at FULL dimensions,
    one measurement per piece, composed by the step formula:

      * LLaMA-7B, all 32 layers + lm_head, bf16: one S=343 prefill and 2 decode tokens (the 32 layers cycle through
        4 distinct weight sets = 1.6 GB, so nothing is cache-resident; every layer's arithmetic is executed);
      * SDXL-base UNet, fp32 (bf16 has no fast CPU path), 128x128 latents, ONE forward at batch 1 (the CFG pair is
        two of these);
      * SDXL VAE decoder, fp32, 128x128 latents -> 1024x1024, one decode;
      * ViT-G (48 layers cycling 2 weight sets), one 448x448 image (first step of a story only: 1/STORY_LEN).

    story-step = prefill(mean S) + 115 tokens + diffusion_steps x 2 UNet forwards + 1 VAE decode + ViT/STORY_LEN.
    Pieces that would overrun `budget_s` are skipped and priced from the measured flop rate of the previous piece
    (the sample string says which)."""
    import seedstory_oracle as O
    t_start = time.perf_counter()
    threads = torch.get_num_threads()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    notes = []

    def rnd(*s):
        return (torch.randn(*s, generator=g) * 0.02).to(dt)

    NSET = 4
    sets = []
    for _ in range(NSET):
        sets.append({n: rnd(o, i) for n, (o, i) in (
            ("self_attn.q_proj", (H, H)), ("self_attn.k_proj", (H, H)), ("self_attn.v_proj", (H, H)),
            ("self_attn.o_proj", (H, H)), ("mlp.gate_proj", (INTER, H)), ("mlp.up_proj", (INTER, H)),
            ("mlp.down_proj", (H, INTER)))})
    ones = torch.ones(H, dtype=dt)
    wd = {"model.embed_tokens.weight": rnd(1024, H), "lm_head.weight": rnd(VOCAB, H), "model.norm.weight": ones}
    for l in range(NL):
        p = "model.layers.%d." % l
        for n, w in sets[l % NSET].items():
            wd[p + n + ".weight"] = w
        wd[p + "input_layernorm.weight"] = ones
        wd[p + "post_attention_layernorm.weight"] = ones
    dims = O.LlamaDims(H, NH, NL, INTER, VOCAB)
    S = 343
    with torch.no_grad():
        t0 = time.perf_counter()
        _, _, kv = O.llama_forward(wd, dims, rnd(1, S, H), torch.arange(S).unsqueeze(0), None, all_logits=False)
        t_prefill = time.perf_counter() - t0
        t0 = time.perf_counter()
        ntok = 2
        for i in range(ntok):
            _, _, kv = O.llama_forward(wd, dims, rnd(1, 1, H), torch.tensor([[S + i]]), kv, all_logits=False)
        t_tok = (time.perf_counter() - t0) / ntok
    del wd, sets, kv
    mean_S = sum(prompt_len(i) for i in range(STORY_LEN)) / float(STORY_LEN)
    step_s = t_prefill * mean_S / S + T_GEN * t_tok
    notes.append("LLaMA-7B 32 layers bf16: S=343 prefill %.2fs, decode %.3fs/token (x%d tokens, prefill scaled to mean S=%.0f)"
                 % (t_prefill, t_tok, T_GEN, mean_S))
    if with_sdxl:
        import sdxl_oracle as SO
        c = SO.SDXL_BASE_UNET
        wdu = {k: torch.empty(*shp).normal_(0.0, 0.02) if len(shp) > 1 else torch.ones(*shp)
               for k, shp in SO.unet_shapes(c).items()}
        x = torch.randn(1, 4, 128, 128)
        ctx = torch.randn(1, 64, 2048)
        pooled = torch.randn(1, 1280)
        tid = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]], dtype=torch.float32)
        t0 = time.perf_counter()
        with torch.no_grad():
            SO.unet_forward(wdu, c, x, torch.tensor(500.0), ctx, pooled, tid)
        t_unet = time.perf_counter() - t0
        del wdu
        notes.append("SDXL-base UNet fp32, 128x128 latents, batch 1: %.1fs per forward (x2 CFG x%d steps)" % (t_unet, diffusion_steps))
        flop_rate = 6.747e12 / t_unet
        if time.perf_counter() - t_start + 10.5e12 / flop_rate < budget_s:
            vc = SO.SDXL_BASE_VAE
            wdv = {k: torch.empty(*shp).normal_(0.0, 0.02) if len(shp) > 1 else torch.ones(*shp)
                   for k, shp in SO.vae_decoder_shapes(vc).items()}
            t0 = time.perf_counter()
            with torch.no_grad():
                SO.vae_decode(wdv, vc, torch.randn(1, 4, 128, 128) * 0.1)
            t_vae = time.perf_counter() - t0
            del wdv
            notes.append("SDXL VAE decode fp32 -> 1024x1024: %.1fs" % t_vae)
        else:
            t_vae = 10.5e12 / flop_rate
            notes.append("VAE decode NOT run (budget): priced at the UNet's measured %.2f TFLOP/s = %.1fs" % (flop_rate / 1e12, t_vae))
        step_s += diffusion_steps * 2 * t_unet + t_vae
    # ViT-G: 48 layers over 2 weight sets, one image; amortised over the story
    if time.perf_counter() - t_start < budget_s - 20:
        import synth
        Wd, MLP = 1664, 8192
        vsets = [{k: (v if v.dim() == 1 else torch.randn(v.shape, generator=g) * 0.02)
                  for k, v in synth.vit_block_weights(900 + i, 8, 16).items()} for i in range(2)]
        for vs in vsets:      # full-width tensors (the synth helper only supplied the key names)
            vs.update({"ln_1.weight": torch.ones(Wd), "ln_1.bias": torch.zeros(Wd), "ln_2.weight": torch.ones(Wd),
                       "ln_2.bias": torch.zeros(Wd), "attn.in_proj.weight": torch.randn(3 * Wd, Wd, generator=g) * 0.02,
                       "attn.in_proj.bias": torch.zeros(3 * Wd), "attn.out_proj.weight": torch.randn(Wd, Wd, generator=g) * 0.02,
                       "attn.out_proj.bias": torch.zeros(Wd), "mlp.c_fc.weight": torch.randn(MLP, Wd, generator=g) * 0.02,
                       "mlp.c_fc.bias": torch.zeros(MLP), "mlp.c_proj.weight": torch.randn(Wd, MLP, generator=g) * 0.02,
                       "mlp.c_proj.bias": torch.zeros(Wd)})
        xv = torch.randn(1, 1024, Wd, generator=g)
        t0 = time.perf_counter()
        with torch.no_grad():
            for l in range(48):
                xv = O.vit_block_forward(vsets[l % 2], "", xv, 16)
        t_vit = time.perf_counter() - t0
        step_s += t_vit / STORY_LEN
        notes.append("ViT-G trunk fp32 (48 blocks, 1 image): %.1fs, amortised over %d steps" % (t_vit, STORY_LEN))
    else:
        notes.append("ViT-G not run (budget); < 1 percent of the step")
    return {"value": round(1.0 / step_s, 6), "unit": "story-steps/s", "cores": threads, "kind": "port",
            "seconds_per_story_step": round(step_s, 1),
            "arithmetic": {"llama": "bf16 (torch CPU)", "unet": "fp32 (bf16 has no fast CPU path), batch 1 x 2 for the CFG pair", "vae": "fp32",
                           "vit": "fp32", "note": "MIXED: the MLLM leg is priced in bf16, the de-tokenizer legs in fp32 — a baseline for "
                                                  "orientation, not a like-for-like dtype comparison with the bf16 GPU line"},
            "sample": "full-dimension single measurements composed by the step formula: " + "; ".join(notes) +
                      " (resamplers excluded, < 0.1 %% of the step; wall time of this sample %.0fs)" % (time.perf_counter() - t_start)}


def run_config0(seq_len=256):
    """BASELINE configs[0] as named: "gen_george.py MLLM text-only next-token on CPU, random-init LLaMA-7B config, seq_len=256, 1 image
    placeholder (plumbing, no GPU)".  The reference's own CPU-runnable case, so it runs the ORACLE (the CPU restatement of the
    reference, `kind` = "port") — the product path has no CPU fallback by construction.  Full LLaMA-2-7B dimensions (hidden 4096, 32
    layers, inter 11008, vocab 32066; the 32 layers cycle through 4 distinct random weight sets so nothing is cache-resident), bf16:
    a 256-token prompt whose 64 <img_i> placeholder rows are replaced by the full-size input resampler's output for one random
    256 x 4096 ViT feature (gen_george.py:176-188 / models.py:135-150), one prefill, the image-token logits processor, the greedy token."""
    import seedstory_oracle as O
    import synth
    threads = torch.get_num_threads()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(0)

    def rnd(*s):
        return (torch.randn(*s, generator=g) * 0.02).to(dt)

    t_build = time.perf_counter()
    sets = [{n: rnd(o, i) for n, (o, i) in (("self_attn.q_proj", (H, H)), ("self_attn.k_proj", (H, H)), ("self_attn.v_proj", (H, H)),
                                             ("self_attn.o_proj", (H, H)), ("mlp.gate_proj", (INTER, H)), ("mlp.up_proj", (INTER, H)),
                                             ("mlp.down_proj", (H, INTER)))} for _ in range(4)]
    ones = torch.ones(H, dtype=dt)
    wd = {"model.embed_tokens.weight": rnd(VOCAB, H), "lm_head.weight": rnd(VOCAB, H), "model.norm.weight": ones}
    for l in range(NL):
        pfx = "model.layers.%d." % l
        for n, w in sets[l % 4].items():
            wd[pfx + n + ".weight"] = w
        wd[pfx + "input_layernorm.weight"] = ones
        wd[pfx + "post_attention_layernorm.weight"] = ones
    rw = synth.resampler_weights(21, "", 8, H, dtype=dt)              # input resampler: 64 queries over the 256-token ViT feature
    t_build = time.perf_counter() - t_build
    dims = O.LlamaDims(H, NH, NL, INTER, VOCAB)
    n_txt = seq_len - 66
    ids = torch.cat([torch.tensor([BOS]), torch.randint(3, 32000, (n_txt - 1,), generator=g), torch.tensor(IMG_IDS)])   # text, <img> 64 placeholders </img>
    assert ids.numel() == seq_len
    with torch.no_grad():
        t0 = time.perf_counter()
        feat = O.resampler_forward(rw, "", rnd(1, 256, H) * 50.0, 32)                                      # [1, 64, 4096]
        emb = wd["model.embed_tokens.weight"][ids].clone()
        emb[n_txt + 1:n_txt + 65] = feat[0]
        t_embed = time.perf_counter() - t0
        t0 = time.perf_counter()
        logits, _, kv = O.llama_forward(wd, dims, emb.unsqueeze(0), torch.arange(seq_len).unsqueeze(0), None, all_logits=False)
        t_prefill = time.perf_counter() - t0
        t0 = time.perf_counter()
        sc = O.image_token_logits_processor(int(ids[-1]), logits.reshape(-1, VOCAB)[-1].float(), IMG_IDS)
        tok = int(sc.argmax())
        logits2, _, kv = O.llama_forward(wd, dims, wd["model.embed_tokens.weight"][tok].view(1, 1, H), torch.tensor([[seq_len]]), kv, all_logits=False)
        t_next = time.perf_counter() - t0
    total = t_embed + t_prefill + t_next
    return {"metric": "seconds to the next token, CPU plumbing case (BASELINE configs[0])", "value": round(total, 3), "unit": "s", "n_gpus": 0,
            "steps": 1, "warmup": 0, "ms_per_step": round(total * 1e3, 1), "higher_is_better": False, "scaling": "none", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (random-init LLaMA-2-7B-shaped weights, 4 distinct layer weight sets cycled over 32 layers)",
            "config": {"workload": "BASELINE configs[0]: text-only prompt of %d tokens with 1 image placeholder (64 rows from the full-size input "
                                   "resampler), prefill + logits processor + one KV-cached next-token forward, on the host CPU" % seq_len},
            "cpu_baseline": {"value": round(total, 3), "unit": "s", "cores": threads, "kind": "port",
                             "sample": "the whole case: input resampler + splice %.2fs, S=%d prefill of 32 layers + lm_head %.2fs, processor + next-token "
                                       "forward %.2fs (weight build %.1fs not counted)" % (t_embed, seq_len, t_prefill, t_next, t_build)},
            "first_token": tok, "finite": bool(torch.isfinite(logits2.float()).all())}


def flush_c_stdio():
    """RCCL prints its version banner through C stdio when the first communicator comes up; flushed only at exit it would
    land BEHIND the JSON line in a redirected stdout.  Flushing C stdio right after the first collective keeps the JSON
    line the last line of the output."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


MAX_SLOTS = 8       # sequence slots of one engine = stories sharing one sweep of the weights per decode token


def slot_groups(spg, max_slots=None):
    """Stories per GPU -> sizes of the lock-step decode groups (an engine sweeps the weights once per token for <= 8 slots:
    1 - 2 through the dot-product GEMV, 3 - 8 through the MFMA form)."""
    m = max_slots or MAX_SLOTS
    n = (spg + m - 1) // m
    return [spg // n + (1 if g < spg % n else 0) for g in range(n)]


def build_engines(device, dtype, spg):
    """One engine per group of <= MAX_SLOTS story slots, all over the same synthetic weight tensors."""
    engs, shared = [], None
    for n in slot_groups(spg):
        e, shared = build_engine(device, dtype, n, shared)
        engs.append(e)
    return engs, shared


def build_engine(device, dtype, n_seq, shared=None):
    """LLaMA engine over synthetic weights; `shared` = another engine's weight tensors (batch-1 engine of the same model)."""
    from seedstory.llama import LlamaEngine
    if shared is None:
        torch.manual_seed(1234)

        def rnd(*s):
            return torch.randn(*s, device=device, dtype=dtype) * 0.02

        ones = lambda n: torch.ones(n, device=device, dtype=dtype)  # noqa: E731
        shared = dict(layers=[(rnd(3 * H, H), rnd(H, H), rnd(2 * INTER, H), rnd(H, INTER), ones(H), ones(H)) for _ in range(NL)],
                      embed=rnd(VOCAB, H), lm_head=rnd(VOCAB, H), final_norm=ones(H))
    eng = LlamaEngine.from_prebuilt(hidden=H, n_heads=NH, n_layers=NL, inter=INTER, vocab=VOCAB, dtype=dtype, device=device,
                                    cache_cap=CACHE_CAP, max_new=128, max_prefill_rows=1024 * n_seq, img_ids=IMG_IDS, eos_id=EOS,
                                    n_seq=n_seq, **shared)
    return eng, shared


class Runner:
    """Drives rounds of `spg` lock-step stories on one engine: sequentially, or with the MLLM half of round r+1 on a
    second HIP stream (second host thread) under the render of round r — the render does not feed the next MLLM half
    (the context takes the regressed FEATURE, gen_george.py:224, not the decoded image).  Work per round is unchanged."""

    def __init__(self, eng, rin, rout, vit, adapter, spg, device, args, seed0):
        self.eng, self.rin, self.rout, self.vit, self.adapter = eng, rin, rout, vit, adapter
        self.spg, self.device, self.args = spg, device, args
        self.story_no = seed0
        self.sts = None
        self.overlap = adapter is not None and not args.no_overlap
        self.overlap_fallback = None       # repr of the exception that made warm() drop the two-stream schedule
        self.side = torch.cuda.Stream(device=device) if self.overlap else None

    def next_stories(self):
        if self.sts is None or self.sts[0].step >= STORY_LEN:
            self.sts = []
            for _ in range(self.spg):
                self.story_no += 1
                self.sts.append(Story(self.story_no, self.device))
        return self.sts

    def one_step(self):
        return run_round(self.next_stories(), self.eng, self.rin, self.rout, self.vit, self.args.kv_reuse, self.adapter,
                         self.args.diffusion_steps)

    def _mllm_async(self, box):
        import threading

        def work():
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self.side):
                    box["stories"] = self.next_stories()
                    box["feat"] = mllm_part(box["stories"], self.eng, self.rin, self.rout, self.vit, self.args.kv_reuse)
                self.side.synchronize()
            except BaseException as ex:   # surfaced by the driver thread (run)
                box["err"] = ex
        th = threading.Thread(target=work)
        th.start()
        return th

    def run(self, n):
        if n <= 0:
            return
        if not self.overlap:
            for _ in range(n):
                self.one_step()
            return
        torch.cuda.synchronize()
        box = {}
        self._mllm_async(box).join()                   # round 0's MLLM half has nothing to hide under
        for r in range(n):
            if "err" in box:
                raise box["err"]
            cur_stories, cur_feat = box["stories"], box["feat"]
            box = {}
            th = self._mllm_async(box) if r + 1 < n else None
            sdxl_part(cur_stories, self.adapter, cur_feat, self.args.diffusion_steps)
            if th is not None:
                th.join()
        torch.cuda.synchronize()

    def warm(self, n):
        try:
            self.run(n if (n > 0 or not self.overlap) else 1)   # >= 1 untimed round checks the schedule
        except Exception as ex:      # the two-stream schedule is an optimisation: never let it take the measurement down
            print("bench: overlapped schedule failed (%r); falling back to the sequential one" % (ex,), file=sys.stderr)
            self.overlap = False
            self.overlap_fallback = repr(ex)[:300]
            torch.cuda.synchronize()
            self.sts = None
            self.run(n)
        self.sts = None              # the timed region starts at a story boundary


def measure_roofline(eng, adapter, SPG, RG, device, dtype, args):
    """The `roofline` section of the JSON line (rank 0, after the timed region; shared by the replica and the slot-ring
    partitions): the HBM-bound half (decode GEMV, stacked image-token block, stacked prompt prefill) measured with HIP
    events on the first decode group, and — with a de-tokenizer — the MFMA-bound half (one UNet forward of a render
    group's batch, the dominant ff1 GEGLU GEMM over rotating weights, the fp8 variant)."""
    GRP = eng.n_seq
    dbg = (lambda m: (torch.cuda.synchronize(), print("ROOF", m, flush=True))) if os.environ.get("SS_ROOF_DEBUG") else (lambda m: None)
    dbg("start")
    # ---- roofline of the dominant kernel (decode GEMV, HBM-bound), measured live with HIP events ----
    roof = None
    if True:
        for b in range(GRP):
            eng.select(b).set_lengths(343, 343)
        prof = eng.profile_decode(8)
        dbg("profile_decode done")
        # dominant kernel = the decode weight-streaming GEMV.  With <= 2 slots per sweep the K=hidden projections
        # (qkv, o, gate|up per layer + lm_head: 97 launches/token) are ss::gemv_kernel<bf16,8,2,NB> and the down
        # projection a different symbol.
        # With 3-8 slots every projection runs ss::gemv_mfma_exact_kernel (v_mfma_f32_16x16x32, 8 waves = 8 K slices of a
        # 16-row tile; the 11008-deep down projection in the packed 43-step form): 129 launches/token, averaged together.
        if GRP <= 2:
            kern = "gemv_kernel<bf16_t,8,2,%d>" % GRP
            n_launch, tot_bytes, tot_ms = prof["gemv_launches"], prof["gemv_bytes"], prof["gemv_ms"]
        else:
            kern = "gemv_mfma_exact_kernel<bf16_t>"
            n_launch = prof["gemv_launches"] + prof["gemv_down_launches"]
            tot_bytes = prof["gemv_bytes"] + prof["gemv_down_bytes"]
            tot_ms = prof["gemv_ms"] + prof["gemv_down_ms"]
        per_launch_bytes = tot_bytes / n_launch
        per_launch_ms = tot_ms / n_launch
        achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
        down = prof["gemv_down_bytes"] / (prof["gemv_down_ms"] * 1e-3) / 1e9
        # HBM bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command,
        # corrected as MI355X_MICROARCH.md prescribes (tools/pmc_traffic.py -> profiles/round2_pmc_summary.json)
        traffic, under_render_us = None, None
        pmcj = {}
        for name in ("round6_pmc_summary.json", "round4_pmc_summary.json", "round3_pmc_summary.json", "round2_pmc_summary.json", "round1_pmc_summary.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc):
                try:
                    pmcj = json.load(open(pmc))
                    break
                except Exception:
                    pmcj = {}
        if pmcj.get("stories_per_gpu") == GRP:      # counters were collected at this many slots per sweep
            # (the MFMA form is several template instances — plain | SiLU pair, 16 | 43 steps: dispatch-weighted mean)
            base = kern.split("<")[0]
            recs = [v for k, v in pmcj.get("gemv_hbm_traffic", {}).items() if k.split("<")[0] == base and v.get("dispatches")]
            if recs:
                traffic = round(sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in recs) / sum(v["dispatches"] for v in recs))
            calls = pmcj.get("gemv_calls_under_render", {})          # launch-weighted over the template instances when recorded
            us = [(v, calls.get(k, 1)) for k, v in pmcj.get("gemv_avg_us_under_render", {}).items() if k.split("<")[0] == base]
            if us:
                under_render_us = round(sum(v * c for v, c in us) / sum(c for _, c in us), 3)
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                "kernel": "ss::" + kern, "slots_per_sweep": GRP,
                "launches_per_token": n_launch, "bytes_per_launch": round(per_launch_bytes),
                "avg_launch_us": round(per_launch_ms * 1e3, 3),
                "avg_launch_us_note": "HIP events around every launch of un-captured decode tokens with nothing else on the "
                                      "GPU; under the shipped overlapped schedule (render on the other stream) the "
                                      "rocprofv3 kernel-trace average is avg_launch_us_under_render",
                "avg_launch_us_under_render": under_render_us,
                "also": {"down projection GB/s": round(down, 1),
                         "all_gemv GB/s per token": round((prof["gemv_bytes"] + prof["gemv_down_bytes"]) /
                                                          ((prof["gemv_ms"] + prof["gemv_down_ms"]) * 1e-3) / 1e9, 1),
                         "token_ms_eager": round(prof["token_ms"], 4), "attn_ms": round(prof["attn_ms"], 4),
                         "misc_ms": round(prof["misc_ms"], 4),
                         "weight_bytes_per_generated_token_per_story": round(13.215e9 / GRP)}}
    if roof is not None:
        # the two BATCHED parts of the MLLM half (seedstory/llama.py::prefill_batch), HIP events on the stream:
        #  * image-token block continuation: after <img> the logits processor forces 65 tokens (generation.py:19-31); the
        #    66 rows [<img> .. </img>] of every slot of the group run as ONE stacked forward of GRP x 66 rows — HBM-bound,
        #    the 13.2 GB of layer weights are streamed once (per group, per story step) instead of 66 x GRP times;
        #  * prompt prefill: the group's prompts (S = 115 .. 913 each) as one stacked forward — MFMA-bound.
        def timed_prefill(rows_per_slot, kv0, reps=3):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            z = [torch.zeros(rows_per_slot, H, device=device, dtype=dtype)] * GRP
            tot = 0.0
            for i in range(reps + 1):
                for b in range(GRP):
                    eng.select(b).set_lengths(kv0, kv0)
                ev0.record()
                if GRP == 1:
                    eng.select(0).prefill(z[0])
                else:
                    eng.prefill_batch(z)
                ev1.record()
                torch.cuda.synchronize()
                if i:
                    tot += ev0.elapsed_time(ev1)
            return tot / reps
        layer_w_bytes = NL * (3 * H * H + H * H + 2 * INTER * H + H * INTER) * 2
        blk_ms = timed_prefill(66, 343 + CAPTION)
        dbg("block done")
        blk_flops = GRP * 66 * (2.0 * layer_w_bytes / 2 + 4.0 * NL * H * (343 + CAPTION + 33))   # projections + attention
        hbm_floor, mfma_floor = layer_w_bytes / 8e12, blk_flops / 2.5e15
        roof["block_continuation"] = {
            "what": "the 65 processor-forced image tokens of a story step: %d slots x 66 rows as one stacked forward" % GRP,
            "bound": "hbm" if hbm_floor >= mfma_floor else "mfma", "rows": GRP * 66, "ms": round(blk_ms, 3),
            "weight_bytes": layer_w_bytes, "flops": blk_flops,
            "hbm": {"achieved": round(layer_w_bytes / (blk_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(layer_w_bytes / (blk_ms * 1e-3) / 8e12, 4)},
            "mfma": {"achieved": round(blk_flops / (blk_ms * 1e-3) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": round(blk_flops / (blk_ms * 1e-3) / 2.5e15, 4)},
            "achieved": round(layer_w_bytes / (blk_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(layer_w_bytes / (blk_ms * 1e-3) / 8e12, 4),
            "note": "algorithmic bytes = the 32 layers' weights once (lm_head on the last rows excluded); per story step "
                    "this replaces 65 of the 115 decode iterations.  The ridge of the chip (2.5 PFLOP/s / 8 TB/s = 312 flop/B) "
                    "is at 312 stacked rows: 4 slots (264 rows) are HBM-bound, 8 slots (528 rows) MFMA-bound — `bound` names "
                    "the larger floor, both fractions are given; achieved/frac at the top level stay the HBM ones"}
        S_big = prompt_len(STORY_LEN - 1)
        pf_ms = timed_prefill(S_big, 0, reps=2)
        dbg("prefill done")
        pf_flops = GRP * S_big * (2.0 * layer_w_bytes / 2 + 4.0 * NL * H * (S_big + 1) / 2)   # projections + causal attention
        roof["prompt_prefill"] = {
            "what": "stacked prompt prefill of the %d lock-step stories, S = %d each (the window after step %d)" % (GRP, S_big, WINDOW - 1),
            "bound": "mfma", "rows": GRP * S_big, "ms": round(pf_ms, 3), "flops": pf_flops,
            "achieved": round(pf_flops / (pf_ms * 1e-3) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
            "frac": round(pf_flops / (pf_ms * 1e-3) / 2.5e15, 4)}
        roof["tokens_per_story_step"] = {"generated": T_GEN, "iterated_in_the_device_loop": T_GEN - 65 if eng.img_block_enabled() else T_GEN,
                                         "fed_as_one_stacked_block": 65 if eng.img_block_enabled() else 0}
        for b in range(GRP):
            eng.select(b).reset()
        eng.select(0)
    if adapter is not None:
        # MFMA-bound half: one SDXL-base UNet forward (batch 2S = CFG pairs of the S resident stories, 128x128
        # latents), HIP events on the stream
        UB = 2 * SPG // RG                                     # one render group's CFG batch
        x = torch.randn(UB, 4, 128, 128, device=device, dtype=dtype)
        ctx = torch.randn(UB, 64, 2048, device=device, dtype=dtype)
        cond = {"text_embeds": torch.randn(UB, 1280, device=device, dtype=dtype),
                "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * UB, dtype=torch.float32)}
        fp8_was = getattr(adapter.unet, "_fp8", False)
        adapter.unet.enable_fp8(False)                         # this block prices the bf16 forward whatever --unet-fp8 says
        adapter.unet(x, 500.0, ctx, added_cond_kwargs=cond)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            adapter.unet(x, 500.0, ctx, added_cond_kwargs=cond)
        e1.record()
        torch.cuda.synchronize()
        ms_eager = e0.elapsed_time(e1) / 3
        # the render REPLAYS the forward from a hipGraph (steps 1..n-1 of every image): the same three forwards as graph replays.  An eager
        # forward is ~1100 launches and reads the HOST's launch rate on a busy box (round 6: one closing run read 115.4 ms eagerly while its
        # rounds implied <= 104 ms per forward); `forward_ms` is the replay, the eager figure is kept beside it.
        ms, ms_graph = ms_eager, None
        try:
            from seedstory.diffusion import timestep_embedding
            st = timestep_embedding(torch.tensor([500.0]), adapter.unet.cfg["block_out_channels"][0]).to(device=device, dtype=dtype).expand(UB, -1).contiguous()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                adapter.unet(x, None, ctx, added_cond_kwargs=cond, return_dict=False, temb_in=st)
            gr.replay()
            e0.record()
            for _ in range(3):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            ms_graph = e0.elapsed_time(e1) / 3
            ms = min(ms_graph, ms_eager)
            del gr
        except Exception:
            torch.cuda.synchronize()
        flops = UB * 6.747e12                                  # 3.3735 TMAC per sample per forward
        # the single dominant MFMA kernel of the forward: the GEGLU ff1 projection of the 1280-wide transformer blocks
        # (60 launches per forward, ~17 % of its time): [UB*1024, 1280] x [10240, 1280]^T with the value*gelu(gate)
        # epilogue.  Timed over rotating weight copies (each UNet weight is touched once per forward).
        from seedstory import ops as _ops
        Mg, Ng, Kg = UB * 1024, 10240, 1280
        ag = torch.randn(Mg, Kg, device=device, dtype=dtype)
        wg = [torch.randn(Ng, Kg, device=device, dtype=dtype) * 0.02 for _ in range(4)]
        bg = torch.zeros(Ng, device=device, dtype=dtype)
        for i in range(2):
            _ops.gemm_geglu(ag, wg[i], bg)
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
        for i in range(12):
            _ops.gemm_geglu(ag, wg[i % 4], bg)
        g1.record()
        torch.cuda.synchronize()
        gemm_us = g0.elapsed_time(g1) / 12 * 1e3
        gemm_tf = 2.0 * Mg * Ng * Kg / (gemm_us * 1e-6) / 1e12
        # the same forward / the same GEMM with fp8 (e4m3) operands (SURVEY §8 ★ row; priced against the 5 PFLOP/s
        # dense fp8 peak).  Not part of `value` unless --unet-fp8 was given.
        fp8_leg = None
        try:
            was = fp8_was
            adapter.unet.enable_fp8(True)
            adapter.unet(x, 500.0, ctx, added_cond_kwargs=cond)
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record()
            for _ in range(3):
                adapter.unet(x, 500.0, ctx, added_cond_kwargs=cond)
            f1.record()
            torch.cuda.synchronize()
            ms8 = f0.elapsed_time(f1) / 3
            adapter.unet.enable_fp8(was)
            a8, sa8 = _ops.quantize_rows_fp8(ag)
            w8s = [_ops.quantize_rows_fp8(wi) for wi in wg]
            for i in range(2):
                _ops.gemm_fp8(a8, sa8, w8s[i][0], w8s[i][1], bias=bg, geglu=True)
            f0.record()
            for i in range(12):
                _ops.gemm_fp8(a8, sa8, w8s[i % 4][0], w8s[i % 4][1], bias=bg, geglu=True)
            f1.record()
            torch.cuda.synchronize()
            us8 = f0.elapsed_time(f1) / 12 * 1e3
            tf8 = 2.0 * Mg * Ng * Kg / (us8 * 1e-6) / 1e12
            fp8_leg = {"forward_ms": round(ms8, 3), "speedup_vs_bf16_forward": round(ms_eager / ms8, 3),    # eager vs eager
                       "linear_layers": "proj_in, q|k|v, to_out, to_q, ff1 (GEGLU), ff2, proj_out of every transformer block; "
                                        "convs / attention / context K,V stay bf16",
                       "dominant_kernel": {"kernel": "ss::gemm_sp_kernel<fp8_t,...> (v_mfma_scale_f32_16x16x128_f8f6f4) + GEGLU epilogue",
                                           "shape_MNK": [Mg, Ng, Kg], "avg_launch_us": round(us8, 1), "achieved": round(tf8, 1),
                                           "unit": "TFLOP/s", "peak": 5000.0, "frac": round(tf8 / 5000.0, 4)},
                       "in_value": bool(args.unet_fp8)}
            del a8, sa8, w8s
        except Exception as exc:           # the bf16 numbers above stand on their own
            fp8_leg = {"error": str(exc)[:200]}
        del ag, wg
        # control for the "what does a long-K GEMM sustain on this box" figure: an 8192^3 bf16 GEMM on the 256x256 tile (cfg 60),
        # N(0,1) activations x N(0,0.02) weights, two rotating weight copies, measured in THIS run (the round-4
        # line carried the constant 1300.0 here)
        ctl = None
        try:
            from seedstory import _lib

            def sq(scale=1.0, uniform=False):
                t = torch.rand(8192, 8192, device=device) * 2 - 1 if uniform else torch.randn(8192, 8192, device=device) * scale
                return t.to(dtype)
            # round 6: the shipped long-K tile is the ping-pong 8-phase 256x256 kernel (cfg 54); the round-5 one-barrier tile
            # (cfg 60) is measured beside it on the same operands, INTERLEAVED (3 rounds over the 4 combinations, median): measured one
            # after the other the chip's power state decides the ranking, not the kernel (first line of round 6: 1252 vs 1352 on uniform
            # operands sequentially, 1547 vs 1381 interleaved in tools/gemm_ubench)
            ops_n = (sq(), [sq(0.02), sq(0.02)])
            ops_u = (sq(uniform=True), [sq(uniform=True), sq(uniform=True)])
            samples = {(c, k): [] for c in (54, 60) for k in ("n", "u")}
            for rnd in range(3):
                for cfgc in (54, 60):
                    _lib.set_tuning("gemm_cfg", cfgc)
                    for k, (ac, wc) in (("n", ops_n), ("u", ops_u)):
                        _ops.gemm(ac, wc[0])
                        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        c0.record()
                        for i in range(8):
                            _ops.gemm(ac, wc[i % 2])
                        c1.record()
                        torch.cuda.synchronize()
                        samples[(cfgc, k)].append(c0.elapsed_time(c1) / 8 * 1e3)

            def med(c, k):
                us = sorted(samples[(c, k)])[1]
                return round(us, 1), round(2.0 * 8192 ** 3 / (us * 1e-6) / 1e12, 1)
            us_n, tf_n = med(54, "n")
            us_u, tf_u = med(54, "u")
            us_n5, tf_n5 = med(60, "n")
            us_u5, tf_u5 = med(60, "u")
            del ops_n, ops_u
            ctl = {"shape_MNK": [8192, 8192, 8192], "tile": "gemm_pp_kernel<bf16,256,8> cfg 54 (ping-pong 8-phase, ss_gemm_pp.inc)", "unit": "TFLOP/s",
                   "method": "3 interleaved rounds over {cfg 54, cfg 60} x {operand class}, 8 launches over 2 rotating weight copies each, median round",
                   "randn_x_0.02randn": {"avg_launch_us": us_n, "achieved": tf_n, "frac": round(tf_n / 2500.0, 4)},
                   "uniform_pm1_both": {"avg_launch_us": us_u, "achieved": tf_u, "frac": round(tf_u / 2500.0, 4)},
                   "round5_tile_cfg60": {"tile": "gemm_sp_kernel<bf16,256,256,...> cfg 60 (one barrier per K tile)",
                                         "randn_x_0.02randn": {"avg_launch_us": us_n5, "achieved": tf_n5},
                                         "uniform_pm1_both": {"avg_launch_us": us_u5, "achieved": tf_u5}},
                   "note": "measured in this run; the guide's 256x256 8-phase + st_16x32 template reads ~1470 on uniform [-1,1) "
                           "operands at this shape (cdna_hip_programming.md:614) — this is the library's own kernel, not a ceiling"}
        except Exception as exc:
            ctl = {"error": str(exc)[:200]}
        finally:
            from seedstory import _lib
            _lib.set_tuning("gemm_cfg", 0)
        roof_mllm = roof
        from seedstory import tune as _tune
        ff1_cfg = _tune.lookup(Mg, Ng, Kg, 1)
        # the counters describe ONE tile configuration: a record taken with another (cfg, XCD group) than the shipped table holds for
        # this shape (or carrying neither: rounds 1-3) is refused rather than quoted (the round-3 line carried the
        # ff1 over-fetch of a retired XCD group).  Checked per row, so that ADDING shapes to the table leaves the record valid.
        import hashlib
        table_sha = hashlib.sha256(open(os.path.join(ROOT, "seed-story_amd", "seedstory", "tune_gfx950.json"), "rb").read()).hexdigest()[:16]
        ghbm = pmcj.get("gemm_hbm_traffic", {})
        ff1_rec = ghbm.get("ff1_%dx%dx%d_geglu" % (Mg, Ng, Kg)) or ghbm.get("ff1_%dx%dx%d" % (Mg, Ng, Kg)) or {}
        shipped_cfg = "%d/%d" % tuple(ff1_cfg) if ff1_cfg else None
        traffic_ok = bool(ff1_rec) and ff1_rec.get("cfg_swz") is not None and ff1_rec.get("cfg_swz") == shipped_cfg
        ff1_traffic = ff1_rec.get("hbm_bytes_per_launch") if traffic_ok else None
        traffic_note = ("HBM-side bytes per launch of the ff1 GEMM from profiles/round6_pmc_summary.json (tools/pmc_round6.sh), counters collected on tile cfg/XCD group "
                        "%s = the shipped table's entry for this shape" % shipped_cfg
                        if ff1_traffic else "no PMC record of this shape on the shipped tile (table: %s; record: %s) — traffic withheld"
                        % (shipped_cfg, ff1_rec.get("cfg_swz")))
        roof = {"bound": "mfma", "achieved": round(flops / (ms * 1e-3) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                "frac": round(flops / (ms * 1e-3) / 2.5e15, 4), "traffic": ff1_traffic,
                "kernel": "SDXL UNet forward, all kernels (ss::gemm_pp_kernel / ss::gemm_sp_kernel<bf16,*> incl. implicit-GEMM conv3x3, ss::flash_attn3p_kernel<bf16,64>, norms); traffic = HBM bytes per launch of the dominant ff1 GEMM",
                "traffic_note": traffic_note, "tile_table_sha16": table_sha,
                "pmc_record_tile_table_sha16": pmcj.get("tile_table_sha16"),    # (the table the counters ran on; rows are matched per shape + tile)
                "flops_per_forward": flops, "forward_ms": round(ms, 3), "forward_ms_eager": round(ms_eager, 3),
                "forward_ms_graph_replay": None if ms_graph is None else round(ms_graph, 3), "unet_batch": UB,
                "gemm_8192cubed_control": ctl,      # this library's own long-K GEMM on random operands, measured in this run (not a ceiling claim)
                "dominant_kernel": {"kernel": "%s (tile table cfg %s) + GEGLU epilogue" % ("ss::gemm_pp_kernel" if ff1_cfg and 50 <= ff1_cfg[0] < 60 else "ss::gemm_sp_kernel", ff1_cfg),
                                    "shape_MNK": [Mg, Ng, Kg], "avg_launch_us": round(gemm_us, 1),
                                    "achieved": round(gemm_tf, 1), "unit": "TFLOP/s", "frac": round(gemm_tf / 2500.0, 4),
                                    "algorithmic_bytes": 2 * (Mg * Kg + Ng * Kg + Mg * Ng // 2), "traffic": ff1_traffic},
                "note": "a round is %d UNet forwards of batch %d (MFMA-bound) + per story %d generated tokens = %d iterated decode "
                        "tokens (weight-streaming GEMV, HBM-bound, %d slots per sweep: mllm_decode_gemv) + 65 processor-forced "
                        "image tokens fed as one stacked block (mllm_decode_gemv.block_continuation)"
                        % (args.diffusion_steps, UB, T_GEN, T_GEN - 65 if eng.img_block_enabled() else T_GEN, GRP),
                "unet_fp8": fp8_leg,
                "mllm_decode_gemv": roof_mllm}
    return roof


def measure_tolerance_modes(runner, engs, shared, rin, rout, vit, adapter, SPG, device, dtype, args, value_bf16):
    """What the modes that meet the reference's arithmetic cost, next to the benched bf16 ones (rank 0, N = 1, after the
    timed region):
      * `vae_fp32` — diffusers up-casts the VAE to fp32 (`force_upcast`; the reference loads it at gen_george.py:62), the
        benched decode runs in bf16: the same schedule re-timed with the exact-fp32 decoder, and the uint8 deviation of the
        bf16 decode from the fp32 one on the same latents;
      * `mllm_fp32` — the part the north-star 1e-3 gate is about (`img_gen_feat`: MLLM half + regressor) in exact fp32
        arithmetic (fp32 weights, exact-fp32 MFMA chains): its story-steps/s next to the bf16 MLLM half alone."""
    from seedstory import _lib, ops
    out = {}
    if adapter is not None:
        a0 = adapter[0] if isinstance(adapter, (list, tuple)) else adapter
        vae = a0.sdxl_pipe.vae
        lat = (torch.randn(1, 4, 128, 128, device=device) * 0.13025 * 3.0).to(dtype)
        sc = 1.0 / vae.config.scaling_factor
        imgs, ms = {}, {}
        for mode in (0, 1):
            _lib.set_tuning("vae_fp32", mode)
            try:
                vae.decode_nhwc(lat, prescale=sc)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                img, Hh, Ww = vae.decode_nhwc(lat, prescale=sc)
                e1.record()
                torch.cuda.synchronize()
                ms[mode] = e0.elapsed_time(e1)
                imgs[mode] = ops.image_to_u8(img, Hh * Ww).view(Hh, Ww, 3).int()
            finally:
                _lib.set_tuning("vae_fp32", 0)
        dev_ = (imgs[0] - imgs[1]).abs().float()
        # the fp32-tensor decode with its convolutions / linears as split-bf16 MFMA products (gemm_f32_split): 4.5e-6 per product
        _lib.set_tuning("vae_fp32", 1)
        _lib.set_tuning("gemm_f32_split", 1)
        try:
            vae.decode_nhwc(lat, prescale=sc)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            img, Hh, Ww = vae.decode_nhwc(lat, prescale=sc)
            e1.record()
            torch.cuda.synchronize()
            ms_split = e0.elapsed_time(e1)
            dev_s = (ops.image_to_u8(img, Hh * Ww).view(Hh, Ww, 3).int() - imgs[1]).abs().float()
        finally:
            _lib.set_tuning("vae_fp32", 0)
            _lib.set_tuning("gemm_f32_split", 0)

        def rate(knobs, n=2):
            for k in knobs:
                _lib.set_tuning(k, 1)
            try:
                runner.sts = None
                runner.warm(1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                runner.run(n)
                torch.cuda.synchronize()
                return n * SPG / (time.perf_counter() - t0)
            finally:
                for k in knobs:
                    _lib.set_tuning(k, 0)
                runner.sts = None
        n = 2
        v32 = rate(("vae_fp32",))
        v32s = rate(("vae_fp32", "gemm_f32_split"))
        out["vae_fp32"] = {"value_vae_fp32": round(v32, 4), "value_vae_fp32_split": round(v32s, 4), "value_bf16_vae": value_bf16,
                           "unit": "story-steps/s", "rounds_timed": n, "decode_ms_bf16": round(ms[0], 2), "decode_ms_fp32": round(ms[1], 2),
                           "decode_ms_fp32_split": round(ms_split, 2),
                           "uint8_dev_bf16_vs_fp32_decode": {"mean": round(float(dev_.mean()), 3), "max": int(dev_.max()),
                                                             "weights": "synthetic (random) VAE, latents ~ 3 x N(0, 1) x scaling"},
                           "uint8_dev_fp32_split_vs_fp32_decode": {"mean": round(float(dev_s.mean()), 5), "max": int(dev_s.max())},
                           "note": "diffusers decodes an fp16 SDXL VAE in fp32 (force_upcast) and a bf16 one in bf16; `value` above uses the bf16 "
                                   "decoder; fp32_split = fp32 tensors with the convolutions as three bf16 MFMA products per fragment pair"}
    # the MLLM half alone, bf16 (the benched engines) and exact fp32 (a second set of modules with fp32 weights)
    def mllm_rate(engines, rin_, rout_, vit_, n=2):
        r = Runner(engines if len(engines) > 1 else engines[0], rin_, rout_, vit_, None, SPG, device, args, 555000)
        r.one_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r.one_step()
        torch.cuda.synchronize()
        return n * SPG / (time.perf_counter() - t0), (time.perf_counter() - t0) / n * 1e3
    class _CastAdapter:
        """The bf16 de-tokenizer behind an fp32 MLLM half: the regressed feature is cast at the hand-over."""

        def __init__(self, a, dt):
            self.a, self.dt = a, dt

        def generate(self, image_embeds=None, **kw):
            return self.a.generate(image_embeds=image_embeds.to(self.dt), **kw)

        def __getattr__(self, k):
            return getattr(self.a, k)

    def full_rate(engines, rin_, rout_, vit_, n=2):
        """The WHOLE pipeline (overlapped schedule, bf16 render) with the given MLLM half: story-steps/s over n rounds."""
        ads = [_CastAdapter(a, dtype) for a in adapter] if isinstance(adapter, (list, tuple)) else _CastAdapter(adapter, dtype)
        r = Runner(engines if len(engines) > 1 else engines[0], rin_, rout_, vit_, ads, SPG, device, args, 777000)
        r.warm(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.run(n)
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        return n * SPG / dt_, dt_ / n * 1e3, r.overlap_fallback

    try:
        v16, ms16 = mllm_rate(engs, rin, rout, vit)
        e32, _ = build_engines(device, torch.float32, SPG)
        rin32, rout32, vit32 = build_frontend(device, torch.float32)
        v32, ms32 = mllm_rate(e32, rin32, rout32, vit32)
        out["mllm_fp32"] = {"value_fp32": round(v32, 4), "value_bf16": round(v16, 4), "unit": "story-steps/s (MLLM half only: prefill, "
                            "decode, image-token block, ViT at story start, input / output resamplers)",
                            "ms_per_round_fp32": round(ms32, 1), "ms_per_round_bf16": round(ms16, 1), "stories": SPG,
                            "note": "exact fp32 MFMA chains (1/16 rate): fp32 9.4e-6 vs the real reference at hidden 4096; bf16 2.3e-2 "
                                    "where the reference's own bf16 run is 3.2e-2 from its fp32 run"}
        if adapter is not None:
            vfull, msfull, fb = full_rate(e32, rin32, rout32, vit32)
            out["mllm_fp32"].update({"value_full_pipeline": round(vfull, 4), "ms_per_round_full_pipeline": round(msfull, 1),
                                     "overlap_fallback": fb})
        # gate mode: the same fp32 modules with every fp32-tensor GEMM as three bf16 MFMA products on (hi, lo) operand halves
        # (`gemm_f32_split`, csrc/ss_gemm.hip SPLIT) and the 8-slot decode GEMVs through the split-bf16 MFMA form (csrc/ss_gemv.hip
        # gemv_split_f32_kernel: ONE sweep of the fp32 weights per token, 2 x the bf16 bytes; the decode graph is keyed on the knob)
        _lib.set_tuning("gemm_f32_split", 1)
        try:
            vg, msg = mllm_rate(e32, rin32, rout32, vit32)
            gm = {"ms_per_round_mllm": round(msg, 1), "value_mllm_only": round(vg, 4), "unit": "story-steps/s", "stories": SPG,
                  "arithmetic": "MLLM half + regressor on fp32 tensors (fp32 weights, fp32 KV cache): GEMMs AND the lock-step decode GEMVs as "
                                "Ahi*Whi + Ahi*Wlo + Alo*Whi on the bf16 matrix pipe with fp32 accumulation (4.5e-6 per product vs the fp64 "
                                "product); attention / norms / RoPE exact fp32; render bf16",
                  "gate": "img_gen_feat <= 1e-3 vs the REAL reference rows at hidden 4096: tests/test_frontend_full_gpu.py::test_gate_mode_*"}
            if adapter is not None:
                vfull, msfull, fb = full_rate(e32, rin32, rout32, vit32)
                gm.update({"value_full_pipeline": round(vfull, 4), "ms_per_round_full_pipeline": round(msfull, 1),
                           "value_bf16": value_bf16, "overlap_fallback": fb})
            out["gate_mode"] = gm
        finally:
            _lib.set_tuning("gemm_f32_split", 0)
        del e32, rin32, rout32, vit32
        torch.cuda.empty_cache()
    except Exception as ex:
        out.setdefault("mllm_fp32", {})["error"] = repr(ex)[:300]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--story-len", type=int, default=10, help="story steps per story (10 = the StoryStream chunk of the metric; 5 = configs[2])")
    ap.add_argument("--kv-reuse", action="store_true", help="65-row KV-cached continuation instead of re-prefill")
    ap.add_argument("--sink", action="store_true",
                    help="BASELINE configs[4]: the multimodal attention sink of vis_george_sink.py:243-295 on the KV slab — 65-row "
                         "continuation every step and, past the window, eviction by ss_llama_kv_gather (sink prefix + 12 rows around "
                         "the evicted image's <img> and </img> kept) instead of cutting the prompt and re-prefilling the window; "
                         "use with --story-len 25")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config0", action="store_true", help="BASELINE configs[0]: the CPU plumbing case (S=256, 1 image placeholder, full 7B dims) on the host, no GPU")
    ap.add_argument("--no-roofline", action="store_true", help="flow tests only: skip the roofline section (the JSON line is then not a bench record)")
    ap.add_argument("--no-tolerance-modes", action="store_true",
                    help="skip the extra legs that price the reference-arithmetic modes (fp32 VAE decode, fp32 MLLM half)")
    ap.add_argument("--no-batch1", action="store_true", help="skip the additional 1-story-per-GPU (reference batch-1) measurement")
    ap.add_argument("--mllm-only", action="store_true", help="BASELINE configs[1]: no SDXL render, 3-pair stories")
    ap.add_argument("--diffusion-steps", type=int, default=30)
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the MLLM half and the render of a round back to back (default: the next round's MLLM half "
                         "runs on a second HIP stream under the current round's render)")
    ap.add_argument("--stories-per-gpu", type=int, default=None, choices=[1, 2, 3, 4, 6, 8, 12, 16],
                    help="stories resident per GPU, advanced in lock-step (1 = the reference's batch-1 loop); more than 4 "
                         "run as groups of <= 4 decode slots over shared weights and ONE render batch (UNet batch 2 x stories).  "
                         "Default: 8 for the replica partition (round 4, same box: 1.995 / 1.996 story-steps/s against 1.927 / 1.935 "
                         "with 4 — the batch-16 UNet forward costs 59.0 ms per 8 samples against 61.1), 4 for the slot ring; 12 / 16 = two decode "
                         "groups and a UNet batch of 24 / 32 (not measured yet: their shapes are tuned in-process)")
    ap.add_argument("--max-slots", type=int, default=0,
                    help="sequence slots per decode engine (stories sharing one sweep of the weights per token): 1..8; "
                         "default 8 (3 - 8 slots decode through the MFMA form of the GEMV)")
    ap.add_argument("--render-groups", type=int, default=0,
                    help="render the round's images as this many independent batches on separate HIP streams at once (de-tokenizer "
                         "replicas over equal weights); default 1.  Measured with 8 resident stories: 2 groups of 4 = 1.955 story-steps/s, one "
                         "batch of 16 = 2.035 on a faster box, 4 stories = 1.92 / 1.98 — the two forwards overlap (59.5 vs 63.5 ms "
                         "per forward alone) but share the GPU with the two MLLM halves of the round")
    ap.add_argument("--partition", choices=["replicas", "slots"], default="replicas",
                    help="N > 1: 'replicas' = independent stories per rank (throughput mode, no data-path collective); "
                         "'slots' = ONE story stream per node: rank 0 runs the MLLM recurrence, image slot t is rendered "
                         "on rank 1 + t mod (N-1) (RCCL send of img_gen_feat), BASELINE configs[3]")
    ap.add_argument("--unet-fp8", action="store_true",
                    help="BASELINE configs[4]: run the UNet's transformer-block linear layers through the fp8 (OCP e4m3) "
                         "MFMA GEMM (row-wise dynamic activation scales); the headline number is the bf16 default")
    ap.add_argument("--no-splitk", action="store_true",
                    help="A/B: run the 128 < M <= 512 LLaMA projections (stacked image-token block, first prompts) through the "
                         "regular GEMM tiles instead of the split-K weight-streaming path")
    ap.add_argument("--save-tune-table", default=None, help="write the GEMM tile table of this run to this JSON path")
    ap.add_argument("--knob", action="append", default=[], metavar="NAME=INT",
                    help="A/B runs: set a library tuning knob before anything is built (e.g. --knob attn_ver=6); recorded in "
                         "config.knobs — a line with knobs is not the shipped configuration")
    args = ap.parse_args()
    if args.max_slots:
        global MAX_SLOTS
        MAX_SLOTS = max(1, min(8, args.max_slots))
    if args.config0:
        print(json.dumps(run_config0()))
        return
    knobs = {}
    for kv in args.knob:
        name, _, val = kv.partition("=")
        knobs[name.strip()] = int(val)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("SS_BENCH_SINGLE_DEVICE") or os.environ.get("SS_BENCH_SHARE_DEVICE"):
        # flow tests of the N > 1 path on a 1-GPU box: all ranks on cuda:0 — SINGLE_DEVICE under gloo, SHARE_DEVICE under
        # nccl (RCCL refuses two ranks on one device, "Duplicate GPU detected": the test records that and falls back)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    force_dist = world == 1 and bool(os.environ.get("SS_BENCH_FORCE_DIST"))   # 1-GPU box: still go through RCCL (world size 1)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if os.environ.get("SS_BENCH_SINGLE_DEVICE"):
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    dtype = torch.bfloat16
    if knobs:
        from seedstory import _lib as _lk
        for name, val in knobs.items():
            _lk.set_tuning(name, val)
    if args.no_splitk:
        from seedstory import _lib as _l
        _l.set_tuning("gemm_splitk", 0)
    global STORY_LEN, SINK, CACHE_CAP
    STORY_LEN = 3 if args.mllm_only else args.story_len
    SINK = bool(args.sink)
    if SINK and ((world > 1 or force_dist) and args.partition == "slots"):
        raise SystemExit("--sink is a replica-partition mode (the slot ring ships KV rows between ranks with its own eviction rule)")
    if SINK and args.kv_reuse:
        raise SystemExit("--sink already implies the 65-row continuation; drop --kv-reuse")
    if SINK:
        CACHE_CAP = (4 + 24 * max(0, STORY_LEN - WINDOW) + 1 + 114 * (WINDOW + 1) + 128 + 127) // 128 * 128
    slots_mode = (world > 1 or force_dist) and args.partition == "slots"
    if args.stories_per_gpu is None:
        args.stories_per_gpu = 4 if slots_mode else 8
    SPG = args.stories_per_gpu
    if slots_mode:
        from seedstory import parallel
        return parallel.bench_slot_partition(args, rank, world, device, dtype, sys.modules[__name__])

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
            flush_c_stdio()
        torch.cuda.synchronize()

    engs, shared = build_engines(device, dtype, SPG)
    eng = engs[0]                                   # the roofline section profiles the first decode group
    GRP = eng.n_seq
    rin, rout, vit = build_frontend(device, dtype)
    RG = args.render_groups if args.render_groups > 0 else 1
    if SPG % RG:
        raise SystemExit("--render-groups must divide --stories-per-gpu")
    adapters = [] if args.mllm_only else [build_detokenizer(device, dtype, vit) for _ in range(RG)]
    adapter = adapters[0] if adapters else None
    for a_ in adapters:
        if args.unet_fp8:
            a_.unet.enable_fp8(True)
    runner = Runner(engs if len(engs) > 1 else eng, rin, rout, vit, (adapters if RG > 1 else adapter), SPG, device, args,
                    rank * 100003)

    # Tile-table entries (seedstory/tune.py) must exist before the timed region whatever --warmup is: the prompt grows
    # by 114 rows per story step (prefill GEMM M buckets of 128), and one round touches every other shape (ViT,
    # resamplers, UNet, VAE).  Shapes already in the shipped table cost nothing here.
    for i in range(STORY_LEN):
        rows = 65 if ((args.kv_reuse and 0 < i < WINDOW) or (SINK and i > 0)) else prompt_len(i)
        for b in range(GRP):
            eng.select(b).reset()
        if GRP == 1:
            eng.prefill(torch.zeros(rows, H, device=device, dtype=dtype))
        else:                                   # the stacked prefill of a lock-step group: M = GRP x rows
            eng.prefill_batch([torch.zeros(rows, H, device=device, dtype=dtype)] * GRP)
    if GRP > 1:
        for b in range(GRP):
            eng.select(b).reset()
        eng.prefill_batch([torch.zeros(66, H, device=device, dtype=dtype)] * GRP)      # the image-token block (GRP x 66 rows)
    for b in range(GRP):
        eng.select(b).reset()
    eng.select(0)
    global RENDER_CONCURRENT
    RENDER_CONCURRENT = False
    runner.one_step()
    RENDER_CONCURRENT = True
    runner.sts = None
    runner.warm(args.warmup)
    barrier()
    from seedstory import tune as _tt0
    tuned_before = len(_tt0.tuned_log())            # shapes the shipped table lacked, tuned before the clock starts
    t0 = time.perf_counter()
    runner.run(args.steps)
    barrier()
    dt_s = time.perf_counter() - t0
    sink_snapshot = [x for st in (runner.sts or []) for x in st.sink_log] if SINK else None
    per_rank = None
    if world > 1 or force_dist:
        cdev = "cpu" if os.environ.get("SS_BENCH_SINGLE_DEVICE") else device
        mine = torch.tensor([dt_s, float(args.steps * SPG)], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                       # every rank's own clock and story-step count, through the communicator
        per_rank = [{"rank": i, "seconds": round(float(v[0]), 4), "story_steps": int(v[1]),
                     "story_steps_per_s": round(float(v[1]) / float(v[0]), 4)} for i, v in enumerate(every)]
        t = torch.tensor([dt_s], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_s = float(t.item())
    overlap = runner.overlap

    # ---- the reference's batch-1 configuration (1 story per GPU), same model, same schedule, rank 0 at N = 1 --------
    batch1 = None
    if rank == 0 and world == 1 and SPG > 1 and not args.no_batch1:
        eng1, _ = build_engine(device, dtype, 1, shared)
        r1 = Runner(eng1, rin, rout, vit, adapter, 1, device, args, 777000)
        r1.one_step()
        r1.sts = None
        r1.warm(1)
        n1 = STORY_LEN                                  # one whole story
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r1.run(n1)
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t1
        batch1 = {"value": round(n1 / d1, 4), "unit": "story-steps/s", "stories_per_gpu": 1, "steps": n1,
                  "ms_per_story_step": round(d1 / n1 * 1e3, 1), "mllm_render_overlap": bool(r1.overlap)}
        del eng1, r1

    roof = measure_roofline(eng, adapter, SPG, RG, device, dtype, args) if (rank == 0 and not args.no_roofline) else None
    tol_modes = None
    if rank == 0 and world == 1 and not args.no_tolerance_modes:
        tol_modes = measure_tolerance_modes(runner, engs, shared, rin, rout, vit, (adapters if RG > 1 else adapter), SPG, device,
                                            dtype, args, round(args.steps * world * SPG / dt_s, 4))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(with_sdxl=not args.mllm_only, diffusion_steps=args.diffusion_steps)
    if rank == 0:
        from seedstory import tune as _tt
        total_steps = args.steps * world * SPG
        if args.mllm_only:
            workload = ("BASELINE configs[1]: LLaMA-7B MLLM (prefill S=115/229/343 + 115 generated tokens per step: 50 iterated "
                        "in the device loop + 65 processor-forced image tokens as one stacked block) + Qwen "
                        "ViT-G encode per story + input/output Resampler regression, bf16, 3 image-text pairs, no SDXL")
            metric = "story-steps/sec (text + image-feature regression, MLLM half only)"
        else:
            workload = ("%s: full pipeline on 1 GPU per replica — LLaMA-7B MLLM (prefill S=%d..%d + 115 generated tokens per "
                        "step: 50 iterated in the device loop + 65 processor-forced image tokens as one stacked block) + Qwen "
                        "ViT-G encode + Resampler regression + SDXL de-tokenizer (ResamplerXLV2, %d "
                        "Euler steps x CFG batch 2 UNet, VAE decode) -> 1024x1024 uint8 image, bf16, story length %d, "
                        "%d-image context window (%s from step %d on)"
                        % ("the metric's 10-seq StoryStream chunk (BASELINE configs[3] workload per story)" if STORY_LEN == 10
                           else "BASELINE configs[2]" if STORY_LEN == 5 else "story length %d" % STORY_LEN,
                           prompt_len(0), prompt_len(0) if SINK else prompt_len(STORY_LEN - 1), args.diffusion_steps, STORY_LEN, WINDOW,
                           "multimodal ATTENTION SINK on the KV slab, vis_george_sink.py:243-295 as intended: 65-row continuation per step, "
                           "oldest image evicted by ss_llama_kv_gather with the first 4 positions + 12 rows around its <img> and </img> kept"
                           if SINK else "oldest pair evicted and the window re-prefilled", WINDOW))
            metric = "story-steps/sec (text + 1024x1024 image)"
        sink_info = None
        if SINK:
            log = sink_snapshot or []
            row_bytes = NL * 2 * H * 2                                     # one KV position: K and V of every layer, bf16
            sink_info = {"cache_cap_rows": CACHE_CAP, "evictions_in_last_round_stories": len(log),
                         "kv_rows_kept_per_eviction": (round(sum(k for k, _ in log) / len(log), 1) if log else None),
                         "kv_rows_dropped_per_eviction": (round(sum(d for _, d in log) / len(log), 1) if log else None),
                         "kv_bytes_repacked_per_eviction": (int(sum(k for k, _ in log) / len(log) * row_bytes) if log else None),
                         "kv_bytes_per_position": row_bytes,
                         "note": "ss_llama_kv_gather packs the kept rows through the engine's scratch (one gather + one copy-back per "
                                 "layer plane: 2 x the bytes above read and written); the re-prefill this replaces streamed 13.2 GB of "
                                 "weights over 913 rows per story"}
        out = {"metric": metric,
               "value": round(total_steps / dt_s, 4), "unit": "story-steps/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(dt_s / args.steps * 1e3, 3), "higher_is_better": True,
               "story_steps_per_step": SPG, "mllm_render_overlap": bool(overlap), "render_groups": RG,
               "overlap_fallback": runner.overlap_fallback is not None, "overlap_fallback_error": runner.overlap_fallback,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": workload, "unet_linear_dtype": "fp8_e4m3" if args.unet_fp8 else "bf16", "diffusion_steps": None if args.mllm_only else args.diffusion_steps,
                          "kv_reuse": bool(args.kv_reuse), "attention_sink": sink_info, "tokens_per_step": T_GEN, "knobs": knobs or None,
                          "img_block_decode": bool(eng.img_block_enabled()),
                          "img_block_decode_note": "the 65 tokens the logits processor forces behind <img> are fed as ONE stacked "
                                                   "continuation for all lock-step slots of a decode group (same layers / attention "
                                                   "/ hidden rows / KV entries / per-position lm_head as the token-by-token loop, "
                                                   "weights streamed once per group); the prompts are prefilled as one stacked "
                                                   "forward too; SEEDSTORY_IMG_BLOCK=0 restores the token-by-token loop",
                          "stories_per_gpu": SPG,
                          "step_definition": "one lock-step round of the %d resident stories = %d story-steps" % (SPG, SPG),
                          "decode_groups": slot_groups(SPG),
                          "parallelism": "story replicas x%d, %d lock-step story slots per GPU (decode groups of %s over shared "
                                         "weights, %d concurrent render batch(es) of %d on separate HIP streams)"
                                         % (world, SPG, slot_groups(SPG), RG, 2 * SPG // RG)},
               "batch1": batch1,
               "rccl_ranks": (dist.get_world_size() if (world > 1 or force_dist) else 1),
               "collective_backend": (dist.get_backend() if (world > 1 or force_dist) else None),
               "per_rank": per_rank,
               "tolerance_modes": tol_modes,
               "tile_table": {"entries": len(_tt.export_table()), "tuned_in_this_process": len(_tt.tuned_log()),
                              "tuned_before_timed_region": tuned_before,
                              "tuned_shapes": [[k, list(sh), c, x, round(us, 1)] for k, sh, c, x, us in _tt.tuned_log()],
                              "note": "GEMM/conv tile choices come from seedstory/tune_gfx950.json; shapes missing from it are "
                                      "tuned explicitly (ss_gemm_tune) BEFORE the timed region"},
               "roofline": roof, "cpu_baseline": cpu}
        if args.save_tune_table:
            _tt.save_table(args.save_tune_table, note="written by bench.py")
        print(json.dumps(out))
    if world > 1 or force_dist:
        # rank 0 spends another ~30 s on the roofline section after the timed region: every rank waits for it here, so that
        # the communicators are torn down together (a rank that destroys its group while rank 0 still owns live
        # communicators can leave rank 0 hanging at exit)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
