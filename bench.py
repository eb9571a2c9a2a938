#!/usr/bin/env python
"""bench.py — story-steps/sec of the MI355X-native SEED-Story hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (default = BASELINE.json configs[2], the single-GPU configuration the metric
"story-steps/sec (text + 1024x1024 image)" is quoted on; SURVEY.md §8d synthetic schedule):
LLaMA-2-7B-shaped MLLM (random N(0,0.02) weights, bf16) + Qwen ViT-G encode + learnable-query
image-feature regression + SDXL de-tokenizer (ResamplerXLV2 -> 30-step Euler, CFG 7.5, 1024x1024,
random-init SDXL-base UNet/VAE, bf16), stories of 5 steps.  ``--mllm-only`` runs configs[1] (3 pairs,
no SDXL).  One *step* = one ``agent.generate`` of the reference
(gen_george.py:189/257): embed + splice the window's image features (input resampler over every
image in context) -> prefill of the whole prompt (S = 115 / 229 / 343; "as released", no KV reuse;
``--kv-reuse`` switches to the 65-row continuation) -> 115 greedy decode iterations under the forced
token schedule (48 caption ids, ``<img>``, 64 image tokens + ``</img>`` forced by the reference's
logits processor, EOS) -> output resampler regression of the 64 hidden states to the 256x4096 image
feature -> ``adapter.generate`` (gen_george.py:210): ResamplerXLV2 conditioning + 30 x {UNet (batch 2,
CFG) + Euler update} + VAE decode to a uint8 1024x1024 image.  The first step of every story also
encodes the 448x448 input image with ViT-G (and the constant all-zeros negative image once).

``--stories-per-gpu S`` (default 4): S independent stories are resident on each GPU and advance in
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
STORY_LEN = 5
T_GEN = CAPTION + 66 + 1                 # caption + image tokens + EOS = 115 decode iterations


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


def build_models(device, dtype, n_seq=1):
    from seedstory.llama import LlamaEngine
    from src.models.qwen_visual import Resampler, VisionTransformerWithAttnPool
    torch.manual_seed(1234)

    def rnd(*s):
        return torch.randn(*s, device=device, dtype=dtype) * 0.02

    ones = lambda n: torch.ones(n, device=device, dtype=dtype)  # noqa: E731
    layers = [(rnd(3 * H, H), rnd(H, H), rnd(2 * INTER, H), rnd(H, INTER), ones(H), ones(H)) for _ in range(NL)]
    eng = LlamaEngine.from_prebuilt(embed=rnd(VOCAB, H), lm_head=rnd(VOCAB, H), final_norm=ones(H), layers=layers,
                                    hidden=H, n_heads=NH, n_layers=NL, inter=INTER, vocab=VOCAB, dtype=dtype,
                                    device=device, cache_cap=1024, max_new=128, max_prefill_rows=640, img_ids=IMG_IDS,
                                    eos_id=EOS, n_seq=n_seq)
    rin = Resampler(grid_size=8, embed_dim=H, num_heads=32, kv_dim=H).to(device=device, dtype=dtype).init_synthetic(1)
    rout = Resampler(grid_size=16, embed_dim=H, num_heads=32, kv_dim=H).to(device=device, dtype=dtype).init_synthetic(2)
    vit = VisionTransformerWithAttnPool(image_size=448, patch_size=14, width=1664, layers=48, heads=16,
                                        mlp_ratio=4.9231, output_dim=4096).to(device=device, dtype=dtype)
    vit.init_synthetic(3)
    return eng, rin, rout, vit


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

    def forced(self):
        cap = torch.randint(3, 32000, (CAPTION,), generator=self.g).tolist()
        return cap + IMG_IDS + [EOS]


def mllm_part(sts, eng, rin, rout, vit, kv_reuse):
    """The MLLM half of one multimodal step of every resident story (slot b of the engine = story sts[b]; all
    stories of a round are at the same step index): ``agent.generate`` of gen_george.py:189/257.  Advances the
    stories' context (ids, image features) and returns img_gen_feat [S,256,4096]."""
    from seedstory import ops
    dev = sts[0].device
    for b, st in enumerate(sts):
        eng.select(b)
        if st.step == 0:
            st.image_embeds = vit(st.image)                               # [1,256,4096]  gen_george.py:187-188
        ids = torch.tensor(st.ids, dtype=torch.int32, device=dev)
        emb = ops.gather_rows(eng.embed, ids)                             # models.py:127
        lm = rin(st.image_embeds)                                         # [Nimg,64,H]   models.py:133
        pos = [i + 1 for i, t in enumerate(st.ids) if t == IMG_IDS[0]]
        idx = torch.tensor([p + j for p in pos for j in range(64)], dtype=torch.int32, device=dev)
        ops.scatter_rows_(emb, idx, lm.reshape(-1, H))                    # models.py:135
        S = len(st.ids)
        if kv_reuse and st.step > 0:
            keep = S - 65                                                 # ... caption + <img> stay cached
            eng.set_lengths(keep, keep)
            eng.prefill(emb[keep:])
        else:
            eng.reset()
            eng.prefill(emb)
    forced = [st.forced() for st in sts]
    if len(sts) == 1:
        ns = [eng.generate(500, sts[0].ids[-1], forced[0])]               # max_new_tokens=500 (gen_george.py:194)
    else:
        ns = eng.generate_batch(500, [st.ids[-1] for st in sts], forced)  # the same loop, all slots per weight sweep
    assert all(n == T_GEN for n in ns), ns
    e = CAPTION + 65                                                      # index of </img> in the generated ids
    feats = torch.stack([eng.select(b).hidden_rows[e - 64:e] for b in range(len(sts))]).contiguous()   # models.py:197
    img_gen_feat = rout(feats)                                            # models.py:205  [S,256,4096]
    for b, st in enumerate(sts):
        st.image_embeds = torch.cat([st.image_embeds, img_gen_feat[b:b + 1]], dim=0)   # gen_george.py:224
        st.ids = st.ids + forced[b][:CAPTION] + IMG_IDS                   # prompt + text + image_tokens (:231)
        st.step += 1
    return img_gen_feat


def sdxl_part(sts, adapter, img_gen_feat, steps):
    """The de-tokenizer half: ``adapter.generate`` (gen_george.py:210) for the S images of the round."""
    imgs = adapter.generate(image_embeds=img_gen_feat, num_inference_steps=steps, output_type="pt")
    imgs = imgs.unsqueeze(0) if len(sts) == 1 else imgs
    for b, st in enumerate(sts):
        st.last_image = imgs[b]
    return imgs


def run_round(sts, eng, rin, rout, vit, kv_reuse, adapter=None, steps=30):
    """One multimodal step of every resident story, sequentially (MLLM half, then the render)."""
    img_gen_feat = mllm_part(sts, eng, rin, rout, vit, kv_reuse)
    if adapter is not None:
        sdxl_part(sts, adapter, img_gen_feat, steps)
    return img_gen_feat


def run_step(st, eng, rin, rout, vit, kv_reuse, adapter=None, steps=30):
    return run_round([st], eng, rin, rout, vit, kv_reuse, adapter, steps)


def cpu_baseline(seconds_budget=25.0, with_sdxl=True, diffusion_steps=30):
    """The oracle (CPU restatement of the reference, 'port') timed on this box's host cores on a
    BOUNDED sample: a 2-layer full-width (4096/11008/32 heads) bf16 slice — one S=115 prefill and 4
    decode tokens — extrapolated to 32 layers + lm_head and to the 3-step story schedule."""
    import seedstory_oracle as O
    threads = torch.get_num_threads()
    L = 2
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(0)

    def rnd(*s):
        return (torch.randn(*s, generator=g) * 0.02).to(dt)

    wd = {"model.embed_tokens.weight": rnd(1024, H), "lm_head.weight": rnd(VOCAB, H), "model.norm.weight": torch.ones(H, dtype=dt)}
    for l in range(L):
        p = "model.layers.%d." % l
        for n, (o, i) in (("self_attn.q_proj", (H, H)), ("self_attn.k_proj", (H, H)), ("self_attn.v_proj", (H, H)),
                          ("self_attn.o_proj", (H, H)), ("mlp.gate_proj", (INTER, H)), ("mlp.up_proj", (INTER, H)),
                          ("mlp.down_proj", (H, INTER))):
            wd[p + n + ".weight"] = rnd(o, i)
        wd[p + "input_layernorm.weight"] = torch.ones(H, dtype=dt)
        wd[p + "post_attention_layernorm.weight"] = torch.ones(H, dtype=dt)
    dims = O.LlamaDims(H, NH, L, INTER, VOCAB)
    S = 115
    emb = rnd(1, S, H)
    t0 = time.perf_counter()
    with torch.no_grad():
        _, _, kv = O.llama_forward(wd, dims, emb, torch.arange(S).unsqueeze(0), None, all_logits=False)
        t_prefill = time.perf_counter() - t0
        t0 = time.perf_counter()
        ntok = 0
        while ntok < 4 and (time.perf_counter() - t0) < seconds_budget:
            _, _, kv = O.llama_forward(wd, dims, rnd(1, 1, H), torch.tensor([[S + ntok]]), kv, all_logits=False)
            ntok += 1
        t_tok = (time.perf_counter() - t0) / max(ntok, 1)
    # per-layer costs (lm_head share measured separately is folded in: it ran once per call)
    layer_tok = t_tok / (L + 0.65)          # lm_head = 263 MB ~ 0.65 of a 404 MB layer
    tok_full = layer_tok * (NL + 0.65)
    prefill_full_115 = t_prefill / (L + 0.65 / S) * NL
    # 3-step story: prefill S = 115, 229, 343 (linear in S at these sizes) + 115 tokens each
    lens = [115 + 114 * i for i in range(STORY_LEN)]
    step_s = (prefill_full_115 * sum(lens) / 115.0 / len(lens)) + T_GEN * tok_full
    sample = ("oracle llama_forward, bf16, 2 full-width layers: one S=115 prefill (%.2fs) + %d decode tokens "
              "(%.3fs/token), extrapolated to 32 layers+lm_head and the %d-step story" % (t_prefill, ntok, t_tok, STORY_LEN))
    if with_sdxl:
        import sdxl_oracle as S
        c = S.SDXL_BASE_UNET
        t0 = time.perf_counter()
        wdu = {k: torch.empty(*shp).normal_(0.0, 0.02) if len(shp) > 1 else torch.ones(*shp)
               for k, shp in S.unet_shapes(c).items()}
        t_w = time.perf_counter() - t0
        x = torch.randn(1, 4, 64, 64)
        ctx = torch.randn(1, 64, 2048)
        pooled = torch.randn(1, 1280)
        tid = torch.tensor([[512, 512, 0, 0, 512, 512]], dtype=torch.float32)
        t0 = time.perf_counter()
        with torch.no_grad():
            S.unet_forward(wdu, c, x, torch.tensor(500.0), ctx, pooled, tid)
        t_unet = time.perf_counter() - t0
        del wdu
        # 1024^2 = 4x the pixels of the sample (conv/FF scale 4x; self-attention 16x, ~11 % of the MACs): x4.4;
        # x2 CFG branches x diffusion steps; VAE decode (10.5 TFLOP fp32) priced at the UNet sample's flop rate
        unet_full = t_unet * 4.4
        flop_rate = 1.69e12 / t_unet
        render_s = diffusion_steps * 2 * unet_full + 10.5e12 / flop_rate
        step_s += render_s
        sample += ("; + oracle SDXL-base UNet forward fp32, batch 1 at 64x64 latents (%.1fs; weights %.0fs), extrapolated "
                   "to 128x128 latents x2 (CFG) x%d Euler steps + VAE decode at the same flop rate" % (t_unet, t_w, diffusion_steps))
    return {"value": round(1.0 / step_s, 6), "unit": "story-steps/s", "cores": threads, "kind": "port",
            "sample": sample + " (ViT/resamplers excluded, <2% of the step)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--kv-reuse", action="store_true", help="65-row KV-cached continuation instead of re-prefill")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mllm-only", action="store_true", help="BASELINE configs[1]: no SDXL render, 3-pair stories")
    ap.add_argument("--diffusion-steps", type=int, default=30)
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the MLLM half and the render of a round back to back (default: the next round's MLLM half "
                         "runs on a second HIP stream under the current round's render)")
    ap.add_argument("--stories-per-gpu", type=int, default=4, choices=[1, 2, 3, 4],
                    help="stories resident per GPU, advanced in lock-step (1 = the reference's batch-1 loop)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("SS_BENCH_SINGLE_DEVICE"):      # flow test of the N>1 path on a 1-GPU box (gloo, all ranks on cuda:0)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("SS_BENCH_SINGLE_DEVICE"):
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    dtype = torch.bfloat16
    global STORY_LEN
    if args.mllm_only:
        STORY_LEN = 3
    SPG = args.stories_per_gpu
    eng, rin, rout, vit = build_models(device, dtype, SPG)
    adapter = None if args.mllm_only else build_detokenizer(device, dtype, vit)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    story_no = [rank * 100003]
    sts = [None]

    def one_step():
        if sts[0] is None or sts[0][0].step >= STORY_LEN:
            sts[0] = []
            for _ in range(SPG):
                story_no[0] += 1
                sts[0].append(Story(story_no[0], device))
        return run_round(sts[0], eng, rin, rout, vit, args.kv_reuse, adapter, args.diffusion_steps)

    # The GEMM tile autotuner (first call of a new shape) must not run inside the timed region whatever --warmup is:
    # the prompt grows by 114 rows per story step, so every step has its own prefill GEMM shapes.  Touch them once.
    for i in range(STORY_LEN):
        S_i = 115 + 114 * i
        eng.select(0).reset()
        eng.prefill(torch.zeros(65 if (args.kv_reuse and i > 0) else S_i, H, device=device, dtype=dtype))
    eng.select(0).reset()
    one_step()                      # + every other shape of a round (ViT, resamplers, UNet, VAE), whatever --warmup is
    sts[0] = None
    # ---- pipelined schedule: the render of round r does not feed round r+1's MLLM half (the context takes the
    # regressed FEATURE, gen_george.py:224, not the decoded image), so the HBM-bound decode of round r+1 runs on a
    # second HIP stream, driven by a second host thread, under the MFMA-bound UNet loop of round r.  Work per timed
    # step is unchanged: K MLLM halves + K renders.
    import threading
    overlap = adapter is not None and not args.no_overlap
    side = torch.cuda.Stream(device=device) if overlap else None

    def next_stories():
        if sts[0] is None or sts[0][0].step >= STORY_LEN:
            sts[0] = []
            for _ in range(SPG):
                story_no[0] += 1
                sts[0].append(Story(story_no[0], device))
        return sts[0]

    def mllm_async(box):
        def work():
            try:
                torch.cuda.set_device(device)
                with torch.cuda.stream(side):
                    box["stories"] = next_stories()
                    box["feat"] = mllm_part(box["stories"], eng, rin, rout, vit, args.kv_reuse)
                side.synchronize()
            except BaseException as ex:   # surfaced by the driver thread (run_rounds)
                box["err"] = ex
        th = threading.Thread(target=work)
        th.start()
        return th

    def take(box):
        if "err" in box:
            raise box["err"]
        return box["stories"], box["feat"]

    mode = {"overlap": overlap}

    def run_rounds(n):
        if not mode["overlap"]:
            for _ in range(n):
                one_step()
            return
        if n <= 0:
            return
        torch.cuda.synchronize()
        box = {}
        mllm_async(box).join()                         # round 0's MLLM half has nothing to hide under
        for r in range(n):
            cur_stories, cur_feat = take(box)
            box = {}
            th = mllm_async(box) if r + 1 < n else None
            sdxl_part(cur_stories, adapter, cur_feat, args.diffusion_steps)
            if th is not None:
                th.join()
        torch.cuda.synchronize()

    try:
        run_rounds(args.warmup if (args.warmup > 0 or not overlap) else 1)   # >= 1 untimed round checks the schedule
    except Exception as ex:      # the two-stream schedule is an optimisation: never let it take the measurement down
        print("bench: overlapped schedule failed (%r); falling back to the sequential one" % (ex,), file=sys.stderr)
        mode["overlap"] = False
        torch.cuda.synchronize()
        sts[0] = None
        run_rounds(args.warmup)
    overlap = mode["overlap"]
    sts[0] = None  # timed region starts at a story boundary
    barrier()
    t0 = time.perf_counter()
    run_rounds(args.steps)
    barrier()
    dt_s = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt_s], dtype=torch.float64, device="cpu" if os.environ.get("SS_BENCH_SINGLE_DEVICE") else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_s = float(t.item())

    # ---- roofline of the dominant kernel (decode GEMV, HBM-bound), measured live with HIP events ----
    roof = None
    if rank == 0:
        for b in range(SPG):
            eng.select(b).set_lengths(343, 343)
        prof = eng.profile_decode(8)
        # dominant kernel = the decode weight-streaming GEMV.  With <= 2 slots per sweep the K=hidden projections
        # (qkv, o, gate|up per layer + lm_head: 97 launches/token) are ss::gemv_kernel<bf16,8,2,NB> and the down
        # projection a different symbol; with 3-4 slots every projection runs ss::gemv_ldsx_kernel<bf16,2,NB>
        # (129 launches/token), so the launch average is taken over all of them.
        if SPG <= 2:
            kern = "gemv_kernel<bf16_t,8,2,%d>" % SPG
            n_launch, tot_bytes, tot_ms = prof["gemv_launches"], prof["gemv_bytes"], prof["gemv_ms"]
        else:
            kern = "gemv_ldsx_kernel<bf16_t,2,%d>" % SPG
            n_launch = prof["gemv_launches"] + prof["gemv_down_launches"]
            tot_bytes = prof["gemv_bytes"] + prof["gemv_down_bytes"]
            tot_ms = prof["gemv_ms"] + prof["gemv_down_ms"]
        per_launch_bytes = tot_bytes / n_launch
        per_launch_ms = tot_ms / n_launch
        achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
        down = prof["gemv_down_bytes"] / (prof["gemv_down_ms"] * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "round1_pmc_summary.json")
        if os.path.exists(pmc):      # HBM bytes per launch from the separate rocprofv3 --pmc passes of this command
            try:
                pj = json.load(open(pmc))      # counters were collected at pj["stories_per_gpu"] slots per sweep
                if pj.get("stories_per_gpu") == SPG:
                    traffic = pj.get("gemv_hbm_traffic", {}).get(kern, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                "kernel": "ss::" + kern, "slots_per_sweep": SPG,
                "launches_per_token": n_launch, "bytes_per_launch": round(per_launch_bytes),
                "avg_launch_us": round(per_launch_ms * 1e3, 3),
                "also": {"down projection GB/s": round(down, 1),
                         "all_gemv GB/s per token": round((prof["gemv_bytes"] + prof["gemv_down_bytes"]) /
                                                          ((prof["gemv_ms"] + prof["gemv_down_ms"]) * 1e-3) / 1e9, 1),
                         "token_ms_eager": round(prof["token_ms"], 4), "attn_ms": round(prof["attn_ms"], 4),
                         "misc_ms": round(prof["misc_ms"], 4),
                         "weight_bytes_per_generated_token_per_story": round(13.215e9 / SPG)}}
    if rank == 0 and adapter is not None:
        # MFMA-bound half: one SDXL-base UNet forward (batch 2S = CFG pairs of the S resident stories, 128x128
        # latents), HIP events on the stream
        UB = 2 * SPG
        x = torch.randn(UB, 4, 128, 128, device=device, dtype=dtype)
        ctx = torch.randn(UB, 64, 2048, device=device, dtype=dtype)
        cond = {"text_embeds": torch.randn(UB, 1280, device=device, dtype=dtype),
                "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * UB, dtype=torch.float32)}
        adapter.unet(x, 500.0, ctx, added_cond_kwargs=cond)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            adapter.unet(x, 500.0, ctx, added_cond_kwargs=cond)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
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
        del ag, wg
        roof_mllm = roof
        roof = {"bound": "mfma", "achieved": round(flops / (ms * 1e-3) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                "frac": round(flops / (ms * 1e-3) / 2.5e15, 4), "traffic": None,
                "kernel": "SDXL UNet forward, all kernels (ss::gemm_sp_kernel / gemm_glds_kernel<bf16,*> incl. implicit-GEMM conv3x3, ss::flash_attn2_kernel<bf16,64>, norms)",
                "flops_per_forward": flops, "forward_ms": round(ms, 3), "unet_batch": UB,
                "dominant_kernel": {"kernel": "ss::gemm_sp_kernel / gemm_glds_kernel (autotuned tile) + GEGLU epilogue",
                                    "shape_MNK": [Mg, Ng, Kg], "avg_launch_us": round(gemm_us, 1),
                                    "achieved": round(gemm_tf, 1), "unit": "TFLOP/s", "frac": round(gemm_tf / 2500.0, 4)},
                "note": "a round is 30 UNet forwards of batch %d (MFMA-bound) + 115 decode tokens for %d slots (HBM-bound): see mllm_decode_gemv" % (UB, SPG),
                "mllm_decode_gemv": roof_mllm}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(with_sdxl=not args.mllm_only, diffusion_steps=args.diffusion_steps)
    if rank == 0:
        total_steps = args.steps * world * SPG
        if args.mllm_only:
            workload = ("BASELINE configs[1]: LLaMA-7B MLLM (prefill S=115/229/343 + 115 greedy decode iterations) + Qwen "
                        "ViT-G encode per story + input/output Resampler regression, bf16, 3 image-text pairs, no SDXL")
            metric = "story-steps/sec (text + image-feature regression, MLLM half only)"
        else:
            workload = ("BASELINE configs[2]: full pipeline on 1 GPU per replica — LLaMA-7B MLLM (prefill S=115..571 + 115 "
                        "greedy decode iterations) + Qwen ViT-G encode + Resampler regression + SDXL de-tokenizer "
                        "(ResamplerXLV2, %d Euler steps x CFG batch 2 UNet, VAE decode) -> 1024x1024 uint8 image, bf16, "
                        "story length 5" % args.diffusion_steps)
            metric = "story-steps/sec (text + 1024x1024 image)"
        out = {"metric": metric,
               "value": round(total_steps / dt_s, 4), "unit": "story-steps/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(dt_s / args.steps * 1e3, 3), "higher_is_better": True,
               "story_steps_per_step": SPG, "mllm_render_overlap": bool(overlap),
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": workload, "diffusion_steps": None if args.mllm_only else args.diffusion_steps,
                          "kv_reuse": bool(args.kv_reuse), "tokens_per_step": T_GEN,
                          "stories_per_gpu": SPG,
                          "step_definition": "one lock-step round of the %d resident stories = %d story-steps" % (SPG, SPG),
                          "parallelism": "story replicas x%d, %d lock-step story slots per GPU" % (world, SPG)},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
