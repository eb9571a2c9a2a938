"""GPU-box diagnostics: kernel micro-benchmarks + knob sweeps, written to gpurun_out/diag.json.
Usage (on the GPU box): python tools/gpu_diag.py [gemv] [gemm] [attn] [decode]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch  # noqa: E402
from seedstory import _lib, ops  # noqa: E402

DEV = "cuda:0"
OUT = {}


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def bench_gemv():
    res = []
    shapes = [("qkv", 12288, 4096, {}), ("o", 4096, 4096, {}), ("gateup", 11008, 4096, {"silu_mul": True}),
              ("down", 4096, 11008, {}), ("lm_head", 32066, 4096, {})]
    # rotate over several weight copies so the 256 MiB Infinity Cache cannot serve the stream
    for name, N, K, kw in shapes:
        rows = N * 2 if kw.get("silu_mul") else N
        ncopy = max(2, int(600e6 // (rows * K * 2)) + 1)
        ws = [torch.randn(rows, K, device=DEV, dtype=torch.bfloat16) * 0.02 for _ in range(ncopy)]
        x = torch.randn(K, device=DEV, dtype=torch.bfloat16)
        nw = torch.ones(K, device=DEV, dtype=torch.bfloat16)
        for gpw in (0, 1, 2, 4):
            for nt in (1, 0):
                _lib.set_tuning("gemv_groups_per_wave", gpw)
                _lib.set_tuning("gemv_pipe", nt)
                st = {"i": 0}

                def f():
                    st["i"] += 1
                    ops.gemv(ws[st["i"] % ncopy], x, norm_w=nw if name in ("qkv", "gateup") else None, eps=1e-5, **kw)
                ms = timeit(f, iters=30)
                gbs = rows * K * 2 / ms / 1e6
                res.append(dict(name=name, N=N, K=K, gpw=gpw, pipe=nt, ms=round(ms, 4), GBps=round(gbs, 1)))
                print(res[-1], flush=True)
        del ws
    _lib.set_tuning("gemv_groups_per_wave", 0)
    _lib.set_tuning("gemv_pipe", 1)
    OUT["gemv"] = res


def bench_gemm():
    res = []
    shapes = [(343, 12288, 4096), (913, 12288, 4096), (913, 4096, 4096), (913, 22016, 4096), (913, 4096, 11008),
              (65, 12288, 4096), (65, 22016, 4096), (114, 4096, 11008), (1024, 4992, 1664), (1024, 1664, 1664),
              (1024, 8192, 1664), (1024, 1664, 8192), (256, 4096, 4096), (4096, 4096, 4096)]
    for M, N, K in shapes:
        a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=DEV, dtype=torch.bfloat16) * 0.02
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        for cfg in (0, 1, 2, 3):
            _lib.set_tuning("gemm_cfg", cfg)
            ms = timeit(lambda: ops.gemm(a, w, out=out), iters=10)
            res.append(dict(M=M, N=N, K=K, cfg=cfg, ms=round(ms, 4), TFLOPs=round(2.0 * M * N * K / ms / 1e9, 1),
                            GBps=round((N * K + M * K + M * N) * 2 / ms / 1e6, 1)))
            print(res[-1], flush=True)
    _lib.set_tuning("gemm_cfg", 0)
    OUT["gemm"] = res


def bench_attn():
    res = []
    for (B, H, hd, Lq, Lk, causal) in [(1, 32, 128, 343, 343, True), (1, 32, 128, 913, 913, True),
                                        (1, 32, 128, 65, 900, True), (1, 16, 104, 1024, 1024, False),
                                        (8, 32, 128, 64, 256, False), (8, 10, 64, 4096, 4096, False),
                                        (8, 20, 64, 1024, 1024, False), (8, 10, 64, 4096, 64, False),
                                        (8, 20, 64, 1024, 64, False), (2, 10, 64, 4096, 4096, False)]:
        E = H * hd
        q = torch.randn(B, Lq, E, device=DEV, dtype=torch.bfloat16)
        k = torch.randn(B, Lk, E, device=DEV, dtype=torch.bfloat16)
        v = torch.randn(B, Lk, E, device=DEV, dtype=torch.bfloat16)
        ms = timeit(lambda: ops.attention(q, k, v, H, None, causal), iters=10)
        fl = 4.0 * B * H * Lq * Lk * hd * (0.5 if causal and Lq == Lk else 1.0)
        res.append(dict(B=B, H=H, hd=hd, Lq=Lq, Lk=Lk, causal=causal, ms=round(ms, 4), TFLOPs=round(fl / ms / 1e9, 1)))
        print(res[-1], flush=True)
    kc = torch.randn(32, 2048, 128, device=DEV, dtype=torch.bfloat16)
    vc = torch.randn(32, 2048, 128, device=DEV, dtype=torch.bfloat16)
    q = torch.randn(4096, device=DEV, dtype=torch.bfloat16)
    for kv in (100, 500, 1000, 2000):
        for ns in (4, 8, 16):
            _lib.set_tuning("attn_decode_nsplit", ns)
            n = torch.tensor([kv], dtype=torch.int32, device=DEV)
            ms = timeit(lambda: ops.attn_decode(q, kc, vc, n), iters=20)
            res.append(dict(decode_kv=kv, nsplit=ns, ms=round(ms, 4), GBps=round(2 * 32 * kv * 128 * 2 / ms / 1e6, 1)))
            print(res[-1], flush=True)
    _lib.set_tuning("attn_decode_nsplit", 0)
    OUT["attn"] = res


def make_7b_engine(n_layers=32, cache_cap=2048, max_new=512, max_rows=1024, n_seq=1):
    from seedstory.llama import LlamaEngine
    H, I, V = 4096, 11008, 32066
    dt = torch.bfloat16

    def rnd(*s):
        return torch.randn(*s, device=DEV, dtype=dt) * 0.02

    layers = [(rnd(3 * H, H), rnd(H, H), rnd(2 * I, H), rnd(H, I), torch.ones(H, device=DEV, dtype=dt),
               torch.ones(H, device=DEV, dtype=dt)) for _ in range(n_layers)]
    return LlamaEngine.from_prebuilt(embed=rnd(V, H), lm_head=rnd(V, H), final_norm=torch.ones(H, device=DEV, dtype=dt),
                                     layers=layers, hidden=H, n_heads=32, n_layers=n_layers, inter=I, vocab=V, dtype=dt,
                                     device=DEV, cache_cap=cache_cap, max_new=max_new, max_prefill_rows=max_rows,
                                     img_ids=list(range(32000, 32066)), n_seq=n_seq)


def bench_slots():
    """Lock-step decode of n_seq story slots: per-token time and per-story token rate."""
    res = {}
    shapes = [("qkv", 12288, 4096, {}), ("gateup", 11008, 4096, {"silu_mul": True}), ("down", 4096, 11008, {})]
    for name, N, K, kw in shapes:
        rows = N * 2 if kw.get("silu_mul") else N
        ncopy = max(2, int(600e6 // (rows * K * 2)) + 1)
        ws = [torch.randn(rows, K, device=DEV, dtype=torch.bfloat16) * 0.02 for _ in range(ncopy)]
        for nb in (1, 2, 3, 4):
            x = torch.randn(nb, K, device=DEV, dtype=torch.bfloat16)
            for regpacks in (16, 32):
                if nb * (K // 512) <= 16 and regpacks == 32:
                    continue
                _lib.set_tuning("gemv_x_reg_packs", regpacks)
                st = {"i": 0}

                def f():
                    st["i"] += 1
                    ops.gemv_batched(ws[st["i"] % ncopy], x, **kw)

                def fn():
                    st["i"] += 1
                    ops.gemv_batched(ws[st["i"] % ncopy], x, norm_w=nw, eps=1e-5, **kw)
                nw = torch.ones(K, device=DEV, dtype=torch.bfloat16)
                ms = timeit(f, iters=30)
                msn = timeit(fn, iters=30)
                res["gemv_%s_nb%d_reg%d" % (name, nb, regpacks)] = dict(ms=round(ms, 4), GBps=round(rows * K * 2 / ms / 1e6, 1), norm_GBps=round(rows * K * 2 / msn / 1e6, 1))
                print(name, nb, regpacks, res["gemv_%s_nb%d_reg%d" % (name, nb, regpacks)], flush=True)
        del ws
    _lib.set_tuning("gemv_x_reg_packs", 16)
    for n_seq in (1, 2, 4):
        eng = make_7b_engine(n_seq=n_seq)
        S = 343
        for b in range(n_seq):
            eng.select(b).prefill(torch.randn(S + 3 * b, 4096, device=DEV, dtype=torch.bfloat16) * 0.02)
        forced = [torch.randint(3, 32000, (115,)).tolist() for _ in range(n_seq)]
        for regpacks in ((16,) if n_seq <= 2 else (16, 32)):
            _lib.set_tuning("gemv_x_reg_packs", regpacks)
            eng.generate_batch(8, [5] * n_seq, [f[:8] for f in forced])
            for b in range(n_seq):
                eng.select(b).set_lengths(S + 3 * b, S + 3 * b)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ns = eng.generate_batch(115, [5] * n_seq, forced)
            torch.cuda.synchronize()
            dtm = time.perf_counter() - t0
            for b in range(n_seq):
                eng.select(b).set_lengths(S + 3 * b, S + 3 * b)
            key = "slots%d_reg%d" % (n_seq, regpacks)
            res[key] = dict(tok_ms=round(dtm * 1e3 / ns[0], 4), per_story_tok_ms=round(dtm * 1e3 / ns[0] / n_seq, 4),
                            profile=eng.profile_decode(4))
            for b in range(n_seq):
                eng.select(b).set_lengths(S + 3 * b, S + 3 * b)
            print(key, res[key], flush=True)
        _lib.set_tuning("gemv_x_reg_packs", 16)
        del eng
        torch.cuda.empty_cache()
    OUT["slots"] = res


def bench_nsplit():
    """Decode attention split count at 4 story slots (graph replays)."""
    res = {}
    for ns in (4, 8, 16, 32):
        _lib.set_tuning("attn_decode_nsplit", ns)
        eng = make_7b_engine(n_seq=4)
        S = 400
        for b in range(4):
            eng.select(b).prefill(torch.randn(S, 4096, device=DEV, dtype=torch.bfloat16) * 0.02)
        forced = [torch.randint(3, 32000, (115,)).tolist() for _ in range(4)]
        eng.generate_batch(8, [5] * 4, [f[:8] for f in forced])
        for b in range(4):
            eng.select(b).set_lengths(S, S)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ns_out = eng.generate_batch(115, [5] * 4, forced)
        torch.cuda.synchronize()
        res["nsplit%d_tok_ms" % ns] = round((time.perf_counter() - t0) * 1e3 / ns_out[0], 4)
        print(ns, res, flush=True)
        del eng
        torch.cuda.empty_cache()
    _lib.set_tuning("attn_decode_nsplit", 0)
    OUT["nsplit"] = res


def bench_decode():
    res = {}
    eng = make_7b_engine()
    for pipe in (1, 0):
        _lib.set_tuning("gemv_pipe", pipe)
        emb = torch.randn(343, 4096, device=DEV, dtype=torch.bfloat16) * 0.02
        eng.reset(); eng.prefill(emb)
        forced = torch.randint(3, 32000, (115,)).tolist()
        _lib.set_tuning("llama_graph", 0)
        eng.generate(115, 5, forced)
        eng.set_lengths(343, 343)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = eng.generate(115, 5, forced)
        torch.cuda.synchronize()
        res["eager_tok_ms_pipe%d" % pipe] = round((time.perf_counter() - t0) * 1e3 / n, 4)
        _lib.set_tuning("llama_graph", 1)
        print(res, flush=True)
    _lib.set_tuning("gemv_pipe", int(os.environ.get("SS_GEMV_PIPE", "1")))
    H = 4096
    for S in (343, 913):
        emb = torch.randn(S, H, device=DEV, dtype=torch.bfloat16) * 0.02
        eng.reset()
        eng.prefill(emb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.reset()
        eng.prefill(emb)
        torch.cuda.synchronize()
        res["prefill_%d_ms" % S] = round((time.perf_counter() - t0) * 1e3, 3)
        forced = torch.randint(3, 32000, (115,)).tolist()
        eng.generate(8, 5, forced[:8])  # capture + warm
        eng.set_lengths(S, S)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = eng.generate(115, 5, forced)
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t0
        res["decode_115_at_%d_ms" % S] = round(dtm * 1e3, 3)
        res["decode_tok_ms_at_%d" % S] = round(dtm * 1e3 / n, 4)
        print(S, res, flush=True)
        # continuation prefill of 65 rows against the cache
        eng.set_lengths(S, S)
        emb2 = torch.randn(65, H, device=DEV, dtype=torch.bfloat16) * 0.02
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.prefill(emb2)
        torch.cuda.synchronize()
        res["continuation_65_at_%d_ms" % S] = round((time.perf_counter() - t0) * 1e3, 3)
    eng.set_lengths(913, 913)
    res["profile"] = eng.profile_decode(4)
    p = res["profile"]
    res["gemv_GBps_in_token"] = round(p["gemv_bytes"] / p["gemv_ms"] / 1e6, 1)
    print(res, flush=True)
    OUT["decode"] = res


def bench_gemm_unet():
    """UNet GEMM / conv shapes at batch 8: tile configs x XCD-aware tile order (gemm_xcd_swizzle = group size, 0 = off).
    Operands rotate over several copies so that the Infinity Cache cannot hold them between calls (in the UNet every
    weight is touched once per forward)."""
    res = []
    shapes = [(8192, 3840, 1280), (8192, 1280, 1280), (8192, 10240, 1280), (8192, 1280, 5120), (32768, 1920, 640),
              (32768, 640, 640), (32768, 5120, 640), (32768, 640, 2560), (2048, 10240, 1280), (2048, 1280, 5120),
              (4096, 4096, 4096)]
    for M, N, K in shapes:
        ncopy = 4
        a = [torch.randn(M, K, device=DEV, dtype=torch.bfloat16) for _ in range(ncopy)]
        w = [torch.randn(N, K, device=DEV, dtype=torch.bfloat16) * 0.02 for _ in range(ncopy)]
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        row = dict(M=M, N=N, K=K)
        for cfg in (24, 26, 28, 29):
            for swz in (0, 8):
                _lib.set_tuning("gemm_cfg", cfg)
                _lib.set_tuning("gemm_xcd_swizzle", swz)
                st = {"i": 0}

                def f():
                    st["i"] += 1
                    ops.gemm(a[st["i"] % ncopy], w[st["i"] % ncopy], out=out)
                ms = timeit(f, iters=12)
                row["c%d_s%d" % (cfg, swz)] = round(2.0 * M * N * K / ms / 1e9)
        res.append(row)
        print(row, flush=True)
        del a, w
    convs = [(8, 128, 128, 320, 320), (8, 64, 64, 640, 640), (8, 32, 32, 1280, 1280), (8, 32, 32, 2560, 1280),
             (8, 64, 64, 1920, 640), (1, 1024, 1024, 128, 128)]
    for B, H, W, Ci, Co in convs:
        x = torch.randn(B * H * W, Ci, device=DEV, dtype=torch.bfloat16)
        w = torch.randn(Co, 9 * Ci, device=DEV, dtype=torch.bfloat16) * 0.02
        row = dict(conv=(B, H, W, Ci, Co))
        for cfg in (26, 28, 29):
            for swz in (0, 8):
                _lib.set_tuning("gemm_cfg", cfg)
                _lib.set_tuning("gemm_xcd_swizzle", swz)
                ms = timeit(lambda: ops.conv3x3(x, w, B, H, W), iters=5)
                row["c%d_s%d" % (cfg, swz)] = round(2.0 * B * H * W * Co * 9 * Ci / ms / 1e9)
        res.append(row)
        print(row, flush=True)
    _lib.set_tuning("gemm_cfg", 0)
    _lib.set_tuning("gemm_xcd_swizzle", 8)
    OUT["gemm_unet"] = res


def bench_sdxl():
    from seedstory.diffusion import AutoencoderKL, EulerDiscreteScheduler, StableDiffusionXLPipeline, UNet2DConditionModel
    res = {}
    dt = torch.bfloat16
    unet = UNet2DConditionModel().to(DEV, dt).init_synthetic(1)
    vae = AutoencoderKL().to(DEV, dt).init_synthetic(2)
    x = torch.randn(2, 4, 128, 128, device=DEV, dtype=dt)
    ctx = torch.randn(2, 64, 2048, device=DEV, dtype=dt)
    pooled = torch.randn(2, 1280, device=DEV, dtype=dt)
    tid = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * 2, dtype=torch.float32)
    cond = {"text_embeds": pooled, "time_ids": tid}
    for _ in range(2):
        unet(x, 500.0, ctx, added_cond_kwargs=cond)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        y = unet(x, 500.0, ctx, added_cond_kwargs=cond).sample
    torch.cuda.synchronize()
    res["unet_fwd_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
    res["unet_TFLOPs"] = round(2 * 6.747e12 / (res["unet_fwd_ms"] * 1e-3) / 1e12, 1)
    res["unet_out_finite"] = bool(torch.isfinite(y.float()).all())
    print(res, flush=True)
    lat = torch.randn(1, 4, 128, 128, device=DEV, dtype=dt)
    vae.decode_nhwc(lat)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img, H, W = vae.decode_nhwc(lat)
    torch.cuda.synchronize()
    res["vae_decode_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    print(res, flush=True)
    pipe = StableDiffusionXLPipeline(vae=vae, unet=unet, scheduler=EulerDiscreteScheduler())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pipe(prompt_embeds=ctx[:1], negative_prompt_embeds=ctx[1:], pooled_prompt_embeds=pooled[:1],
               negative_pooled_prompt_embeds=pooled[1:], num_inference_steps=30, height=1024, width=1024, output_type="pt").images
    torch.cuda.synchronize()
    res["pipeline_30_steps_s"] = round(time.perf_counter() - t0, 3)
    res["image_shape"] = list(out.shape)
    print(res, flush=True)
    OUT["sdxl"] = res


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemv", "gemm", "attn", "decode"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for w in which:
        try:
            {"gemv": bench_gemv, "gemm": bench_gemm, "attn": bench_attn, "decode": bench_decode, "sdxl": bench_sdxl, "slots": bench_slots, "nsplit": bench_nsplit, "gemm_unet": bench_gemm_unet}[w]()
        except Exception as ex:  # keep going: one broken kernel must not hide the other numbers
            import traceback
            traceback.print_exc()
            OUT[w + "_error"] = repr(ex)
        with open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w") as f:
            json.dump(OUT, f, indent=1)
