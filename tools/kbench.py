"""Kernel micro-benchmarks on the GPU box (round 2): tile table of every GEMM / conv shape on the path (with the
per-candidate log), flash-attention generations A/B, UNet / VAE forward times.

    python tools/kbench.py tune   [--batch 8]     -> gpurun_out/tune_gfx950.json, gpurun_out/tune_log.txt (stderr)
    python tools/kbench.py attn                   -> gpurun_out/attn_ab.json
    python tools/kbench.py unet   [--batch 8]     -> gpurun_out/unet_time.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch  # noqa: E402

from seedstory import _lib, ops, tune  # noqa: E402

DEV = "cuda:0"
BF16 = _lib.SS_BF16
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)


def unet_shapes(UB):
    T32, T64, T128 = UB * 1024, UB * 4096, UB * 16384
    gemms = [(T32, 1280, 1280, 0), (T32, 3840, 1280, 0), (T32, 10240, 1280, _lib.EPI_GEGLU_PAIR), (T32, 1280, 5120, 0),
             (T64, 640, 640, 0), (T64, 1920, 640, 0), (T64, 5120, 640, _lib.EPI_GEGLU_PAIR), (T64, 640, 2560, 0),
             (UB * 64, 2560, 2048, 0), (UB * 64, 1280, 2048, 0), (T32, 1280, 2560, 0), (T32, 1280, 1920, 0),
             (T64, 640, 1920, 0), (T64, 640, 1280, 0), (T64, 640, 960, 0), (T64, 640, 320, 0), (T128, 320, 960, 0),
             (T128, 320, 640, 0), (T32, 1280, 640, 0)]
    convs = [(UB, 128, 128, 320, 320, 1, 0), (UB, 64, 64, 320, 640, 1, 0), (UB, 64, 64, 640, 640, 1, 0),
             (UB, 32, 32, 640, 1280, 1, 0), (UB, 32, 32, 1280, 1280, 1, 0), (UB, 32, 32, 2560, 1280, 1, 0),
             (UB, 32, 32, 1920, 1280, 1, 0), (UB, 64, 64, 1920, 640, 1, 0), (UB, 64, 64, 1280, 640, 1, 0),
             (UB, 64, 64, 960, 640, 1, 0), (UB, 128, 128, 960, 320, 1, 0), (UB, 128, 128, 640, 320, 1, 0),
             (UB, 128, 128, 320, 320, 2, 0), (UB, 64, 64, 640, 640, 2, 0), (UB, 32, 32, 1280, 1280, 1, 1),
             (UB, 64, 64, 640, 640, 1, 1), (UB, 128, 128, 8, 320, 1, 0), (UB, 128, 128, 320, 8, 1, 0)]
    return gemms, convs


def vae_shapes():
    gemms = [(16384, 512, 512, 0), (512 * 512, 256, 512, 0), (1024 * 1024, 128, 256, 0), (16384, 8, 8, 0)]
    convs = [(1, 128, 128, 8, 512, 1, 0), (1, 128, 128, 512, 512, 1, 0), (1, 128, 128, 512, 512, 1, 1),
             (1, 256, 256, 512, 512, 1, 0), (1, 256, 256, 512, 512, 1, 1), (1, 512, 512, 512, 256, 1, 0),
             (1, 512, 512, 256, 256, 1, 0), (1, 512, 512, 256, 256, 1, 1), (1, 1024, 1024, 256, 128, 1, 0),
             (1, 1024, 1024, 128, 128, 1, 0), (1, 1024, 1024, 128, 8, 1, 0)]
    return gemms, convs


def mllm_shapes():
    H, I = 4096, 11008
    g = []
    for M in range(128, 1025, 128):        # LLaMA prefill buckets
        g += [(M, 3 * H, H, 0), (M, H, H, 0), (M, 2 * I, H, 0), (M, H, I, 0)]
    W, MLP = 1664, 8192                    # ViT-G, one image (and the batch-2 CFG pair of the adapter)
    for rows in (1024, 2048):
        g += [(rows, W, 640, 0), (rows, 3 * W, W, 0), (rows, W, W, 0), (rows, MLP, W, _lib.EPI_GELU), (rows, W, MLP, 0),
              (rows, 4096, W, 0), (rows, 4096, 4096, 0)]
    g += [(256, 4096, 4096, 0), (512, 4096, 4096, 0), (2048, 4096, 4096, 0),      # resamplers: 64..256 queries x images
          (512, 1024, 4096, 0), (1024, 1024, 4096, 0), (640, 2048, 1024, 0), (128, 1024, 1024, 0), (256, 4096, 1024, 0)]
    return g


def cmd_tune(batches):
    _lib.set_tuning("gemm_autotune_log", 1)
    t0 = time.time()
    sets = [unet_shapes(b) for b in batches] + [vae_shapes()]
    for gemms, convs in sets:
        for (M, N, K, epi) in gemms:
            tune.ensure_gemm(M, N, K, BF16, epi, torch.device(DEV))
        for c in convs:
            tune.ensure_conv(*c[:5], c[5], bool(c[6]), BF16, torch.device(DEV))
    for (M, N, K, epi) in mllm_shapes():
        tune.ensure_gemm(M, N, K, BF16, epi, torch.device(DEV))
    n = tune.save_table(os.path.join(OUT, "tune_gfx950.json"),
                        note="measured on MI355X by tools/kbench.py tune (UNet batches %s, VAE, ViT-G, LLaMA prefill buckets)" % (batches,))
    log = [{"kind": k, "shape": list(s), "cfg": c, "swz": z, "best_us": round(us, 2)} for (k, s, c, z, us) in tune.tuned_log()]
    for r in log:
        if r["kind"] == "gemm":
            M, N, K = r["shape"]
            r["tflops"] = round(2.0 * ((M + 127) // 128 * 128) * N * K / (r["best_us"] * 1e-6) / 1e12, 1)
        else:
            B, Hh, Ww, Ci, Co, st, up = r["shape"]
            Ho = (Hh * (2 if up else 1) + 2 - 3) // st + 1
            r["tflops"] = round(2.0 * B * Ho * Ho * Co * 9 * Ci / (r["best_us"] * 1e-6) / 1e12, 1)
    json.dump(log, open(os.path.join(OUT, "tune_results.json"), "w"), indent=0)
    print("tuned %d shapes, table %d entries, %.0f s" % (len(log), n, time.time() - t0))
    for r in log:
        print(r)


def timed(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def cmd_attn():
    res = []
    dt = torch.bfloat16
    for (B, Hh, hd, L, causal) in [(16, 10, 64, 4096, False), (16, 20, 64, 1024, False), (8, 10, 64, 4096, False), (8, 20, 64, 1024, False), (2, 10, 64, 4096, False),
                                   (1, 16, 104, 1024, False), (1, 32, 128, 343, True), (1, 32, 128, 913, True)]:
        E = Hh * hd
        q = torch.randn(B, L, E, device=DEV, dtype=dt)
        k = torch.randn(B, L, E, device=DEV, dtype=dt)
        v = torch.randn(B, L, E, device=DEV, dtype=dt)
        row = {"B": B, "heads": Hh, "hd": hd, "L": L, "causal": causal}
        flops = 4.0 * B * Hh * L * L * hd * (0.5 if causal else 1.0)
        for ver, xcd in ((2, 0), (3, 0), (3, 1), (6, 1), (5, 1)):      # 5 / 6: the v3p options (head_dim 64 only; else they run v3)
            _lib.set_tuning("attn_ver", ver)
            _lib.set_tuning("attn_xcd", xcd)
            us = min(timed(lambda: ops.attention(q, k, v, Hh, None, causal), n=8) for _ in range(3))
            tag = "v%d%s" % (ver, "x" if xcd else "")
            row[tag + "_us"] = round(us, 1)
            row[tag + "_tflops"] = round(flops / (us * 1e-6) / 1e12, 1)
        _lib.set_tuning("attn_ver", 6)
        _lib.set_tuning("attn_xcd", 1)
        res.append(row)
        print(row)
    json.dump(res, open(os.path.join(OUT, "attn_ab.json"), "w"), indent=0)


def cmd_cross(UB=8):
    """cross-attention (64 context tokens) and self-attention launches of the UNet as the forward issues them."""
    import math
    dt = torch.bfloat16
    res = []
    for (heads, L) in ((20, 1024), (10, 4096)):
        E = heads * 64
        q = torch.randn(UB * L, E, device=DEV, dtype=dt)
        kv = torch.randn(UB * 64, 2 * E, device=DEV, dtype=dt)
        qkv = torch.randn(UB * L, 3 * E, device=DEV, dtype=dt)
        us_c = min(timed(lambda: ops.attention_q_kvpacked(q, kv, UB, L, 64, heads), n=20) for _ in range(3))
        _lib.set_tuning("attn_cross64", 0)          # the flash path the register-resident kernel replaced (round 5)
        us_f = min(timed(lambda: ops.attention_q_kvpacked(q, kv, UB, L, 64, heads), n=20) for _ in range(3))
        _lib.set_tuning("attn_cross64", 1)
        us_s = min(timed(lambda: ops.attention_qkv_packed(qkv, UB, L, heads), n=10) for _ in range(3))
        row = {"B": UB, "heads": heads, "L": L, "cross_us": round(us_c, 1), "cross_GBps": round(2 * q.numel() * 2 / us_c / 1e3, 1),
               "cross_flash_us": round(us_f, 1), "cross_flash_GBps": round(2 * q.numel() * 2 / us_f / 1e3, 1),
               "self_us": round(us_s, 1), "self_tflops": round(4.0 * UB * heads * L * L * 64 / us_s / 1e6, 1)}
        print(row)
        res.append(row)
    json.dump(res, open(os.path.join(OUT, "cross_attn.json"), "w"), indent=0)


def cmd_unet(UB):
    from seedstory.diffusion import UNet2DConditionModel
    dt = torch.bfloat16
    unet = UNet2DConditionModel().to(DEV, dt).init_synthetic(1)
    if os.environ.get("KB_LNFOLD") is not None:
        unet.enable_lnfold(os.environ["KB_LNFOLD"] != "0")
    x = torch.randn(UB, 4, 128, 128, device=DEV, dtype=dt)
    ctx = torch.randn(UB, 64, 2048, device=DEV, dtype=dt)
    cond = {"text_embeds": torch.randn(UB, 1280, device=DEV, dtype=dt),
            "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * UB, dtype=torch.float32)}
    unet(x, 500.0, ctx, added_cond_kwargs=cond)
    torch.cuda.synchronize()
    ops.softmax_rows_(torch.zeros(1, 8, device=DEV, dtype=dt), 1.0)   # marker kernel for tools/trace_summary.py
    us = min(timed(lambda: unet(x, 500.0, ctx, added_cond_kwargs=cond), n=3, warm=0) for _ in range(2))
    out = {"lnfold": unet._lnfold_on(), "unet_batch": UB, "forward_ms_eager": round(us / 1e3, 2), "tflops": round(UB * 6.747e12 / (us * 1e-6) / 1e12, 1)}
    print(out)
    json.dump(out, open(os.path.join(OUT, "unet_time_b%d%s.json" % (UB, "" if unet._lnfold_on() else "_nolnfold")), "w"))


def cmd_fp8(UB):
    """fp8 vs bf16: the UNet's linear shapes one by one (sustained: 20 back-to-back launches over rotating operands would
    be colder; here the operands are > L2 for the large shapes) and the whole forward."""
    import math
    from seedstory.diffusion import UNet2DConditionModel
    dt = torch.bfloat16
    res = {"gemm": []}
    T32, T64 = UB * 1024, UB * 4096
    for (M, N, K, geglu) in [(T32, 3840, 1280, False), (T32, 1280, 1280, False), (T32, 10240, 1280, True), (T32, 1280, 5120, False),
                             (T64, 1920, 640, False), (T64, 640, 640, False), (T64, 5120, 640, True), (T64, 640, 2560, False)]:
        a = torch.randn(M, K, device=DEV, dtype=dt)
        w = torch.randn(N, K, device=DEV, dtype=dt) / math.sqrt(K)
        bias = torch.randn(N, device=DEV, dtype=dt)
        a8, sa = ops.quantize_rows_fp8(a)
        w8, sw = ops.quantize_rows_fp8(w)
        row = {"M": M, "N": N, "K": K, "geglu": geglu}
        flops = 2.0 * M * N * K
        for cfg in (81, 82, 80):
            _lib.set_tuning("gemm_fp8_cfg", cfg)
            us = min(timed(lambda: ops.gemm_fp8(a8, sa, w8, sw, bias=bias, geglu=geglu), n=10) for _ in range(2))
            row["fp8_cfg%d_us" % cfg] = round(us, 1)
            row["fp8_cfg%d_tflops" % cfg] = round(flops / us / 1e6, 1)
        _lib.set_tuning("gemm_fp8_cfg", 0)
        row["fp8_us"] = round(min(timed(lambda: ops.gemm_fp8(a8, sa, w8, sw, bias=bias, geglu=geglu), n=10) for _ in range(2)), 1)
        row["quant_us"] = round(min(timed(lambda: ops.quantize_rows_fp8(a), n=10) for _ in range(2)), 1)
        f16 = (lambda: ops.gemm_geglu(a, w, bias)) if geglu else (lambda: ops.gemm(a, w, bias=bias))
        row["bf16_us"] = round(min(timed(f16, n=10) for _ in range(2)), 1)
        row["bf16_tflops"] = round(flops / row["bf16_us"] / 1e6, 1)
        print(row)
        res["gemm"].append(row)
    unet = UNet2DConditionModel().to(DEV, dt).init_synthetic(1)
    x = torch.randn(UB, 4, 128, 128, device=DEV, dtype=dt)
    ctx = torch.randn(UB, 64, 2048, device=DEV, dtype=dt)
    cond = {"text_embeds": torch.randn(UB, 1280, device=DEV, dtype=dt),
            "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * UB, dtype=torch.float32)}
    for mode in (False, True):
        unet.enable_fp8(mode)
        unet(x, 500.0, ctx, added_cond_kwargs=cond)
        torch.cuda.synchronize()
        us = min(timed(lambda: unet(x, 500.0, ctx, added_cond_kwargs=cond), n=3, warm=0) for _ in range(2))
        res["unet_forward_ms_%s" % ("fp8" if mode else "bf16")] = round(us / 1e3, 2)
    print({k: v for k, v in res.items() if k != "gemm"})
    json.dump(res, open(os.path.join(OUT, "fp8_bench_b%d.json" % UB), "w"), indent=0)


def cmd_vae():
    from seedstory.diffusion import AutoencoderKL
    dt = torch.bfloat16
    vae = AutoencoderKL().to(DEV, dt).init_synthetic(2)
    lat = torch.randn(1, 4, 128, 128, device=DEV, dtype=dt) * 0.5
    vae.decode_nhwc(lat, prescale=1.0 / 0.13025)
    usv = timed(lambda: vae.decode_nhwc(lat, prescale=1.0 / 0.13025), n=3, warm=0)
    out = {"vae_decode_ms": round(usv / 1e3, 2)}
    print(out)
    json.dump(out, open(os.path.join(OUT, "vae_time.json"), "w"))


if __name__ == "__main__":
    for kv in os.environ.get("KB_KNOBS", "").split(","):       # A/B runs: KB_KNOBS="attn_cross64=0,attn_ver=3"
        if "=" in kv:
            _lib.set_tuning(kv.split("=")[0].strip(), int(kv.split("=")[1]))
    cmd = sys.argv[1] if len(sys.argv) > 1 else "tune"
    batch = 8
    if "--batch" in sys.argv:
        batch = int(sys.argv[sys.argv.index("--batch") + 1])
    if cmd == "tune":
        cmd_tune([8, 2])
    elif cmd == "attn":
        cmd_attn()
    elif cmd == "unet":
        cmd_unet(batch)
    elif cmd == "cross":
        cmd_cross(batch)
    elif cmd == "vae":
        cmd_vae()
    elif cmd == "fp8":
        cmd_fp8(batch)
