"""fp8 GEMM tiles: per-row correctness against the dequantised operands and sustained timing over rotating weights."""
import sys, math, torch
sys.path.insert(0, "/root/repo/seed-story_amd"); sys.path.insert(0, "/root/repo/oracle")
import synth
from seedstory import ops, _lib
DEV, BF = "cuda:0", torch.bfloat16
def timed(fn, n=12):
    for _ in range(2): fn(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, geglu, res) in [(8192, 10240, 1280, True, False), (8192, 3840, 1280, False, False), (8192, 1280, 5120, False, True),
                              (8192, 1280, 1280, False, True), (32768, 5120, 640, True, False)]:
    a = synth.normal_like(M + K, (M, K), 1.0).to(BF).to(DEV)
    ws = [(synth.normal_like(N + K + 1 + i, (N, K), 1.0 / math.sqrt(K)).to(BF).to(DEV)) for i in range(4)]
    bias = synth.normal_like(5, (N,), 0.3).to(BF).to(DEV)
    r = synth.normal_like(8, (M, N), 1.0).to(BF).to(DEV) if res else None
    a8, sa = ops.quantize_rows_fp8(a)
    w8s = [ops.quantize_rows_fp8(w) for w in ws]
    af = a8.view(torch.float8_e4m3fn).float() * sa[:, None]
    wf = w8s[0][0].view(torch.float8_e4m3fn).float() * w8s[0][1][:, None]
    t = (af @ wf.t() + bias.float()).to(BF).float()
    if geglu:
        ref = (t[:, 0::2] * torch.nn.functional.gelu(t[:, 1::2]).to(BF).float())
    else:
        ref = t + (r.float() if res else 0)
    for cfg in (0, 81, 82, 95, 96):
        _lib.set_tuning("gemm_fp8_cfg", cfg)
        try:
            bad = 0
            for _ in range(3):
                z = ops.gemm_fp8(a8, sa, w8s[0][0], w8s[0][1], bias=bias, residual=r, geglu=geglu)
                e = (z.float() - ref).norm(dim=1) / (ref.norm(dim=1) + 1e-30)
                bad += int((e > 2e-2).sum())
            us = min(timed(lambda i: ops.gemm_fp8(a8, sa, w8s[i % 4][0], w8s[i % 4][1], bias=bias, residual=r, geglu=geglu)) for _ in range(3))
            print((M, N, K), "geglu" if geglu else ("res" if res else "plain"), "cfg", cfg, "rel %.2e" % float((z.float() - ref).norm() / ref.norm()),
                  "bad rows", bad, "%.1f us  %.0f TF" % (us, 2.0 * M * N * K / us * 1e-6), flush=True)
        except Exception as ex:
            print((M, N, K), "cfg", cfg, "error", str(ex)[:100])
    _lib.set_tuning("gemm_fp8_cfg", 0)
