import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
from seedstory import ops, _lib
BF = torch.bfloat16; F8 = torch.float8_e4m3fn
DEV = "cuda:0"
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
M, N, K, geglu = 8192, 10240, 1280, True
a = synth.normal_like(11, (M, K), 1.0).to(BF).to(DEV)
w = synth.normal_like(12, (N, K), 1.0 / math.sqrt(K)).to(BF).to(DEV)
bias = synth.normal_like(13, (N,), 0.1).to(BF).to(DEV)
a8, sa = ops.quantize_rows_fp8(a)
w8, sw = ops.quantize_rows_fp8(w)
y8 = ops.gemm_fp8(a8, sa, w8, sw, bias=bias, geglu=geglu)
y16 = ops.gemm_geglu(a, w, bias)
for step in (97, 64, 1):
    rows = torch.arange(0, M, step)
    c = a[rows].double().cpu() @ w.double().cpu().T + bias.double().cpu()
    ref = c[:, 0::2] * torch.nn.functional.gelu(c[:, 1::2])
    print("step", step, "e8 %.3e e16 %.3e" % (rel(y8[rows.to(DEV)], ref), rel(y16[rows.to(DEV)], ref)), flush=True)
    if step == 97:
        d = (y8[rows.to(DEV)].double().cpu() - ref).norm(dim=1) / ref.norm(dim=1)
        print("per-row rel:", [round(float(v), 3) for v in d[:40]])
