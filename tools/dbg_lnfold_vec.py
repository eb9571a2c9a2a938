import sys, os, math, torch
sys.path.insert(0, "/root/repo/seed-story_amd"); sys.path.insert(0, "/root/repo/oracle")
import synth
from seedstory import ops, _lib
DEV, BF = "cuda:0", torch.bfloat16
def rel(a,b): return float((a.float()-b.float()).norm()/(b.float().norm()+1e-30))
M,N,K = 32768,640,640
y = (synth.normal_like(3, (M, N), 1.4) + 9.0).to(BF).to(DEV)
Nc = 640
gamma = (1.0 + synth.normal_like(9, (N,), 0.1)).float()
wc = synth.normal_like(10, (Nc, N), 1.0 / math.sqrt(N)).float()
wg = (wc * gamma[None, :]).to(BF).to(DEV).contiguous()
colsum = wg.float().sum(1).contiguous()
rstd, shift = ops.rowstats(y, 1e-5)
yf = y.float()
zref = (rstd[:,None]*(yf @ wg.float().t()) + shift[:,None]*colsum[None,:])
for cfg in (0, 61, 62, 65, 67, 60):
    _lib.set_tuning("gemm_cfg", cfg)
    tot = 0
    for it in range(4):
        z0 = ops.gemm_lnfold(y, wg, rstd, shift, colsum)
        ee = ((z0.float()-zref).norm(dim=1)/(zref.norm(dim=1)+1e-30))
        tot += int((ee > 1e-2).sum())
    print("cfg", cfg, "bad rows over 4 runs:", tot, "rel", rel(z0, zref))
_lib.set_tuning("gemm_cfg", 0)
for cfg in (61,):
    _lib.set_tuning("gemm_cfg", cfg)
    tot = 0
    ref = yf @ wg.float().t()
    for it in range(4):
        z = ops.gemm(y, wg)
        ee = ((z.float()-ref).norm(dim=1)/(ref.norm(dim=1)+1e-30))
        tot += int((ee > 1e-2).sum())
    print("plain gemm cfg", cfg, "bad rows over 4 runs:", tot)
_lib.set_tuning("gemm_cfg", 0)
