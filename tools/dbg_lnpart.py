import sys, os, math, torch
sys.path.insert(0, "/root/repo/seed-story_amd"); sys.path.insert(0, "/root/repo/oracle")
import synth
from seedstory import ops
DEV, BF = "cuda:0", torch.bfloat16
def rel(a,b): return float((a.float()-b.float()).norm()/(b.float().norm()+1e-30))
for (M,N,K) in [(32768,640,640),(8192,1280,1280),(1000,640,640)]:
    a = synth.normal_like(M + K, (M, K), 1.0).to(BF).to(DEV)
    w = synth.normal_like(N + K + 1, (N, K), 1.0 / math.sqrt(K)).to(BF).to(DEV)
    b = (synth.normal_like(7, (N,), 0.3) + 9.0).to(BF).to(DEV)
    strips = ops.rowpart_strips(M, N, K, BF)
    part = torch.full((M, strips, 2), float("nan"), dtype=torch.float32, device=DEV)
    y = ops.gemm(a, w, bias=b, rowpart=part)
    Nc = 640
    gamma = (1.0 + synth.normal_like(9, (N,), 0.1)).float()
    wc = synth.normal_like(10, (Nc, N), 1.0 / math.sqrt(N)).float()
    wg = (wc * gamma[None, :]).to(BF).to(DEV).contiguous()
    colsum = wg.float().sum(1).contiguous()
    z1 = ops.gemm_lnfold_part(y, wg, part, N, 1e-5, colsum)
    rstd, shift = ops.rowstats(y, 1e-5)
    z0 = ops.gemm_lnfold(y, wg, rstd, shift, colsum)
    e = ((z1.float()-z0.float()).norm(dim=1)/(z0.float().norm(dim=1)+1e-30)).cpu()
    bad = (e > 1e-2).nonzero().flatten()
    print(M,N,K,"strips",strips,"rel",rel(z1,z0),"bad rows",bad.numel(), bad[:40].tolist())
    yd = y.double()
    es = ((part[:, :, 0].double().sum(1) - yd.sum(1)).abs() / yd.sum(1).abs()).cpu()
    eq = ((part[:, :, 1].double().sum(1) - (yd*yd).sum(1)).abs() / (yd*yd).sum(1)).cpu()
    bp = ((es > 1e-4) | (eq > 1e-4)).nonzero().flatten()
    print("  rows with wrong PARTIAL sums:", bp.numel(), bp[:24].tolist(), " overlap with bad consumer rows:", len(set(bp.tolist()) & set(bad.tolist())))
    if bp.numel():
        r0 = int(bp[0]); print("  part row", r0, part[r0].cpu().tolist(), "true", float(yd[r0].sum()), float((yd[r0]*yd[r0]).sum()))
        ys = yd[r0].view(strips, -1)
        print("  true strips", ys.sum(1).cpu().tolist())
    yf = y.float()
    mean = yf.double().mean(1); var = yf.double().var(1, unbiased=False)
    r_ = (1.0/torch.sqrt(var+1e-5)).float(); sh_ = (-mean.float()*r_)
    zref = (r_[:,None]*(yf @ wg.float().t()) + sh_[:,None]*colsum[None,:])
    for nm, z in (("z1(part)", z1), ("z0(vec)", z0)):
        ee = ((z.float()-zref).norm(dim=1)/(zref.norm(dim=1)+1e-30)).cpu()
        bb = (ee > 1e-2).nonzero().flatten()
        print("  ", nm, "vs torch fp32: rel", rel(z, zref), "bad rows", bb.numel(), bb[:20].tolist())
        if bb.numel():
            r0 = int(bb[0])
            d = (z[r0].float() - zref[r0]).abs().cpu()
            cols = (d > 0.05 * zref[r0].abs().mean().cpu()).nonzero().flatten()
            print("     row", r0, "bad cols: n=", cols.numel(), "range", (int(cols.min()), int(cols.max())) if cols.numel() else None,
                  "z", z[r0, :4].float().cpu().tolist(), "ref", zref[r0, :4].cpu().tolist())
            ratio = (z[r0].float() / zref[r0]).cpu()
            print("     ratio z/ref median", float(ratio.median()), "first", ratio[:6].tolist())
    print("   rowstats vs fp64: rstd", rel(rstd, r_), "shift", rel(shift, sh_))
    z1b = ops.gemm_lnfold_part(y, wg, part, N, 1e-5, colsum)
    print("  consumer deterministic:", bool(torch.equal(z1b, z1)))
    from seedstory import tune
    print(" consumer cfg", tune.lookup(M, Nc, N, 1), "producer", tune.lookup(M,N,K,1))
