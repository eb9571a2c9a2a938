"""GPU parity tests of every HIP kernel, called through the C ABI (seedstory.ops -> ctypes ->
libseedstory_hip.so) and checked against the CPU oracle (oracle/seedstory_oracle.py) or an fp32
torch-CPU evaluation of the same formula on the same (already rounded) inputs.

Tolerances (stated per test): fp32 mode is the "matches the reference CPU path" gate
(<= 1e-5 relative Frobenius unless noted); bf16 mode rounds where the reference rounds, so
element-wise kernels are bit-exact up to 1 bf16 ulp on a <=0.2 % minority of elements (fp32
reduction-order differences in the statistic), contractions are within bf16 rounding of the fp32
result."""
import math

import pytest
import torch

import seedstory_oracle as O
import synth

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (no CPU fallback exists)")
    from seedstory import ops as _ops
    return _ops


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def ulp_report(a, b):
    """bf16 tensors: fraction of elements that differ and max difference in units of bf16 ulp of b
    (ulp floored at that of 2^-6: outputs that cancel to ~0 carry the fp32 error of their O(1) terms)."""
    a, b = a.float().cpu(), b.float().cpu()
    diff = (a - b).abs()
    ulp = torch.maximum(b.abs(), torch.tensor(2.0 ** -6)) * 2.0 ** -7
    return float((diff > 0).float().mean()), float((diff / ulp).max())


def dev(t, dtype=None):
    return t.to(device=DEV, dtype=dtype if dtype is not None else t.dtype).contiguous()


@pytest.mark.parametrize("rows,cols", [(5, 256), (37, 4096), (3, 1664), (2, 8192)])
def test_rmsnorm(ops, rows, cols):
    x32 = synth.normal_like(1, (rows, cols), 1.5)
    w32 = synth.normal_like(2, (cols,), 0.1, 1.0)
    y = ops.rmsnorm(dev(x32), dev(w32), 1e-5)
    assert rel(y, O.rmsnorm(x32, w32, 1e-5)) < 1e-6
    xb, wb = x32.bfloat16(), w32.bfloat16()
    yb = ops.rmsnorm(dev(xb), dev(wb), 1e-5)
    frac, mx = ulp_report(yb, O.rmsnorm(xb, wb, 1e-5))
    assert frac < 2e-3 and mx <= 1.01, (frac, mx)


@pytest.mark.parametrize("rows,cols,eps", [(7, 256, 1e-5), (64, 4096, 1e-5), (33, 1664, 1e-6), (5, 1024, 1e-5),
                                           (1027, 640, 1e-5), (513, 1280, 1e-5), (300, 2048, 1e-5)])
def test_layernorm(ops, rows, cols, eps):
    x32 = synth.normal_like(3, (rows, cols), 2.0, 0.3)
    w32 = synth.normal_like(4, (cols,), 0.1, 1.0)
    b32 = synth.normal_like(5, (cols,), 0.1)
    y = ops.layernorm(dev(x32), dev(w32), dev(b32), eps)
    assert rel(y, O.layernorm(x32, w32, b32, eps)) < 1e-6
    xb, wb, bb = x32.bfloat16(), w32.bfloat16(), b32.bfloat16()
    yb = ops.layernorm(dev(xb), dev(wb), dev(bb), eps)
    frac, mx = ulp_report(yb, O.layernorm(xb, wb, bb, eps))
    assert frac < 5e-3 and mx <= 1.01, (frac, mx)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rope_kv_append(ops, dtype):
    H, hd, M, cap = 3, 128, 6, 32
    E = H * hd
    qkv = synth.normal_like(6, (M, 3 * E), 1.0, dtype=dtype)
    pos = torch.tensor([3, 9, 10, 40, 63, 4000])
    cos, sin = O.rope_tables(hd, 4096, dtype)
    kc = torch.zeros(H, cap, hd, dtype=dtype, device=DEV)
    vc = torch.zeros(H, cap, hd, dtype=dtype, device=DEV)
    q = ops.rope_kv_append(dev(qkv), kc, vc, dev(cos), dev(sin), H, kv_start=5, pos_ids=pos)
    qr = qkv[:, :E].view(1, M, H, hd).transpose(1, 2)
    kr = qkv[:, E:2 * E].view(1, M, H, hd).transpose(1, 2)
    vr = qkv[:, 2 * E:].view(M, H, hd).transpose(0, 1)
    q_ref = O.apply_rope(qr, cos, sin, pos.unsqueeze(0))[0].transpose(0, 1).reshape(M, E)
    k_ref = O.apply_rope(kr, cos, sin, pos.unsqueeze(0))[0]
    assert torch.equal(q.cpu(), q_ref), "q rope must be bit-exact (same roundings as the reference)"
    assert torch.equal(kc[:, 5:5 + M].cpu(), k_ref)
    assert torch.equal(vc[:, 5:5 + M].cpu(), vr)
    assert float(kc[:, :5].abs().sum()) == 0 and float(kc[:, 5 + M:].abs().sum()) == 0
    # contiguous positions path (pos_start)
    q2 = ops.rope_kv_append(dev(qkv), kc, vc, dev(cos), dev(sin), H, kv_start=0, pos_start=7)
    q2_ref = O.apply_rope(qr, cos, sin, torch.arange(7, 7 + M).unsqueeze(0))[0].transpose(0, 1).reshape(M, E)
    assert torch.equal(q2.cpu(), q2_ref)


GEMV_SHAPES = [(512, 256), (100, 512), (4096, 4096), (4096, 11008), (1000, 1664), (33, 8)]


@pytest.mark.parametrize("N,K", GEMV_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemv_plain_and_epilogues(ops, N, K, dtype):
    w = synth.normal_like(7, (N, K), 0.05, dtype=dtype)
    x = synth.normal_like(8, (K,), 1.0, dtype=dtype)
    res = synth.normal_like(9, (N,), 1.0, dtype=dtype)
    bias = synth.normal_like(10, (N,), 0.5, dtype=dtype)
    tol = 1e-5 if dtype == torch.float32 else 4e-3
    ref = w.float() @ x.float()
    y = ops.gemv(dev(w), dev(x))
    assert rel(y, ref.to(dtype)) < tol
    y = ops.gemv(dev(w), dev(x), bias=dev(bias), residual=dev(res))
    ref2 = ((ref + bias.float()).to(dtype) + res).to(dtype)
    assert rel(y, ref2) < tol
    # fused RMSNorm prologue
    nw = synth.normal_like(11, (K,), 0.1, 1.0, dtype=dtype)
    xn = O.rmsnorm(x, nw, 1e-5)
    y = ops.gemv(dev(w), dev(x), norm_w=dev(nw), eps=1e-5)
    assert rel(y, (w.float() @ xn.float()).to(dtype)) < tol


@pytest.mark.parametrize("I,K", [(256, 256), (512, 4096), (11008, 4096)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemv_silu_mul(ops, I, K, dtype):
    w = synth.normal_like(12, (2 * I, K), 0.05, dtype=dtype)
    x = synth.normal_like(13, (K,), 1.0, dtype=dtype)
    gu = (w.float() @ x.float()).to(dtype)
    ref = torch.nn.functional.silu(gu[:I]) * gu[I:]
    y = ops.gemv(dev(w), dev(x), silu_mul=True)
    assert rel(y, ref) < (1e-5 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize("nb", [2, 3, 4, 5, 8, 16])
@pytest.mark.parametrize("N,K", [(512, 256), (4096, 4096), (4096, 11008), (1000, 1664)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemv_batched_rows_match_single(ops, nb, N, K, dtype):
    """Lock-step decode of nb story slots: row b of the batched sweep == the batch-1 kernel on row b
    (same per-lane summation order in the register path; LDS-staged path within rounding).  nb >= 3 in a 16-bit type with
    K <= 4096 or K = 11008 (nb <= 8) is the MFMA form (K = 4096: the predicate-free stream loop; K = 256 / 1664 and N = 1000: predicated loads,
    ragged last row tile; K = 11008: the packed 43-step form up to 8 sequences); fp32, and K = 11008 at 16 sequences, sweep
    the weights once per half of the sequences."""
    w = dev(synth.normal_like(70, (N, K), 0.05, dtype=dtype))
    x = dev(synth.normal_like(71, (nb, K), 1.0, dtype=dtype))
    res = dev(synth.normal_like(72, (nb, N), 1.0, dtype=dtype))
    nw = dev(synth.normal_like(73, (K,), 0.1, 1.0, dtype=dtype))
    tol = 1e-5 if dtype == torch.float32 else 4e-3
    bias = dev(synth.normal_like(76, (N,), 0.5, dtype=dtype))
    for kw in ({}, {"residual": True}, {"norm": True}, {"norm": True, "bias": True, "residual": True}):
        r = res if kw.get("residual") else None
        n = nw if kw.get("norm") else None
        bs = bias if kw.get("bias") else None
        yb = ops.gemv_batched(w, x, norm_w=n, eps=1e-5, bias=bs, residual=r)
        for b in range(nb if nb <= 4 else 2):
            y1 = ops.gemv(w, x[b].contiguous(), norm_w=n, eps=1e-5, bias=bs, residual=None if r is None else r[b].contiguous())
            assert rel(yb[b], y1) < tol, (kw, b)
        xr = x.float().cpu()
        if n is not None:
            xr = torch.stack([O.rmsnorm(x[b].cpu(), nw.cpu(), 1e-5) for b in range(nb)]).float()
        ref = xr @ w.float().cpu().t()
        if bs is not None:
            ref = ref + bias.float().cpu()
        ref = ref.to(dtype)
        if r is not None:
            ref = (ref + res.cpu()).to(dtype)
        assert rel(yb, ref) < tol, kw
        for b in range(nb):
            assert rel(yb[b], ref[b]) < tol, (kw, b)


@pytest.mark.parametrize("nb", [2, 4, 6, 8])
@pytest.mark.parametrize("I,K,dtype", [(11008, 4096, torch.bfloat16), (100, 512, torch.bfloat16), (11008, 4096, torch.float16)])
def test_gemv_batched_silu(ops, nb, I, K, dtype):
    w = dev(synth.normal_like(74, (2 * I, K), 0.02 if dtype == torch.float16 else 0.05, dtype=dtype))
    x = dev(synth.normal_like(75, (nb, K), 1.0, dtype=dtype))
    yb = ops.gemv_batched(w, x, silu_mul=True)
    gu = (x.float().cpu() @ w.float().cpu().t()).to(dtype)
    ref = torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]
    for b in range(nb):
        y1 = ops.gemv(w, x[b].contiguous(), silu_mul=True)
        assert rel(yb[b], y1) < 6e-3 and rel(yb[b], ref[b]) < 6e-3, b


@pytest.mark.parametrize("nb", [3, 5, 8])
@pytest.mark.parametrize("N,K", [(4096, 4096), (4096, 11008), (32066, 4096), (1000, 2816), (100, 256), (37, 4104)])
def test_gemv_f32_split_gate_mode(ops, nb, N, K):
    """fp32 decode projections for 3..8 lock-step sequences in the gate mode (`gemm_f32_split`): ONE sweep of the fp32 weights
    through the split-bf16 MFMA form (csrc/ss_gemv.hip gemv_split_f32_kernel) instead of two 4-sequence sweeps of the exact
    dot-product kernels.  Against the fp64 product <= 3e-5 (exact kernels <= 3e-6); RMSNorm prologue, bias, residual, ragged N,
    K slices (11008 = 4096 + 4096 + 2816, 4104 = 4096 + 8) and the SiLU pair."""
    from seedstory import _lib
    w = dev(synth.normal_like(74, (N, K), 0.05))
    x = dev(synth.normal_like(75, (nb, K), 1.0))
    res = dev(synth.normal_like(76, (nb, N), 1.0))
    bias = dev(synth.normal_like(73, (N,), 0.5))
    nw = dev(synth.normal_like(72, (K,), 0.1, 1.0))
    cases = [{}, {"residual": True}, {"bias": True, "residual": True}]
    if K <= 4096:
        cases += [{"norm": True}, {"norm": True, "bias": True, "residual": True}]
    for kw in cases:
        r = res if kw.get("residual") else None
        n = nw if kw.get("norm") else None
        bs = bias if kw.get("bias") else None
        exact = ops.gemv_batched(w, x, norm_w=n, eps=1e-5, bias=bs, residual=r)
        _lib.set_tuning("gemm_f32_split", 1)
        try:
            y = ops.gemv_batched(w, x, norm_w=n, eps=1e-5, bias=bs, residual=r)
        finally:
            _lib.set_tuning("gemm_f32_split", 0)
        xr = x.double().cpu()
        if n is not None:
            xr = xr * torch.rsqrt((xr * xr).mean(dim=1, keepdim=True) + 1e-5) * nw.double().cpu()
        ref = xr @ w.double().cpu().t()
        if bs is not None:
            ref = ref + bias.double().cpu()
        if r is not None:
            ref = ref + res.double().cpu()
        e_split = float((y.double().cpu() - ref).norm() / ref.norm())
        e_exact = float((exact.double().cpu() - ref).norm() / ref.norm())
        assert e_exact < 3e-6 and e_split < 3e-5 and not torch.equal(y, exact), (kw, e_split, e_exact)
        for b in range(nb):
            assert float((y[b].double().cpu() - ref[b]).norm() / ref[b].norm()) < 3e-5, (kw, b)
    if K <= 4096 and N % 2 == 0:
        I = N // 2
        _lib.set_tuning("gemm_f32_split", 1)
        try:
            ys = ops.gemv_batched(w, x, norm_w=nw, eps=1e-5, silu_mul=True)
        finally:
            _lib.set_tuning("gemm_f32_split", 0)
        xr = x.double().cpu()
        xr = xr * torch.rsqrt((xr * xr).mean(dim=1, keepdim=True) + 1e-5) * nw.double().cpu()
        gu = xr @ w.double().cpu().t()
        refs = torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]
        assert float((ys.double().cpu() - refs).norm() / refs.norm()) < 5e-5


@pytest.mark.parametrize("nb", [1, 3, 4])
def test_gemv_mfma_form_at_small_batches(ops, nb):
    """The MFMA form is selected from 3 sequences up; by knob it also serves 1 - 4 (same results within rounding), and the
    predicated-load variant of the kernel gives bit-identical results to the predicate-free one at K = 4096 (ragged N)."""
    from seedstory import _lib
    dtype, N, K = torch.bfloat16, 4096 + 40, 4096
    w = dev(synth.normal_like(77, (N, K), 0.05, dtype=dtype))
    x = dev(synth.normal_like(78, (nb, K), 1.0, dtype=dtype))
    res = dev(synth.normal_like(79, (nb, N), 1.0, dtype=dtype))
    nw = dev(synth.normal_like(80, (K,), 0.1, 1.0, dtype=dtype))
    want = ops.gemv_batched(w, x, norm_w=nw, eps=1e-5, residual=res)
    _lib.set_tuning("gemv_mfma_min_nb", 1)
    try:
        got = ops.gemv_batched(w, x, norm_w=nw, eps=1e-5, residual=res)
        _lib.set_tuning("gemv_mfma_generic", 1)
        got_generic = ops.gemv_batched(w, x, norm_w=nw, eps=1e-5, residual=res)
    finally:
        _lib.set_tuning("gemv_mfma_min_nb", 3)
        _lib.set_tuning("gemv_mfma_generic", 0)
    assert rel(got, want) < 4e-3 and torch.equal(got, got_generic)


GEMM_SHAPES = [(1, 64, 64), (37, 100, 256), (65, 4096, 4096), (114, 1000, 4096), (343, 768, 512),
               (130, 4992, 1664), (256, 1664, 608), (300, 256, 8192), (1024, 512, 1664)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm(ops, M, N, K, dtype):
    a = synth.normal_like(14, (M, K), 1.0, dtype=dtype)
    w = synth.normal_like(15, (N, K), 0.05, dtype=dtype)
    bias = synth.normal_like(16, (N,), 0.5, dtype=dtype)
    res = synth.normal_like(17, (M, N), 1.0, dtype=dtype)
    tol = 2e-5 if dtype == torch.float32 else 4e-3
    ref = a.float() @ w.float().t()
    y = ops.gemm(dev(a), dev(w))
    assert rel(y, ref) < tol, "plain"
    y = ops.gemm(dev(a), dev(w), bias=dev(bias), residual=dev(res))
    assert rel(y, ((ref + bias.float()).to(dtype) + res).float()) < tol, "bias+residual"
    y = ops.gemm(dev(a), dev(w), bias=dev(bias), gelu=True)
    assert rel(y, torch.nn.functional.gelu((ref + bias.float()).to(dtype))) < tol * 1.5, "bias+gelu"


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES + [(528, 4096, 4096), (913, 12288, 4096), (2304, 8192, 1024), (2100, 8200, 520)])
def test_gemm_f32_split_bf16_gate_mode(ops, M, N, K):
    """`gemm_f32_split`: fp32 tensors through three bf16 MFMA products on (hi, lo) operand halves with fp32 accumulation
    (csrc/ss_gemm.hip SPLIT).  Against the fp64 product: <= 3e-5 relative (the exact fp32 chain of the same launch: <= 2e-6),
    i.e. two orders inside the 1e-3 gate it exists for; ragged M / N / K edges and the epilogues behave like the exact kernel
    (the last two shapes take the 256x128 eight-wave tile the dispatch picks for M >= 2048, N >= 8192)."""
    from seedstory import _lib
    a = synth.normal_like(14, (M, K), 1.0)
    w = synth.normal_like(15, (N, K), 0.05)
    bias = synth.normal_like(16, (N,), 0.5)
    res = synth.normal_like(17, (M, N), 1.0)
    ref = (a.double() @ w.double().t())
    exact = ops.gemm(dev(a), dev(w))
    _lib.set_tuning("gemm_f32_split", 1)
    try:
        y = ops.gemm(dev(a), dev(w))
        yb = ops.gemm(dev(a), dev(w), bias=dev(bias), residual=dev(res))
        yg = ops.gemm(dev(a), dev(w), bias=dev(bias), gelu=True)
    finally:
        _lib.set_tuning("gemm_f32_split", 0)
    e_split, e_exact = rel(y.double().cpu(), ref), rel(exact.double().cpu(), ref)
    print("gemm [%d, %d, %d] fp32: split-bf16 %.2e vs exact-fp32 %.2e from the fp64 product" % (M, N, K, e_split, e_exact))
    assert e_exact < 2e-6 and e_split < 3e-5 and not torch.equal(y, exact)
    assert rel(yb.double().cpu(), ref + bias.double() + res.double()) < 3e-5
    assert rel(yg.double().cpu(), torch.nn.functional.gelu(ref + bias.double())) < 5e-5


@pytest.mark.parametrize("cfg", [1, 2, 3])
def test_gemm_every_tile_config(ops, cfg):
    from seedstory import _lib
    a = synth.normal_like(18, (200, 512), 1.0, dtype=torch.bfloat16)
    w = synth.normal_like(19, (300, 512), 0.05, dtype=torch.bfloat16)
    _lib.set_tuning("gemm_cfg", cfg)
    try:
        y = ops.gemm(dev(a), dev(w))
    finally:
        _lib.set_tuning("gemm_cfg", 0)
    assert rel(y, a.float() @ w.float().t()) < 4e-3


@pytest.mark.parametrize("cfg", [8, 10, 15, 20, 21, 22, 23, 24, 26, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 54, 55, 56, 57, 58,
                                 60, 61, 62, 63, 64, 65, 66, 67, 68, 69, 70, 71, 72])
@pytest.mark.parametrize("M,N,K", [(200, 300, 512), (1024, 640, 1280), (333, 1000, 64), (4096, 256, 2560), (130, 72, 192)])
def test_gemm_dma_tile_configs_with_epilogues(ops, cfg, M, N, K):
    """Every LDS-DMA tile configuration (8/10/15 double-buffered, 20-46 / 60-72 software-pipelined, 54-58 ping-pong) on ragged and
    tile-aligned shapes, with the bias / residual / GELU / GEGLU-pair epilogues.  (1024, 640, 1280) is a whole-tile shape of the
    256x320 ping-pong tiles 56 / 58; on the ragged shapes those two answer "not eligible" and the call takes their fallback tile.)"""
    from seedstory import _lib
    dtype = torch.bfloat16
    a = dev(synth.normal_like(180, (M, K), 1.0, dtype=dtype))
    w = dev(synth.normal_like(181, (N, K), 0.05, dtype=dtype))
    bias = dev(synth.normal_like(182, (N,), 0.5, dtype=dtype))
    res = dev(synth.normal_like(183, (M, N), 1.0, dtype=dtype))
    ref = a.float() @ w.float().t()
    _lib.set_tuning("gemm_cfg", cfg)
    try:
        y0 = ops.gemm(a, w)
        y1 = ops.gemm(a, w, bias=bias, residual=res)
        y2 = ops.gemm(a, w, bias=bias, gelu=True)
        y3 = ops.gemm_geglu(a, w, bias) if N % 2 == 0 else None
    finally:
        _lib.set_tuning("gemm_cfg", 0)
    assert rel(y0, ref) < 4e-3
    assert rel(y1, ((ref + bias.float()).to(dtype) + res).float()) < 4e-3
    assert rel(y2, torch.nn.functional.gelu((ref + bias.float()).to(dtype))) < 6e-3
    if y3 is not None:
        full = (ref + bias.float()).to(dtype).float()
        assert rel(y3, full[:, 0::2] * torch.nn.functional.gelu(full[:, 1::2].to(dtype)).float()) < 8e-3


@pytest.mark.parametrize("cfg", [54, 55, 56, 57, 58])
@pytest.mark.parametrize("K", [64, 128, 192, 256, 320, 448])
@pytest.mark.parametrize("M,N", [(512, 640), (520, 330), (256, 320)])
def test_gemm_pingpong_k_tile_edge_cases(ops, cfg, K, M, N):
    """The ping-pong tiles (ss_gemm_pp.inc) on 1 .. 7 K tiles: prologue, the odd last tile, the tail forms of the counted vmcnt waits
    (4-phase schedule 54 / 55 / 56, two-super-phase schedule 57 / 58), whole-tile and ragged shapes; repeated launches must be
    bit-identical (a mis-counted DMA wait shows up as run-to-run drift) and equal the one-barrier tile 60 bit for bit (same k order)."""
    from seedstory import _lib
    dtype = torch.bfloat16
    a = dev(synth.normal_like(192, (M, K), 1.0, dtype=dtype))
    w = dev(synth.normal_like(193, (N, K), 0.05, dtype=dtype))
    bias = dev(synth.normal_like(194, (N,), 0.5, dtype=dtype))
    res = dev(synth.normal_like(195, (M, N), 1.0, dtype=dtype))
    try:
        _lib.set_tuning("gemm_cfg", 60)
        y60 = ops.gemm(a, w, bias=bias, residual=res)
        _lib.set_tuning("gemm_cfg", cfg)
        ys = [ops.gemm(a, w, bias=bias, residual=res) for _ in range(6)]
    finally:
        _lib.set_tuning("gemm_cfg", 0)
    assert rel(ys[0], ((a.float() @ w.float().t() + bias.float()).to(dtype) + res).float()) < 4e-3
    for y in ys:
        assert torch.equal(y, y60)


@pytest.mark.parametrize("cfg", [54, 55, 56, 57, 58])
@pytest.mark.parametrize("M,N,K", [(512, 640, 256), (1000, 512, 320), (256, 320, 64), (2048, 1280, 128)])
def test_gemm_pingpong_pipelined_epilogue_variants(ops, cfg, M, N, K):
    """The chunk-pipelined staged epilogue of the ping-pong tiles (gemm_epilogue_staged_pipe: plain, bias, residual with its
    two-chunk prefetch, bias + residual, GELU, GEGLU pair) against the one-barrier tile 60, which keeps the chunk-serial
    gemm_epilogue_staged: bit for bit, on whole-tile shapes and on a ragged M (row predicates of the 256-wide tiles; the 320-wide
    tiles refuse ragged shapes and the call takes their fallback), 1 .. 5 K tiles, repeated launches identical."""
    from seedstory import _lib
    dtype = torch.bfloat16
    a = dev(synth.normal_like(196, (M, K), 1.0, dtype=dtype))
    w = dev(synth.normal_like(197, (N, K), 0.05, dtype=dtype))
    bias = dev(synth.normal_like(198, (N,), 0.5, dtype=dtype))
    res = dev(synth.normal_like(199, (M, N), 1.0, dtype=dtype))

    def run():
        return [ops.gemm(a, w), ops.gemm(a, w, bias=bias), ops.gemm(a, w, residual=res), ops.gemm(a, w, bias=bias, residual=res),
                ops.gemm(a, w, bias=bias, gelu=True), ops.gemm_geglu(a, w, bias)]
    try:
        _lib.set_tuning("gemm_cfg", 60)
        ref = run()
        _lib.set_tuning("gemm_cfg", cfg)
        outs = [run() for _ in range(3)]
    finally:
        _lib.set_tuning("gemm_cfg", 0)
    assert rel(ref[3], ((a.float() @ w.float().t() + bias.float()).to(dtype) + res).float()) < 4e-3
    for o in outs:
        for y, r in zip(o, ref):
            assert torch.equal(y, r)


@pytest.mark.parametrize("cfg", [30, 31, 32, 33, 35, 36, 38, 39, 40, 41, 42, 43, 60, 61, 62, 64])
@pytest.mark.parametrize("K", [64, 128, 192, 256, 320])
def test_gemm_ring_depth_k_tile_edge_cases(ops, cfg, K):
    """1 .. 5 K tiles through the 2- and 3-slot LDS rings (prologue / drain paths of the counted vmcnt waits) and the
    uneven DMA deal of the 256x160 tile."""
    from seedstory import _lib
    dtype = torch.bfloat16
    M, N = 520, 330
    a = dev(synth.normal_like(190, (M, K), 1.0, dtype=dtype))
    w = dev(synth.normal_like(191, (N, K), 0.05, dtype=dtype))
    _lib.set_tuning("gemm_cfg", cfg)
    try:
        ys = [ops.gemm(a, w) for _ in range(4)]
    finally:
        _lib.set_tuning("gemm_cfg", 0)
    assert rel(ys[0], a.float() @ w.float().t()) < 4e-3
    for y in ys[1:]:
        assert torch.equal(y, ys[0])


@pytest.mark.parametrize("cfg", [33, 35, 36, 38, 39, 40, 43, 54, 55, 56, 57, 58, 60, 64, 72])
@pytest.mark.parametrize("M,N,K", [(8200, 3840, 192), (8192, 10240, 128), (8192, 5120, 64), (16384, 2560, 640)])
def test_gemm_persistent_multi_tile(ops, cfg, M, N, K):
    """Persistent configurations with several output tiles per workgroup (grid capped at the CU count): the next tile's
    first K tiles are DMA'd under the epilogue of the current one.  Repeated launches must be bit-identical (a
    staging race shows up as run-to-run drift), results equal fp32 math on the bf16 inputs."""
    from seedstory import _lib
    dtype = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(M + N + K + cfg)
    a = torch.randn(M, K, device="cuda", dtype=dtype, generator=g)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(dtype)
    bias = torch.randn(N, device="cuda", dtype=dtype, generator=g)
    res = torch.randn(M, N, device="cuda", dtype=dtype, generator=g)
    ref = ((a.float() @ w.float().t() + bias.float()).to(dtype) + res).float()
    _lib.set_tuning("gemm_cfg", cfg)
    try:
        ys = [ops.gemm(a, w, bias=bias, residual=res) for _ in range(4)]
        yg = ops.gemm_geglu(a, w, bias)
    finally:
        _lib.set_tuning("gemm_cfg", 0)
    assert rel(ys[0], ref) < 4e-3
    assert float((ys[0].float() - ref).abs().max()) < 0.08 * float(ref.abs().max())
    for y in ys[1:]:
        assert torch.equal(y, ys[0])
    full = (a.float() @ w.float().t() + bias.float()).to(dtype).float()
    assert rel(yg, full[:, 0::2] * torch.nn.functional.gelu(full[:, 1::2].to(dtype)).float()) < 8e-3


def attn_ref(q, k, v, n_heads, scale, causal_br):
    """fp32 softmax attention on [B, L, E] with packed heads."""
    B, Lq, E = q.shape
    Lk = k.shape[1]
    hd = E // n_heads
    qh = q.float().view(B, Lq, n_heads, hd).transpose(1, 2)
    kh = k.float().view(k.shape[0], Lk, n_heads, hd).transpose(1, 2)
    vh = v.float().view(v.shape[0], Lk, n_heads, hd).transpose(1, 2)
    s = torch.matmul(qh, kh.transpose(-1, -2)) * scale
    if causal_br:
        allow = torch.ones(Lq, Lk, dtype=torch.bool).tril(diagonal=Lk - Lq)
        s = s.masked_fill(~allow, float("-inf"))
    o = torch.matmul(torch.softmax(s, -1), vh)
    return o.transpose(1, 2).reshape(B, Lq, E)


ATTN_CASES = [  # B, heads, hd, Lq, Lk, causal
    (1, 2, 128, 37, 37, True), (1, 2, 128, 9, 46, True), (1, 3, 128, 343, 343, True), (1, 2, 128, 65, 400, True),
    (1, 2, 128, 1, 70, True), (2, 2, 104, 16, 16, False), (1, 16, 104, 1024, 1024, False), (3, 2, 128, 16, 64, False),
    (2, 4, 64, 8, 24, False), (2, 16, 64, 64, 320, False), (1, 2, 128, 256, 64, False)]


@pytest.mark.parametrize("B,H,hd,Lq,Lk,causal", ATTN_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_flash_attention(ops, B, H, hd, Lq, Lk, causal, dtype):
    E = H * hd
    q = synth.normal_like(20, (B, Lq, E), 1.0, dtype=dtype)
    k = synth.normal_like(21, (B, Lk, E), 1.0, dtype=dtype)
    v = synth.normal_like(22, (B, Lk, E), 1.0, dtype=dtype)
    scale = 1.0 / math.sqrt(hd)
    o = ops.attention(dev(q), dev(k), dev(v), H, scale, causal)
    ref = attn_ref(q, k, v, H, scale, causal)
    assert rel(o, ref) < (2e-5 if dtype == torch.float32 else 1e-2)


def test_flash_attention_softmax_rescale_branch(ops):
    """Online-softmax correctness when the running max jumps late: spike one key in the last tile."""
    H, hd, Lq, Lk = 1, 128, 32, 200
    q = synth.normal_like(23, (1, Lq, hd), 1.0)
    k = synth.normal_like(24, (1, Lk, hd), 1.0)
    v = synth.normal_like(25, (1, Lk, hd), 1.0)
    k[0, 190] = q[0, 5] * 3.0
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 1e-2)):
        o = ops.attention(dev(q, dtype), dev(k, dtype), dev(v, dtype), H, 1.0 / math.sqrt(hd), False)
        assert rel(o, attn_ref(q.to(dtype), k.to(dtype), v.to(dtype), H, 1.0 / math.sqrt(hd), False)) < tol


@pytest.mark.parametrize("kv_len", [1, 17, 64, 500, 1100])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attn_decode(ops, kv_len, dtype):
    H, hd, cap = 4, 128, 1200
    q = synth.normal_like(26, (H * hd,), 1.0, dtype=dtype)
    kc = synth.normal_like(27, (H, cap, hd), 1.0, dtype=dtype)
    vc = synth.normal_like(28, (H, cap, hd), 1.0, dtype=dtype)
    n = torch.tensor([kv_len], dtype=torch.int32, device=DEV)
    o = ops.attn_decode(dev(q), dev(kc), dev(vc), n)
    qh = q.float().view(H, 1, hd)
    s = torch.matmul(qh, kc[:, :kv_len].float().transpose(-1, -2)) / math.sqrt(hd)
    ref = torch.matmul(torch.softmax(s, -1), vc[:, :kv_len].float()).reshape(H * hd)
    assert rel(o, ref) < (1e-5 if dtype == torch.float32 else 4e-3)


def test_attention_cache_matches_decode(ops):
    """q_len=1 through the flash kernel == split-KV decode kernel (same cache planes)."""
    H, hd, cap, kv = 2, 128, 300, 257
    dtype = torch.bfloat16
    q = dev(synth.normal_like(29, (1, H * hd), 1.0, dtype=dtype))
    kc = dev(synth.normal_like(30, (H, cap, hd), 1.0, dtype=dtype))
    vc = dev(synth.normal_like(31, (H, cap, hd), 1.0, dtype=dtype))
    a = ops.attention_cache(q, kc, vc, kv, causal_br=True)
    b = ops.attn_decode(q[0].contiguous(), kc, vc, torch.tensor([kv], dtype=torch.int32, device=DEV))
    assert rel(a[0], b) < 1e-2


@pytest.mark.parametrize("dtype,M,hd", [(torch.bfloat16, 66, 128), (torch.bfloat16, 20, 128), (torch.bfloat16, 130, 64),
                                        (torch.float32, 66, 128), (torch.float16, 40, 128)])
def test_attention_ragged_slots_equal_per_slot_launches(ops, dtype, M, hd):
    """ss_attention_ragged (one launch, slot b attends to its own kv_lens[b] cache entries, bottom-right causal per slot)
    == one ss_attention launch per slot, bit for bit: every flash kernel variant (q_len < 32 | >= 32, head dim 64 | 128)."""
    S, H, cap = 5, 4, 320
    lens = [M, 300, M + 1, 257, max(M, 129)]
    q = dev(synth.normal_like(40, (S * M, H * hd), 1.0, dtype=dtype))
    kc = dev(synth.normal_like(41, (S, H, cap, hd), 1.0, dtype=dtype))
    vc = dev(synth.normal_like(42, (S, H, cap, hd), 1.0, dtype=dtype))
    got = ops.attention_cache_slots(q, kc, vc, lens)
    for b in range(S):
        want = ops.attention_cache(q[b * M:(b + 1) * M].contiguous(), kc[b], vc[b], lens[b], causal_br=True)
        assert torch.equal(got[b * M:(b + 1) * M], want), b
    from seedstory import _lib
    with pytest.raises(_lib.SSError):
        ops.attention_cache_slots(q, kc, vc, [M - 1] + lens[1:])          # causal: a slot needs kv_len >= q_len


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_imgproc_argmax(ops, dtype, golden):
    g, meta = golden
    lo, hi = meta["IMG_IDS"]
    ids = list(range(lo, hi + 1))
    V = meta["LLAMA"]["vocab"]
    for i, last in enumerate([17, ids[0], ids[5], ids[64], ids[65], 2]):
        sc = synth.normal_like(300 + i, (1, V), 2.0, dtype=dtype)[0]
        ref = O.image_token_logits_processor(last, sc.clone(), ids)
        d = dev(sc)
        tok = ops.imgproc_argmax(d, last, ids)
        assert torch.equal(d.cpu(), ref), "processor must edit the logits exactly like the reference"
        assert int(tok.item()) == int(torch.argmax(ref))
    # ties resolve to the first index (torch.argmax on CPU)
    t = torch.zeros(V, dtype=dtype)
    t[[40, 7, 99]] = 5.0
    assert int(ops.imgproc_argmax(dev(t), 3, ids).item()) == 7
    # a large vocabulary (LLaMA + 66 image tokens)
    big = synth.normal_like(77, (32066,), 3.0, dtype=dtype)
    ids2 = list(range(32000, 32066))
    ref = O.image_token_logits_processor(11, big.clone(), ids2)
    assert int(ops.imgproc_argmax(dev(big), 11, ids2).item()) == int(torch.argmax(ref))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_small_elementwise(ops, dtype):
    gu = synth.normal_like(32, (9, 2 * 512), 1.5, dtype=dtype)
    ref = torch.nn.functional.silu(gu[:, :512]) * gu[:, 512:]
    assert rel(ops.silu_mul(dev(gu)), ref) < (1e-6 if dtype == torch.float32 else 3e-3)
    x = synth.normal_like(33, (3, 16, 256), 1.0, dtype=dtype)
    pz = synth.normal_like(34, (16, 256), 1.0, dtype=dtype)
    assert torch.equal(ops.add_bcast(dev(x), dev(pz)).cpu(), x + pz)
    table = synth.normal_like(35, (50, 256), 1.0, dtype=dtype)
    ids = torch.tensor([3, 49, 0, 3, 17])
    assert torch.equal(ops.gather_rows(dev(table), ids).cpu(), table[ids])
    dst = dev(torch.zeros(20, 256, dtype=dtype))
    src = synth.normal_like(36, (4, 256), 1.0, dtype=dtype)
    ops.scatter_rows_(dst, torch.tensor([5, 6, 7, 19]), dev(src))
    ref = torch.zeros(20, 256, dtype=dtype)
    ref[[5, 6, 7, 19]] = src
    assert torch.equal(dst.cpu(), ref)
    img = synth.normal_like(37, (2, 3, 56, 56), 1.0, dtype=dtype)
    col = ops.im2col_patch(dev(img), 14, 640).cpu()
    ref = torch.nn.functional.unfold(img.float(), kernel_size=14, stride=14).transpose(1, 2).reshape(-1, 588)
    assert torch.equal(col[:, :588].float(), ref) and float(col[:, 588:].abs().sum()) == 0
    xl = synth.normal_like(38, (2, 16, 256), 1.0, dtype=dtype)
    assert rel(ops.l2normalize_dim1(dev(xl)), torch.nn.functional.normalize(xl.float())) < (1e-6 if dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize("M,N,K,res,bias", [(264, 12288, 4096, False, False), (264, 4096, 4096, True, False), (264, 22016, 4096, False, False),
                                             (264, 4096, 11008, True, True), (460, 12288, 4096, False, True), (130, 4096, 4096, True, False),
                                             (512, 4096, 11008, True, False), (300, 1024, 2048, False, False), (66, 4096, 4096, True, False)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_gemm_splitk_small_m_weight_streaming(M, N, K, res, bias, dtype):
    """ss_gemm_splitk (128 < M <= 512 against LLaMA-sized projections: K split over grid.y, fp32 partial sums, reduce pass
    with the epilogue) vs fp32 math on the same operands; shapes outside its range and fp32 fall back to ss_gemm."""
    from seedstory import _lib, ops
    import math
    a = synth.normal_like(M + 1, (M, K), 1.0, dtype=dtype).to(DEV)
    w = synth.normal_like(N + 2, (N, K), 1.0 / math.sqrt(K), dtype=dtype).to(DEV)
    b = synth.normal_like(3, (N,), 0.5, dtype=dtype).to(DEV) if bias else None
    r = synth.normal_like(4, (M, N), 1.0, dtype=dtype).to(DEV) if res else None
    y = ops.gemm_splitk(a, w, bias=b, residual=r)
    ref = a.float() @ w.float().t()
    if b is not None:
        ref = ref + b.float()
    if r is not None:
        ref = ref.to(dtype).float() + r.float()
    tol = 2e-5 if dtype == torch.float32 else 4e-3
    assert rel(y, ref) < tol
    nb = _lib.lib().ss_gemm_splitk_workspace_bytes(M, N, K)
    assert nb % (M * N * 4) == 0                            # S slices of fp32 partial sums (0 = not eligible: plain ss_gemm)
    if (M, N, K) in ((264, 4096, 4096), (264, 4096, 11008), (512, 4096, 11008)):
        assert nb >= 2 * M * N * 4
    if M <= 128 or (M, N, K) in ((264, 22016, 4096), (264, 12288, 4096)):
        assert nb == 0                                      # enough 128x128 workgroups without splitting
    y0 = ops.gemm(a, w, bias=b, residual=r)
    assert rel(y, y0) < tol


@pytest.mark.parametrize("B,heads,hd,Lq,Lk,causal", [(16, 10, 64, 4096, 4096, False), (16, 20, 64, 1024, 1024, False), (16, 20, 64, 1024, 64, False),
                                                    (3, 5, 64, 1000, 1000, False), (2, 3, 64, 130, 77, False), (1, 4, 64, 64, 64, False),
                                                    (2, 4, 64, 333, 333, True), (1, 8, 64, 200, 913, True), (1, 2, 64, 33, 4096, True),
                                                    (2, 4, 32, 300, 300, True), (3, 2, 40, 257, 100, False), (1, 6, 16, 64, 500, True)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_flash_v3p_equals_v3(B, heads, hd, Lq, Lk, causal, dtype):
    """The shipped head-dim <= 64 kernel (attn_ver 6 = v3p: v3 with loop-invariant DMA source addresses, csrc/ss_attn.hip (1d))
    performs the same arithmetic per score as v3, so the two must agree bit for bit on every shape class: self-attention with many
    tiles, cross-attention with one tile, ragged tails in q and kv, bottom-right causal, head dims below 64 (toy models)."""
    from seedstory import _lib, ops
    E = heads * hd
    g = torch.Generator(device=DEV).manual_seed(Lq * 7 + Lk + heads + hd)
    q = torch.randn(B, Lq, E, device=DEV, dtype=dtype, generator=g)
    k = torch.randn(B, Lk, E, device=DEV, dtype=dtype, generator=g)
    v = torch.randn(B, Lk, E, device=DEV, dtype=dtype, generator=g)
    k[:, Lk // 3] *= 6.0                     # a dominant key mid-stream: the deferred-rescale branch
    outs = {}
    for vv in (3, 6):
        _lib.set_tuning("attn_ver", vv)
        try:
            outs[vv] = ops.attention(q, k, v, heads, None, causal).clone()
        finally:
            _lib.set_tuning("attn_ver", 6)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(outs[6].float()).all())
    assert torch.equal(outs[3], outs[6]), float((outs[3].float() - outs[6].float()).abs().max())


@pytest.mark.parametrize("B,heads,Lq,Lk,causal", [(16, 10, 4096, 4096, False), (16, 20, 1024, 1024, False), (16, 20, 1024, 64, False), (3, 5, 1000, 1000, False),
                                                  (2, 3, 130, 77, False), (2, 4, 333, 333, True), (1, 8, 200, 913, True), (1, 2, 33, 4096, True),
                                                  (2, 4, 700, 700, True), (1, 4, 513, 257, False)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_flash_v3p_waves_per_workgroup_equal(B, heads, Lq, Lk, causal, dtype):
    """flash_attn3p with 4 / 8 / 16 waves per workgroup (128 / 256 / 512 query rows sharing the K / V tiles; round 6, knob attn_waves; the
    default rule picks by launch size): per-wave arithmetic is identical, only the deal of the DMA pieces and the number of waves at the
    tile barrier change — bit-equal on full and ragged query blocks, one-tile contexts, bottom-right causal."""
    from seedstory import _lib, ops
    E = heads * 64
    g = torch.Generator(device=DEV).manual_seed(Lq * 5 + Lk + heads)
    q = torch.randn(B, Lq, E, device=DEV, dtype=dtype, generator=g)
    k = torch.randn(B, Lk, E, device=DEV, dtype=dtype, generator=g)
    v = torch.randn(B, Lk, E, device=DEV, dtype=dtype, generator=g)
    k[:, Lk // 2] *= 5.0
    outs = {}
    try:
        for w in (4, 8, 16, 0):
            _lib.set_tuning("attn_waves", w)
            outs[w] = [ops.attention(q, k, v, heads, None, causal).clone() for _ in range(2)]
    finally:
        _lib.set_tuning("attn_waves", 0)
    assert bool(torch.isfinite(outs[4][0].float()).all())
    for w in (8, 16, 0):
        for y in outs[w]:
            assert torch.equal(y, outs[4][0]), (w, float((y.float() - outs[4][0].float()).abs().max()))
