"""Error behaviour of the boundary on a live device: every entry point reports bad arguments / exhausted capacity through
its return code + ``ss_last_error()`` (Python: ``SSError`` with that message) — never a crash, a silent clamp or a CPU
fallback — and the engine stays usable afterwards.  The reference fails at the same points with Python exceptions
(shape errors from torch, ``assert`` in models.py:131 / adapter_modules.py:395)."""
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _err(fn, *needles):
    from seedstory import _lib
    with pytest.raises(_lib.SSError) as ei:
        fn()
    msg = str(ei.value)
    assert all(n in msg for n in needles), msg
    return msg


def test_gemm_conv_attention_argument_errors():
    from seedstory import _lib, ops
    a = torch.zeros(64, 36, device=DEV, dtype=BF)        # K = 36: not a multiple of the 16-byte pack
    w = torch.zeros(32, 36, device=DEV, dtype=BF)
    _err(lambda: ops.gemm(a, w), "K/lda/ldw must be multiples of 8")
    _err(lambda: ops.gemm(a.cpu(), w), "must live on the GPU")
    _err(lambda: ops.gemm(torch.zeros(64, 64, device=DEV, dtype=BF).t(), torch.zeros(32, 64, device=DEV, dtype=BF)), "contiguous")
    x = torch.zeros(2 * 8 * 8, 12, device=DEV, dtype=BF)  # Cin = 12
    _err(lambda: ops.conv3x3(x, torch.zeros(16, 9 * 12, device=DEV, dtype=BF), 2, 8, 8), "Cin=12 must be a multiple of 8")
    q = torch.zeros(1, 16, 2 * 136, device=DEV, dtype=BF)  # head_dim 136 > 128
    _err(lambda: ops.attention(q, q, q, 2), "head_dim 136 unsupported")
    q = torch.zeros(1, 16, 128, device=DEV, dtype=BF)
    k = torch.zeros(1, 8, 128, device=DEV, dtype=BF)
    _err(lambda: ops.attention(q, k, k, 2, None, True), "causal needs kv_len >= q_len")
    a8 = torch.zeros(64, 192, device=DEV, dtype=torch.uint8)
    s = torch.ones(64, device=DEV)
    _err(lambda: ops.gemm_fp8(a8, s, a8, s), "K must be a multiple of 128")
    _err(lambda: ops.quantize_rows_fp8(torch.zeros(4, 128, device=DEV)), "16-bit inputs only")
    # the library is still healthy
    y = ops.gemm(torch.ones(64, 64, device=DEV, dtype=BF), torch.ones(32, 64, device=DEV, dtype=BF))
    assert float(y[0, 0]) == 64.0


def test_engine_capacity_errors_and_recovery(golden):
    from seedstory.llama import LlamaEngine
    g, meta = golden
    d = meta["LLAMA"]
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    eng = LlamaEngine(wd, hidden=d["hidden"], n_heads=d["n_heads"], n_layers=d["n_layers"], inter=d["inter"], vocab=d["vocab"],
                      dtype=torch.float32, device=DEV, cache_cap=48, max_new=8, max_prefill_rows=32, img_ids=list(range(254, 320)))
    emb = wd["model.embed_tokens.weight"]
    ids = synth.randint(1, (40,), 3, 250)
    _err(lambda: eng.prefill(emb[ids]), "exceeds max_prefill_rows=32")
    h0 = eng.prefill(emb[ids[:30]], want_hidden=True)
    assert eng.lengths() == (30, 30)
    _err(lambda: eng.prefill(emb[ids[:20]]), "KV cache overflow (30 + 20 > 48)")
    assert eng.lengths() == (30, 30)                                   # a rejected call changes nothing
    _err(lambda: eng.kv_gather(list(range(31))), "bad arguments")         # more indices than cached rows
    _err(lambda: eng.set_lengths(49, 0), "")
    _err(lambda: eng.select(3), "sequence slot 3 out of range")
    eng.prefill(emb[ids[:15]])                                         # 45 of 48 rows used
    _err(lambda: eng.generate(8, last_prompt_id=5), "KV cache overflow (45 + 8 > 48)")
    eng.set_lengths(30, 30)                                            # truncate back (vis_george_sink.py:243 semantics)
    n = eng.generate(4, last_prompt_id=int(ids[29]), forced=[7, 8, 9, 10])
    assert n == 4 and eng.gen_ids[:4].tolist() == [7, 8, 9, 10] and eng.lengths()[0] == 33
    eng.reset()
    h1 = eng.prefill(emb[ids[:30]], want_hidden=True)
    assert torch.equal(h0, h1)                                         # same result after the failed calls
    with pytest.raises(Exception):
        LlamaEngine(wd, hidden=d["hidden"], n_heads=d["n_heads"], n_layers=d["n_layers"], inter=d["inter"], vocab=d["vocab"],
                    dtype=torch.float32, device=DEV, cache_cap=48, max_new=8, max_prefill_rows=32, img_ids=[], n_seq=9)          # 1..8 slots


def test_prompt_may_fill_the_position_table_exactly(golden):
    """the last RoPE position a prefill uses is pos + rows - 1, so a prompt of exactly max_pos rows is legal on the
    single-slot AND the stacked path (which used to refuse it), with the same hidden rows; decoding past the table is refused."""
    from seedstory.llama import LlamaEngine
    g, meta = golden
    d = meta["LLAMA"]
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    kw = dict(hidden=d["hidden"], n_heads=d["n_heads"], n_layers=d["n_layers"], inter=d["inter"], vocab=d["vocab"],
              dtype=torch.float32, device=DEV, cache_cap=48, max_new=8, max_prefill_rows=64, img_ids=list(range(254, 320)), max_pos=24)
    emb = wd["model.embed_tokens.weight"]
    ids = synth.randint(1, (25,), 3, 250)
    one = LlamaEngine(wd, **kw)
    h1 = one.prefill(emb[ids[:24]], want_hidden=True).clone()
    assert one.lengths() == (24, 24)
    _err(lambda: one.generate(1, last_prompt_id=int(ids[23])), "position overflow")
    two = LlamaEngine(wd, n_seq=2, **kw)
    hb = two.prefill_batch([emb[ids[:24]], emb[ids[:10]]], want_hidden=True)
    assert torch.equal(hb[0], h1)
    _err(lambda: two.prefill_batch([None, emb[ids[:15]]]), "position overflow in slot 1 (10 + 15 > 24)")
    _err(lambda: one.prefill(emb[ids[:1]]), "position overflow")


def test_preprocess_and_misc_argument_errors():
    from seedstory import _lib, preprocess
    pp = preprocess.DevicePreprocessor((0.5, 0.5, 0.5), (0.5, 0.5, 0.5), 64, device=DEV, dtype=BF)
    _err(lambda: pp(torch.zeros(8, 8, 4, dtype=torch.uint8)), "uint8 [H, W, 3]")
    _err(lambda: pp(torch.zeros(8, 8, 3)), "uint8 [H, W, 3]")
    assert pp(torch.zeros(1, 1, 3, dtype=torch.uint8)).shape == (3, 64, 64)         # a 1x1 source is legal (pure replication)
    with pytest.raises(_lib.SSError):
        preprocess.resample_coeffs(0, 10, "bilinear")
    assert _lib.lib().ss_resample_ksize(10, 10, 7) == -1                             # unknown filter id
