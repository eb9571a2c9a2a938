"""TEST INFRASTRUCTURE ONLY — platform-independent synthetic tensors for parity tests.

Values are a pure function of (seed, flat index) through splitmix64 integer hashing and
a 4-term Irwin-Hall sum (only integer ops, float adds and one multiply), so the same
tensors are regenerated bit-for-bit on any box and any numpy/torch version — golden
fixtures then only need to hold *outputs*.  SURVEY.md §8(d) asks for N(0, 0.02)-style
weights (reference ``_init_weights``: src/models_clm/modeling_llama_xformer.py:399-408);
Irwin-Hall(4) scaled to the same std is the stand-in.
"""
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, n: int, stream: int = 0, start: int = 0) -> np.ndarray:
    """n float64 values in [0,1), 24 bits each (exact in fp32), for flat indices [start, start+n)."""
    with np.errstate(over="ignore"):
        idx = np.arange(start, start + n, dtype=np.uint64)
        key = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        bits = _splitmix64(idx ^ key)
    return (bits >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))


_CHUNK = 1 << 21


def _normal_chunk(seed, start, n, std, mean):
    s = np.zeros(n, dtype=np.float64)
    for k in range(4):
        s += uniform01(seed, n, stream=k + 1, start=start)
    # sum of 4 U(0,1): mean 2, var 4/12
    return ((s - 2.0) * (std / np.sqrt(4.0 / 12.0)) + mean).astype(np.float32)


def normal_like(seed: int, shape, std: float = 0.02, mean: float = 0.0, dtype=torch.float32) -> torch.Tensor:
    """Irwin-Hall(4) approximation of N(mean, std) — deterministic everywhere.  Every value is a pure function of
    (seed, flat index), so large tensors are filled chunk by chunk on a thread pool (numpy releases the GIL) with
    bit-identical results to the single-pass evaluation."""
    n = int(np.prod(shape))
    if n <= _CHUNK:
        z = _normal_chunk(seed, 0, n, std, mean)
    else:
        import os
        from concurrent.futures import ThreadPoolExecutor
        z = np.empty(n, dtype=np.float32)
        starts = list(range(0, n, _CHUNK))

        def work(st):
            m = min(_CHUNK, n - st)
            z[st:st + m] = _normal_chunk(seed, st, m, std, mean)

        with ThreadPoolExecutor(max_workers=max(1, min(32, os.cpu_count() or 1))) as ex:
            list(ex.map(work, starts))
    return torch.from_numpy(z).reshape(tuple(shape)).to(dtype)


def uniform(seed: int, shape, lo: float = 0.0, hi: float = 1.0, dtype=torch.float32) -> torch.Tensor:
    n = int(np.prod(shape))
    u = uniform01(seed, n) * (hi - lo) + lo
    return torch.from_numpy(u.astype(np.float32)).reshape(tuple(shape)).to(dtype)


def randint(seed: int, shape, lo: int, hi: int) -> torch.Tensor:
    n = int(np.prod(shape))
    u = uniform01(seed, n)
    return torch.from_numpy((np.floor(u * (hi - lo)) + lo).astype(np.int64)).reshape(tuple(shape))


# ---- synthetic weight dicts in the reference checkpoint key layouts ----


def llama_weights(seed: int, hidden: int, n_heads: int, n_layers: int, inter: int, vocab: int,
                  dtype=torch.float32, lora_r: int = 0, norm_jitter: float = 0.1):
    """HF-named LLaMA weights (+ optional LoRA A/B, both non-zero so the LoRA path is exercised)."""
    wd = {}
    s = [seed * 1000]

    def nxt():
        s[0] += 1
        return s[0]

    wd["model.embed_tokens.weight"] = normal_like(nxt(), (vocab, hidden), 0.02, dtype=dtype)
    for l in range(n_layers):
        p = "model.layers.%d." % l
        for name, (o, i) in (("self_attn.q_proj", (hidden, hidden)), ("self_attn.k_proj", (hidden, hidden)),
                             ("self_attn.v_proj", (hidden, hidden)), ("self_attn.o_proj", (hidden, hidden)),
                             ("mlp.gate_proj", (inter, hidden)), ("mlp.up_proj", (inter, hidden)),
                             ("mlp.down_proj", (hidden, inter))):
            wd[p + name + ".weight"] = normal_like(nxt(), (o, i), 0.02, dtype=dtype)
            if lora_r:
                wd[p + name + ".lora_A.weight"] = normal_like(nxt(), (lora_r, i), 0.02, dtype=dtype)
                wd[p + name + ".lora_B.weight"] = normal_like(nxt(), (o, lora_r), 0.02, dtype=dtype)
        wd[p + "input_layernorm.weight"] = normal_like(nxt(), (hidden,), norm_jitter, 1.0, dtype=dtype)
        wd[p + "post_attention_layernorm.weight"] = normal_like(nxt(), (hidden,), norm_jitter, 1.0, dtype=dtype)
    wd["model.norm.weight"] = normal_like(nxt(), (hidden,), norm_jitter, 1.0, dtype=dtype)
    wd["lm_head.weight"] = normal_like(nxt(), (vocab, hidden), 0.02, dtype=dtype)
    return wd


def resampler_weights(seed: int, prefix: str, grid: int, embed: int, kv_dim=None, dtype=torch.float32):
    """Keys of src/models/qwen_visual.py:95-127 ``Resampler`` (pos_embed is the fixed sincos table)."""
    from seedstory_oracle import sincos_pos_embed_2d
    wd = {}
    nq = grid * grid
    b = seed * 1000 + 500
    wd[prefix + "pos_embed"] = sincos_pos_embed_2d(embed, grid).to(dtype)
    wd[prefix + "query"] = normal_like(b + 1, (nq, embed), 0.02, dtype=dtype)
    if kv_dim is not None and kv_dim != embed:
        wd[prefix + "kv_proj.weight"] = normal_like(b + 2, (embed, kv_dim), 0.02, dtype=dtype)
    wd[prefix + "attn.in_proj_weight"] = normal_like(b + 3, (3 * embed, embed), 0.02, dtype=dtype)
    wd[prefix + "attn.in_proj_bias"] = normal_like(b + 4, (3 * embed,), 0.02, dtype=dtype)
    wd[prefix + "attn.out_proj.weight"] = normal_like(b + 5, (embed, embed), 0.02, dtype=dtype)
    wd[prefix + "attn.out_proj.bias"] = normal_like(b + 6, (embed,), 0.02, dtype=dtype)
    for i, ln in enumerate(("ln_q", "ln_kv")):
        wd[prefix + ln + ".weight"] = normal_like(b + 7 + 2 * i, (embed,), 0.1, 1.0, dtype=dtype)
        wd[prefix + ln + ".bias"] = normal_like(b + 8 + 2 * i, (embed,), 0.02, dtype=dtype)
    return wd


def vit_weights(seed: int, width: int, layers: int, heads: int, mlp_width: int, patch: int, out_dim: int,
                n_queries: int, dtype=torch.float32):
    """Keys of ``VisionTransformerWithAttnPool`` (qwen_visual.py:321-374)."""
    wd = {}
    s = [seed * 1000]

    def nxt():
        s[0] += 1
        return s[0]

    def ln(name):
        wd[name + ".weight"] = normal_like(nxt(), (width if "post" not in name else out_dim,), 0.1, 1.0, dtype=dtype)
        wd[name + ".bias"] = normal_like(nxt(), (width if "post" not in name else out_dim,), 0.02, dtype=dtype)

    wd["conv1.weight"] = normal_like(nxt(), (width, 3, patch, patch), 0.02, dtype=dtype)
    wd["positional_embedding"] = normal_like(nxt(), (256, width), width ** -0.5, dtype=dtype)
    ln("ln_pre")
    for i in range(layers):
        p = "transformer.resblocks.%d." % i
        ln(p + "ln_1")
        ln(p + "ln_2")
        wd[p + "attn.in_proj.weight"] = normal_like(nxt(), (3 * width, width), 0.02, dtype=dtype)
        wd[p + "attn.in_proj.bias"] = normal_like(nxt(), (3 * width,), 0.02, dtype=dtype)
        wd[p + "attn.out_proj.weight"] = normal_like(nxt(), (width, width), 0.02, dtype=dtype)
        wd[p + "attn.out_proj.bias"] = normal_like(nxt(), (width,), 0.02, dtype=dtype)
        wd[p + "mlp.c_fc.weight"] = normal_like(nxt(), (mlp_width, width), 0.02, dtype=dtype)
        wd[p + "mlp.c_fc.bias"] = normal_like(nxt(), (mlp_width,), 0.02, dtype=dtype)
        wd[p + "mlp.c_proj.weight"] = normal_like(nxt(), (width, mlp_width), 0.02, dtype=dtype)
        wd[p + "mlp.c_proj.bias"] = normal_like(nxt(), (width,), 0.02, dtype=dtype)
    grid = int(round(n_queries ** 0.5))
    wd.update(resampler_weights(seed + 77, "attn_pool.", grid, out_dim, kv_dim=width, dtype=dtype))
    ln("ln_post")
    wd["proj"] = normal_like(nxt(), (out_dim, out_dim), out_dim ** -0.5, dtype=dtype)
    return wd


def vit_block_weights(seed: int, width: int, mlp_width: int, dtype=torch.float32):
    """Keys of ONE ``VisualAttentionBlock`` (qwen_visual.py:238-287), no prefix."""
    wd = {}
    s = [seed * 1000]

    def nxt():
        s[0] += 1
        return s[0]

    for ln in ("ln_1", "ln_2"):
        wd[ln + ".weight"] = normal_like(nxt(), (width,), 0.1, 1.0, dtype=dtype)
        wd[ln + ".bias"] = normal_like(nxt(), (width,), 0.02, dtype=dtype)
    wd["attn.in_proj.weight"] = normal_like(nxt(), (3 * width, width), 0.02, dtype=dtype)
    wd["attn.in_proj.bias"] = normal_like(nxt(), (3 * width,), 0.02, dtype=dtype)
    wd["attn.out_proj.weight"] = normal_like(nxt(), (width, width), 0.02, dtype=dtype)
    wd["attn.out_proj.bias"] = normal_like(nxt(), (width,), 0.02, dtype=dtype)
    wd["mlp.c_fc.weight"] = normal_like(nxt(), (mlp_width, width), 0.02, dtype=dtype)
    wd["mlp.c_fc.bias"] = normal_like(nxt(), (mlp_width,), 0.02, dtype=dtype)
    wd["mlp.c_proj.weight"] = normal_like(nxt(), (width, mlp_width), 0.02, dtype=dtype)
    wd["mlp.c_proj.bias"] = normal_like(nxt(), (width,), 0.02, dtype=dtype)
    return wd


def resampler_xlv2_weights(seed: int, dim: int, depth: int, dim_head: int, heads: int, num_queries: int,
                           embedding_dim: int, output1_dim: int, output2_dim: int, ff_mult: int,
                           dtype=torch.float32):
    """Keys of ``ResamplerXLV2`` (src/models_ipa/resampler.py:228-264)."""
    wd = {}
    s = [seed * 1000]

    def nxt():
        s[0] += 1
        return s[0]

    def lnp(name, d):
        wd[name + ".weight"] = normal_like(nxt(), (d,), 0.1, 1.0, dtype=dtype)
        wd[name + ".bias"] = normal_like(nxt(), (d,), 0.02, dtype=dtype)

    inner = dim_head * heads
    wd["latents"] = normal_like(nxt(), (1, num_queries, dim), dim ** -0.5, dtype=dtype)
    wd["proj_in.weight"] = normal_like(nxt(), (dim, embedding_dim), 0.02, dtype=dtype)
    wd["proj_in.bias"] = normal_like(nxt(), (dim,), 0.02, dtype=dtype)
    for i in range(depth):
        p = "layers.%d." % i
        lnp(p + "0.norm1", dim)
        lnp(p + "0.norm2", dim)
        wd[p + "0.to_q.weight"] = normal_like(nxt(), (inner, dim), 0.05, dtype=dtype)
        wd[p + "0.to_kv.weight"] = normal_like(nxt(), (2 * inner, dim), 0.05, dtype=dtype)
        wd[p + "0.to_out.weight"] = normal_like(nxt(), (dim, inner), 0.05, dtype=dtype)
        lnp(p + "1.0", dim)
        wd[p + "1.1.weight"] = normal_like(nxt(), (dim * ff_mult, dim), 0.05, dtype=dtype)
        wd[p + "1.3.weight"] = normal_like(nxt(), (dim, dim * ff_mult), 0.05, dtype=dtype)
    lnp("norm_out", dim)
    wd["unet_proj_1.weight"] = normal_like(nxt(), (output1_dim, dim), 0.05, dtype=dtype)
    wd["unet_proj_1.bias"] = normal_like(nxt(), (output1_dim,), 0.02, dtype=dtype)
    wd["unet_proj_2.weight"] = normal_like(nxt(), (output2_dim, dim), 0.05, dtype=dtype)
    wd["unet_proj_2.bias"] = normal_like(nxt(), (output2_dim,), 0.02, dtype=dtype)
    wd["unet_attnpool.positional_embedding"] = normal_like(nxt(), (num_queries + 1, dim), dim ** -0.5, dtype=dtype)
    for n in ("q_proj", "k_proj", "v_proj"):
        wd["unet_attnpool.%s.weight" % n] = normal_like(nxt(), (dim, dim), 0.05, dtype=dtype)
        wd["unet_attnpool.%s.bias" % n] = normal_like(nxt(), (dim,), 0.02, dtype=dtype)
    wd["unet_attnpool.c_proj.weight"] = normal_like(nxt(), (output2_dim, dim), 0.05, dtype=dtype)
    wd["unet_attnpool.c_proj.bias"] = normal_like(nxt(), (output2_dim,), 0.02, dtype=dtype)
    return wd
