"""TEST INFRASTRUCTURE ONLY — CPU restatement (oracle) of the SEED-Story hot path.

Nothing in the product path (``seed-story_amd/``) imports this file.  It is used by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg, and
only as the checker / the timed CPU baseline.

Every function is a *functional* restatement on a flat ``{name: tensor}`` weight dict
(HF / reference checkpoint key names) of what the reference computes, and cites the
reference lines it follows (paths relative to ``/root/reference``).  Arithmetic is
carried out by torch CPU ops **in the dtype of the tensors passed in**, in the same
operation order as the reference, so that with bf16 weights it rounds where the
reference's CPU path rounds, and with fp32 weights it is the exact fp32 reference.

Pinning: ``oracle/make_golden.py`` imports the *real* reference modules (behind the
stubs in ``oracle/ref_shims.py``), checks each restatement here against them, and
writes the golden fixtures under ``tests/golden/`` that the CPU test-suite re-checks.
Pieces whose arithmetic lives in packages that are absent here (HF-4.34 greedy
search, peft-0.4.0 LoRA, diffusers SDXL) are restated from their published behaviour
 — those are "parity unpinned at the third-party boundary".
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
W = Dict[str, Tensor]

# --------------------------------------------------------------------------------------
# LLaMA building blocks
# --------------------------------------------------------------------------------------


def rmsnorm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """LlamaRMSNorm.forward — src/models_clm/modeling_llama_xformer.py:107-115.

    variance in fp32; x * rsqrt promotes to fp32; cast to the weight dtype when that is
    half/bf16 *before* the multiply by weight (two roundings in half precision)."""
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    h = x * torch.rsqrt(var + eps)
    if weight.dtype in (torch.float16, torch.bfloat16):
        h = h.to(weight.dtype)
    return weight * h


def rope_tables(head_dim: int, max_pos: int, dtype: torch.dtype, base: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """LlamaRotaryEmbedding.__init__ — modeling_llama_xformer.py:118-134.

    fp32 tables ``cat(freqs, freqs)``; the module-level ``.to(dtype)`` in
    src/inference/gen_george.py:57 casts the (non-persistent) buffers to the model dtype."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(max_pos, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor, position_ids: Tensor) -> Tensor:
    """rotate_half / apply_rotary_pos_emb — modeling_llama_xformer.py:158-173.

    x [B,H,q,hd]; tables [max_pos,hd]; position_ids [B,q].  Arithmetic in x.dtype."""
    c = cos[position_ids].unsqueeze(1)
    s = sin[position_ids].unsqueeze(1)
    half = x.shape[-1] // 2
    rot = torch.cat((-x[..., half:], x[..., :half]), dim=-1)
    return (x * c) + (rot * s)


def attention_bottom_right_causal(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """xops.memory_efficient_attention(..., LowerTriangularFromBottomRightMask) —
    call site modeling_llama_xformer.py:289-295 (third party xformers==0.0.23.post1;
    parity unpinned: exact softmax attention, fp32 softmax, is the oracle's definition).

    q [B,H,q,hd], k/v [B,H,kv,hd] -> [B,q,H*hd]; query i attends keys j <= i + (kv-q)."""
    B, H, M, D = q.shape
    N = k.shape[2]
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) / math.sqrt(D)
    allow = torch.ones(M, N, dtype=torch.bool).tril(diagonal=N - M)
    s = s.masked_fill(~allow, float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v.float()).to(q.dtype)
    return o.transpose(1, 2).reshape(B, M, H * D)


def lora_linear(x: Tensor, w: Tensor, a: Optional[Tensor], b: Optional[Tensor], scaling: float) -> Tensor:
    """peft==0.4.0 LoRA Linear in eval mode (configs/clm_models/llama2chat7b_lora.yaml:7-26):
    ``y = x W^T + (alpha/r) * (x A^T) B^T``, dropout inert, not merged (gen_george.py:50-57)."""
    y = F.linear(x, w)
    if a is not None:
        y = y + F.linear(F.linear(x, a), b) * scaling
    return y


def lora_merge(w: Tensor, a: Tensor, b: Tensor, scaling: float) -> Tensor:
    """W' = W + (alpha/r) * B A, accumulated in fp32 and rounded once to w.dtype."""
    return (w.float() + scaling * (b.float() @ a.float())).to(w.dtype)


class LlamaDims:
    def __init__(self, hidden: int, n_heads: int, n_layers: int, inter: int, vocab: int,
                 eps: float = 1e-5, max_pos: int = 4096):
        self.hidden, self.n_heads, self.n_layers = hidden, n_heads, n_layers
        self.inter, self.vocab, self.eps, self.max_pos = inter, vocab, eps, max_pos
        self.head_dim = hidden // n_heads


def _lin(wd: W, x: Tensor, name: str, scaling: float) -> Tensor:
    a = wd.get(name + ".lora_A.weight")
    b = wd.get(name + ".lora_B.weight")
    return lora_linear(x, wd[name + ".weight"], a, b, scaling)


def llama_forward(wd: W, dims: LlamaDims, inputs_embeds: Tensor, position_ids: Tensor,
                  past: Optional[List[Tuple[Tensor, Tensor]]] = None, lora_scaling: float = 2.0,
                  all_logits: bool = True):
    """LlamaModel.forward + LlamaForCausalLM.forward — modeling_llama_xformer.py:532-666, 703-794.

    Returns (logits [B,q,V], last_hidden [B,q,H] (post final norm, :652-656), present kv list).
    Keys are cached AFTER RoPE (:236 then :239-242)."""
    B, q_len, _ = inputs_embeds.shape
    H, hd = dims.n_heads, dims.head_dim
    cos, sin = rope_tables(hd, dims.max_pos, inputs_embeds.dtype)
    h = inputs_embeds
    present = []
    for l in range(dims.n_layers):
        p = "model.layers.%d." % l
        res = h
        x = rmsnorm(h, wd[p + "input_layernorm.weight"], dims.eps)
        q = _lin(wd, x, p + "self_attn.q_proj", lora_scaling).view(B, q_len, H, hd).transpose(1, 2)
        k = _lin(wd, x, p + "self_attn.k_proj", lora_scaling).view(B, q_len, H, hd).transpose(1, 2)
        v = _lin(wd, x, p + "self_attn.v_proj", lora_scaling).view(B, q_len, H, hd).transpose(1, 2)
        q = apply_rope(q, cos, sin, position_ids)
        k = apply_rope(k, cos, sin, position_ids)
        if past is not None:
            k = torch.cat([past[l][0], k], dim=2)
            v = torch.cat([past[l][1], v], dim=2)
        present.append((k, v))
        a = attention_bottom_right_causal(q, k, v)
        h = res + _lin(wd, a, p + "self_attn.o_proj", lora_scaling)
        res = h
        x = rmsnorm(h, wd[p + "post_attention_layernorm.weight"], dims.eps)
        g = _lin(wd, x, p + "mlp.gate_proj", lora_scaling)
        u = _lin(wd, x, p + "mlp.up_proj", lora_scaling)
        h = res + _lin(wd, F.silu(g) * u, p + "mlp.down_proj", lora_scaling)
    last = rmsnorm(h, wd["model.norm.weight"], dims.eps)
    logits = F.linear(last if all_logits else last[:, -1:], wd["lm_head.weight"])
    return logits, last, present


# --------------------------------------------------------------------------------------
# Logits processor + greedy loop + ContinuousLVLM.generate
# --------------------------------------------------------------------------------------


def image_token_logits_processor(last_id: int, scores: Tensor, img_ids: Sequence[int]) -> Tensor:
    """AutoImageTokenGenerationProcessor.__call__ — src/models_clm/generation.py:19-31.

    ``img_ids`` = ids of ``<img><img_00000>..<img_000NN></img>`` (:14-17).  In place on
    ``scores`` [V] (model dtype): forced successor gets ``max+10``; otherwise the image
    tokens after ``<img>`` are *assigned 0.0* (not -inf, :29)."""
    ids = list(img_ids)
    if last_id in ids[:-1]:
        nxt = ids[ids.index(last_id) + 1]
        scores[nxt] = scores.max() + 10.0
    else:
        scores[torch.tensor(ids[1:], dtype=torch.long)] = 0.0
    return scores


def greedy_generate(wd: W, dims: LlamaDims, input_ids: Tensor, inputs_embeds: Tensor, img_ids: Sequence[int],
                    max_new_tokens: int, eos_id: int = 2, forced: Optional[Sequence[int]] = None,
                    lora_scaling: float = 2.0):
    """HF transformers==4.34.0 greedy search as driven by src/models_clm/models.py:137-153
    with ``use_kv_cache_head=False`` (gen_george.py:165) — SURVEY.md Appendix A.1.

    ``forced``: optional teacher-forced prefix of the generated tokens (the benchmark's
    forced caption schedule); the model's own argmax is used after it.
    Returns (generate_ids list, hidden rows [T-1 or T, H] for the generated inputs,
    per-step processed scores list, present kv)."""
    S = input_ids.shape[1]
    pos = torch.arange(S).unsqueeze(0)
    logits, last, kv = llama_forward(wd, dims, inputs_embeds, pos, None, lora_scaling)
    seq = input_ids[0].tolist()
    gen: List[int] = []
    hid: List[Tensor] = []
    scores_log: List[Tensor] = []
    embed = wd["model.embed_tokens.weight"]
    step_logits = logits[0, -1]
    while True:
        sc = image_token_logits_processor(seq[-1], step_logits.clone(), img_ids)
        scores_log.append(sc)
        tok = int(torch.argmax(sc).item())
        if forced is not None and len(gen) < len(forced):
            tok = int(forced[len(gen)])
        gen.append(tok)
        seq.append(tok)
        if tok == eos_id or len(gen) >= max_new_tokens:
            break
        x = embed[torch.tensor([[tok]])]
        p = torch.tensor([[len(seq) - 1]])
        logits, last, kv = llama_forward(wd, dims, x, p, kv, lora_scaling)
        hid.append(last[0, -1])
        step_logits = logits[0, -1]
    hidden = torch.stack(hid) if hid else torch.zeros(0, dims.hidden, dtype=embed.dtype)
    return gen, hidden, scores_log, kv


# ---- Resampler (qwen_visual.py) -------------------------------------------------------


def sincos_pos_embed_2d(embed_dim: int, grid: int) -> Tensor:
    """get_2d_sincos_pos_embed — src/models/qwen_visual.py:45-92 (w index varies fastest,
    first half of channels encodes grid[0] = w-coordinates mesh; [sin | cos] halves)."""
    gh = np.arange(grid, dtype=np.float32)
    gw = np.arange(grid, dtype=np.float32)
    mesh = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, -1)

    def one(d, pos):
        omega = np.arange(d // 2, dtype=np.float32)
        omega /= d / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos, omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([one(embed_dim // 2, mesh[0]), one(embed_dim // 2, mesh[1])], axis=1)
    return torch.from_numpy(emb).float()


def abs_pos_resize(pos: Tensor, tgt_len: int) -> Tensor:
    """get_abs_pos — qwen_visual.py:23-39: bicubic (align_corners=False) resize of a
    square [L,C] position table to tgt_len tokens, computed in fp32, cast back."""
    src = int(math.sqrt(pos.shape[0]))
    tgt = int(math.sqrt(tgt_len))
    if src == tgt:
        return pos
    x = pos.float().reshape(1, src, src, -1).permute(0, 3, 1, 2)
    x = F.interpolate(x, size=(tgt, tgt), mode="bicubic", align_corners=False)
    return x.permute(0, 2, 3, 1).flatten(0, 2).to(pos.dtype)


def layernorm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def mha(q_in: Tensor, k_in: Tensor, v_in: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor,
        n_heads: int) -> Tensor:
    """torch.nn.MultiheadAttention forward (batch dim second in the reference call,
    qwen_visual.py:147-149); here [B,L,E] batch-first.  in_proj in [Q;K;V] block order."""
    E = q_in.shape[-1]
    hd = E // n_heads
    q = F.linear(q_in, in_w[:E], in_b[:E])
    k = F.linear(k_in, in_w[E:2 * E], in_b[E:2 * E])
    v = F.linear(v_in, in_w[2 * E:], in_b[2 * E:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    q = q.view(B, Lq, n_heads, hd).transpose(1, 2)
    k = k.view(B, Lk, n_heads, hd).transpose(1, 2)
    v = v.view(B, Lk, n_heads, hd).transpose(1, 2)
    s = torch.matmul(q * (1.0 / math.sqrt(hd)), k.transpose(-1, -2))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v).transpose(1, 2).reshape(B, Lq, E)
    return F.linear(o, out_w, out_b)


def resampler_forward(wd: W, prefix: str, x: Tensor, n_heads: int, eps: float = 1e-5) -> Tensor:
    """Resampler.forward — qwen_visual.py:138-150.  x [N,Lkv,kv_dim] -> [N,nq,E]."""
    pos = wd[prefix + "pos_embed"]
    query = wd[prefix + "query"]
    if (prefix + "kv_proj.weight") in wd:
        x = F.linear(x, wd[prefix + "kv_proj.weight"])
    pos_kv = abs_pos_resize(pos, x.shape[1])
    x = layernorm(x, wd[prefix + "ln_kv.weight"], wd[prefix + "ln_kv.bias"], eps)
    q = layernorm(query, wd[prefix + "ln_q.weight"], wd[prefix + "ln_q.bias"], eps)
    N = x.shape[0]
    q_in = (q + pos).unsqueeze(0).expand(N, -1, -1)
    k_in = x + pos_kv.unsqueeze(0)
    return mha(q_in, k_in, x, wd[prefix + "attn.in_proj_weight"], wd[prefix + "attn.in_proj_bias"],
               wd[prefix + "attn.out_proj.weight"], wd[prefix + "attn.out_proj.bias"], n_heads)


def lvlm_generate(wd: W, dims: LlamaDims, input_ids: Tensor, image_embeds: Optional[Tensor],
                  embeds_cmp_mask: Optional[Tensor], ids_cmp_mask: Optional[Tensor], img_ids: Sequence[int],
                  max_new_tokens: int = 120, num_img_gen_tokens: int = 64, eos_id: int = 2,
                  forced: Optional[Sequence[int]] = None, n_heads_resampler: int = 32, lora_scaling: float = 2.0):
    """ContinuousLVLM.generate — src/models_clm/models.py:98-221 (no past_key_values).

    wd holds ``llm.`` -less llama keys plus ``input_resampler.*`` / ``output_resampler.*``."""
    embed = wd["model.embed_tokens.weight"]
    input_embeds = embed[input_ids].clone()
    if image_embeds is not None:
        lm = resampler_forward(wd, "input_resampler.", image_embeds, n_heads_resampler)
        input_embeds[ids_cmp_mask] = lm[embeds_cmp_mask].view(-1, dims.hidden)
    gen, hidden, scores, kv = greedy_generate(wd, dims, input_ids, input_embeds, img_ids, max_new_tokens,
                                              eos_id, forced, lora_scaling)
    eoi_id = list(img_ids)[-1]
    eoi = [i for i, t in enumerate(gen) if t == eoi_id]
    feat = None
    if eoi:
        e = eoi[-1]
        rows = hidden[e - num_img_gen_tokens:e].unsqueeze(0)
        feat = resampler_forward(wd, "output_resampler.", rows, n_heads_resampler)
    return {"generate_ids": gen, "has_img_output": bool(eoi), "img_gen_feat": feat,
            "num_gen_imgs": 1 if eoi else 0, "hidden": hidden, "scores": scores, "past_key_values": kv}


def cosine_loss(rec: Tensor, target: Tensor) -> Tensor:
    """src/models_clm/models.py:13-17."""
    target = target / target.norm(dim=-1, keepdim=True)
    rec = rec / rec.norm(dim=-1, keepdim=True)
    return (1 - (target * rec).sum(-1)).mean()


def lvlm_forward(wd: W, dims: LlamaDims, input_ids: Tensor, labels: Tensor, image_embeds: Optional[Tensor],
                 embeds_gen_mask: Optional[Tensor], embeds_cmp_mask: Optional[Tensor], ids_gen_mask: Optional[Tensor],
                 ids_cmp_mask: Optional[Tensor], n_heads_resampler: int = 32, lm_loss_scale: float = 1.0,
                 rec_loss_scale: float = 1.0, lora_scaling: float = 2.0):
    """ContinuousLVLM.forward — src/models_clm/models.py:33-96 (training-side forward, no autograd needed here).

    ``attention_mask`` is not a parameter: the reference hands it to the LLM (:64), whose xformers attention ignores it
    (modeling_llama_xformer.py:281-295, ``attn_bias=LowerTriangularMask()``; the additive mask of :274 only changes the
    discarded ``attn_weights``) — every sequence is plain causal attention over all its (padded) rows.  The token loss is
    CrossEntropyLoss over ``logits[..., :-1, :]`` / ``labels[..., 1:]`` (modeling_llama_xformer.py:761-772).  The
    placeholder branches for batches without images (:41-47, 58-62, 82-90) multiply random tensors by 0.0: exactly zero."""
    embed = wd["model.embed_tokens.weight"]
    input_embeds = embed[input_ids].clone()                               # :36
    bz, sq, dim = input_embeds.shape
    has_image = image_embeds is not None
    has_image_input = has_image and int(embeds_cmp_mask.sum()) > 0
    has_image_output = has_image and int(embeds_gen_mask.sum()) > 0
    if has_image_input:
        lm = resampler_forward(wd, "input_resampler.", image_embeds, n_heads_resampler)      # :40
        input_embeds[ids_cmp_mask] = lm[embeds_cmp_mask].view(-1, dim)                       # :55
    pos = torch.arange(sq).unsqueeze(0)
    logits, last, _ = llama_forward(wd, dims, input_embeds, pos, None, lora_scaling)         # :64-68
    shift_logits = logits[..., :-1, :].contiguous().view(-1, dims.vocab)
    shift_labels = labels[..., 1:].contiguous().view(-1)
    lm_loss = torch.nn.functional.cross_entropy(shift_logits, shift_labels)                  # CrossEntropyLoss(), ignore -100
    recon = None
    if has_image_output:
        target = image_embeds[embeds_gen_mask]                                               # :74
        n = target.shape[0]
        out_embeds = last[ids_gen_mask].view(n, -1, dim)                                     # :76
        recon = resampler_forward(wd, "output_resampler.", out_embeds, n_heads_resampler)    # :79
        rec_loss = cosine_loss(recon, target)                                                # :81
    else:
        rec_loss = torch.zeros((), dtype=input_embeds.dtype)
    total = lm_loss_scale * lm_loss + rec_loss_scale * rec_loss                              # :92
    return {"total_loss": total, "lm_loss": lm_loss, "rec_loss": rec_loss, "recon_image_embeds": recon,
            "last_hidden_state": last, "logits": logits}


# ---- Qwen ViT-G with attention pool ---------------------------------------------------


def vit_block_forward(wd: W, p: str, x: Tensor, heads: int, eps: float = 1e-6) -> Tensor:
    """VisualAttentionBlock.forward — qwen_visual.py:275-287 with VisualAttention.forward :184-235.
    x [B, L, width] (batch-first here; the reference is sequence-first, same arithmetic)."""
    B, L, width = x.shape
    hd = width // heads
    y = layernorm(x, wd[p + "ln_1.weight"], wd[p + "ln_1.bias"], eps)
    qkv = F.linear(y, wd[p + "attn.in_proj.weight"], wd[p + "attn.in_proj.bias"])
    qkv = qkv.view(B, L, heads, 3 * hd)                      # head-interleaved [q|k|v] (:192-199)
    q, k, v = qkv.split(hd, dim=-1)
    q = q.permute(0, 2, 1, 3) / math.sqrt(hd)                # scale on q (:208)
    k = k.permute(0, 2, 1, 3)
    v = v.permute(0, 2, 1, 3)
    pr = torch.matmul(q, k.transpose(-1, -2)).softmax(dim=-1)
    ctx = torch.matmul(pr, v).permute(0, 2, 1, 3).reshape(B, L, width)
    x = x + F.linear(ctx, wd[p + "attn.out_proj.weight"], wd[p + "attn.out_proj.bias"])
    y = layernorm(x, wd[p + "ln_2.weight"], wd[p + "ln_2.bias"], eps)
    y = F.gelu(F.linear(y, wd[p + "mlp.c_fc.weight"], wd[p + "mlp.c_fc.bias"]))
    return x + F.linear(y, wd[p + "mlp.c_proj.weight"], wd[p + "mlp.c_proj.bias"])


def vit_forward(wd: W, x: Tensor, *, width: int, layers: int, heads: int, patch: int, out_dim: int,
                n_queries: int = 256) -> Tensor:
    """VisionTransformerWithAttnPool.forward — qwen_visual.py:376-399 (+VisualAttention
    :184-235, VisualAttentionBlock :275-287).  LayerNorm eps 1e-6 everywhere (:353,366-372)."""
    eps = 1e-6
    B = x.shape[0]
    x = x.to(wd["conv1.weight"].dtype)
    x = F.conv2d(x, wd["conv1.weight"], None, stride=patch)
    x = x.reshape(B, width, -1).permute(0, 2, 1)
    x = x + abs_pos_resize(wd["positional_embedding"], x.shape[1])
    x = layernorm(x, wd["ln_pre.weight"], wd["ln_pre.bias"], eps)
    for i in range(layers):
        x = vit_block_forward(wd, "transformer.resblocks.%d." % i, x, heads, eps)
    x = resampler_forward(wd, "attn_pool.", x, out_dim // 128, eps)
    x = layernorm(x, wd["ln_post.weight"], wd["ln_post.bias"], eps)
    return x @ wd["proj"]


# ---- ResamplerXLV2 (de-tokenizer conditioning) ----------------------------------------


def resampler_xlv2_forward(wd: W, x: Tensor, *, depth: int, heads: int, dim_head: int) -> Tuple[Tensor, Tensor]:
    """ResamplerXLV2.forward — src/models_ipa/resampler.py:266-284 (+PerceiverAttention
    :47-76, FeedForward :10-17, AttentionPool2d :90-118).  Note F.normalize over dim=1
    (the token axis, :269) and softmax in fp32 with sqrt-scale on q and k (:69-71)."""
    B = x.shape[0]
    lat = wd["latents"].repeat(B, 1, 1)
    x = F.normalize(x)
    x = F.linear(x, wd["proj_in.weight"], wd["proj_in.bias"])
    inner = heads * dim_head

    def split(t):
        b, l, _ = t.shape
        return t.view(b, l, heads, -1).transpose(1, 2)

    for i in range(depth):
        p = "layers.%d." % i
        xn = layernorm(x, wd[p + "0.norm1.weight"], wd[p + "0.norm1.bias"], 1e-5)
        ln = layernorm(lat, wd[p + "0.norm2.weight"], wd[p + "0.norm2.bias"], 1e-5)
        q = F.linear(ln, wd[p + "0.to_q.weight"])
        kv_in = torch.cat((xn, ln), dim=-2)
        k, v = F.linear(kv_in, wd[p + "0.to_kv.weight"]).chunk(2, dim=-1)
        q, k, v = split(q), split(k), split(v)
        scale = 1 / math.sqrt(math.sqrt(dim_head))
        w_ = (q * scale) @ (k * scale).transpose(-2, -1)
        w_ = torch.softmax(w_.float(), dim=-1).type(w_.dtype)
        out = (w_ @ v).permute(0, 2, 1, 3).reshape(B, lat.shape[1], inner)
        lat = F.linear(out, wd[p + "0.to_out.weight"]) + lat
        y = layernorm(lat, wd[p + "1.0.weight"], wd[p + "1.0.bias"], 1e-5)
        y = F.linear(F.gelu(F.linear(y, wd[p + "1.1.weight"])), wd[p + "1.3.weight"])
        lat = y + lat
    hidden = layernorm(lat, wd["norm_out.weight"], wd["norm_out.bias"], 1e-5)
    e1 = F.linear(hidden, wd["unet_proj_1.weight"], wd["unet_proj_1.bias"])
    e2 = F.linear(hidden, wd["unet_proj_2.weight"], wd["unet_proj_2.bias"])
    ctx = torch.cat([e1, e2], dim=-1)
    # AttentionPool2d on [B, L, D] (resampler.py:90-118): mean token prepended, learned pos,
    # single query (token 0) multi-head attention, c_proj output.
    t = hidden.permute(1, 0, 2)
    t = torch.cat([t.mean(dim=0, keepdim=True), t], dim=0)
    t = t + wd["unet_attnpool.positional_embedding"][:, None, :].to(t.dtype)
    E = t.shape[-1]
    in_w = torch.cat([wd["unet_attnpool.q_proj.weight"], wd["unet_attnpool.k_proj.weight"],
                      wd["unet_attnpool.v_proj.weight"]])
    in_b = torch.cat([wd["unet_attnpool.q_proj.bias"], wd["unet_attnpool.k_proj.bias"],
                      wd["unet_attnpool.v_proj.bias"]])
    tb = t.permute(1, 0, 2)
    pooled = mha(tb[:, :1], tb, tb, in_w, in_b, wd["unet_attnpool.c_proj.weight"],
                 wd["unet_attnpool.c_proj.bias"], heads)
    return ctx, pooled[:, 0]


# ---- multimodal attention sink (clean spec) --------------------------------------------


def sink_evict_indices(n_kv: int, boi: int, eoi: int, sink_len: int, first: bool,
                       n_start: int = 4, boi_win=(-4, 8), eoi_win=(-8, 4)) -> Tuple[List[int], int]:
    """Index arithmetic of one eviction of the multimodal attention sink — clean spec
    distilled from src/inference/vis_george_sink.py:266-295.

    The live KV has ``n_kv`` entries = [sink (sink_len) | window]; ``boi``/``eoi`` are the
    positions of the oldest image's <img>/</img> *inside the KV* (sink offset included).
    Returns (list of KV indices to keep, in order; new sink length)."""
    keep: List[int] = []
    if first:
        keep += list(range(0, n_start))
    else:
        keep += list(range(0, sink_len))
    keep += list(range(boi + boi_win[0], boi + boi_win[1]))
    keep += list(range(eoi + eoi_win[0], eoi + eoi_win[1]))
    new_sink = len(keep)
    keep += list(range(eoi + 1, n_kv))
    return keep, new_sink
