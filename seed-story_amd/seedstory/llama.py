"""Python handle on the native LLaMA decoder engine (``ss_llama_*`` in the C ABI).

Owns the device memory (torch tensors) the engine works in — merged/concatenated weights, the
KV-cache slab and activation workspace — and exposes the KV cache in the reference's layout
(``past_key_values``: tuple over layers of ``(k, v)`` each ``[1, n_heads, len, head_dim]``, keys
post-RoPE) as zero-copy views.
"""
import ctypes as C

import torch

from . import _lib, ops, tune
from ._lib import check, lib

ATTN_PROJ = ("q_proj", "k_proj", "v_proj", "o_proj")
MLP_PROJ = ("gate_proj", "up_proj", "down_proj")


def rope_tables(head_dim, max_pos, dtype, base=10000.0):
    """cos/sin tables of LlamaRotaryEmbedding (modeling_llama_xformer.py:118-134): fp32
    ``cat(freqs, freqs)``, cast to the model dtype by the module-level ``.to(dtype)``."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(max_pos, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _merged(sd, name, dtype, device, lora_scaling):
    """W (+ (alpha/r) B A when LoRA factors are present), merged in fp32 on the device."""
    w = sd[name + ".weight"].to(device=device)
    a = sd.get(name + ".lora_A.weight", sd.get(name + ".lora_A.default.weight"))
    b = sd.get(name + ".lora_B.weight", sd.get(name + ".lora_B.default.weight"))
    if a is not None and b is not None:
        w = w.float() + lora_scaling * (b.to(device=device).float() @ a.to(device=device).float())
    return w.to(dtype).contiguous()


class LlamaEngine:
    def __init__(self, state_dict, *, hidden, n_heads, n_layers, inter, vocab, dtype=torch.bfloat16, device="cuda:0",
                 rms_eps=1e-5, max_pos=4096, cache_cap=2048, max_new=512, max_prefill_rows=1024, img_ids=(),
                 eos_id=2, lora_scaling=2.0, n_seq=1):
        self.n_seq = int(n_seq)
        self.device = torch.device(device)
        self.dtype = dtype
        self.hidden, self.n_heads, self.n_layers, self.inter, self.vocab = hidden, n_heads, n_layers, inter, vocab
        self.hd = hidden // n_heads
        self.cache_cap, self.max_new, self.max_rows = cache_cap, max_new, max_prefill_rows
        self.img_ids = [int(i) for i in img_ids]
        self.eos_id = eos_id
        sd = state_dict
        dev = self.device
        self._keep = []  # tensors the engine points into

        def own(t):
            t = t.to(device=dev, dtype=dtype).contiguous()
            self._keep.append(t)
            return t

        self.embed = own(sd["model.embed_tokens.weight"])
        self.lm_head = own(sd["lm_head.weight"])
        self.final_norm = own(sd["model.norm.weight"])
        cos, sin = rope_tables(self.hd, max_pos, dtype)
        self.rope_cos, self.rope_sin = own(cos), own(sin)
        layers = (_lib.LlamaLayerWeights * n_layers)()
        for l in range(n_layers):
            pfx = "model.layers.%d." % l
            q, k, v, o = (_merged(sd, pfx + "self_attn." + n, dtype, dev, lora_scaling) for n in ATTN_PROJ)
            g, u, d = (_merged(sd, pfx + "mlp." + n, dtype, dev, lora_scaling) for n in MLP_PROJ)
            wqkv = torch.cat([q, k, v], dim=0).contiguous()
            wgu = torch.cat([g, u], dim=0).contiguous()
            ln1 = own(sd[pfx + "input_layernorm.weight"])
            ln2 = own(sd[pfx + "post_attention_layernorm.weight"])
            self._keep += [wqkv, o, wgu, d]
            layers[l] = _lib.LlamaLayerWeights(wqkv.data_ptr(), o.data_ptr(), wgu.data_ptr(), d.data_ptr(),
                                               ln1.data_ptr(), ln2.data_ptr())
            del q, k, v, g, u
        self._layers = layers
        self._init_engine(max_pos, rms_eps)

    @classmethod
    def from_prebuilt(cls, *, embed, lm_head, final_norm, layers, hidden, n_heads, n_layers, inter, vocab,
                      dtype=torch.bfloat16, device="cuda:0", rms_eps=1e-5, max_pos=4096, cache_cap=2048, max_new=512,
                      max_prefill_rows=1024, img_ids=(), eos_id=2, n_seq=1):
        """Engine over already merged/concatenated device tensors: ``layers`` is a list of
        ``(wqkv, wo, wgu, wdown, ln1, ln2)`` (used by the synthetic-weight benchmark, which
        creates the 13.5 GB of weights directly on the GPU)."""
        self = cls.__new__(cls)
        self.n_seq = int(n_seq)
        self.device = torch.device(device)
        self.dtype = dtype
        self.hidden, self.n_heads, self.n_layers, self.inter, self.vocab = hidden, n_heads, n_layers, inter, vocab
        self.hd = hidden // n_heads
        self.cache_cap, self.max_new, self.max_rows = cache_cap, max_new, max_prefill_rows
        self.img_ids = [int(i) for i in img_ids]
        self.eos_id = eos_id
        self.embed, self.lm_head, self.final_norm = embed, lm_head, final_norm
        cos, sin = rope_tables(self.hd, max_pos, dtype)
        self.rope_cos, self.rope_sin = cos.to(self.device), sin.to(self.device)
        self._keep = [embed, lm_head, final_norm, self.rope_cos, self.rope_sin, layers]
        arr = (_lib.LlamaLayerWeights * n_layers)()
        for l, t in enumerate(layers):
            arr[l] = _lib.LlamaLayerWeights(*[x.data_ptr() for x in t])
        self._layers = arr
        self._init_engine(max_pos, rms_eps)
        return self

    def _init_engine(self, max_pos, rms_eps):
        cfg = _lib.LlamaConfig(self.hidden, self.n_heads, self.n_layers, self.inter, self.vocab, max_pos, rms_eps,
                               ops.dt(self.dtype), self.cache_cap, self.max_new, len(self.img_ids), self.eos_id,
                               self.n_seq)
        self._cfg = cfg
        w = _lib.LlamaWeights(self.embed.data_ptr(), self.lm_head.data_ptr(), self.final_norm.data_ptr(),
                              self.rope_cos.data_ptr(), self.rope_sin.data_ptr(), self._layers)
        nbytes = lib().ss_llama_workspace_bytes(C.byref(cfg), self.max_rows)
        self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        ids = (C.c_int32 * max(1, len(self.img_ids)))(*self.img_ids)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().ss_llama_create(C.byref(cfg), C.byref(w), self._ws.data_ptr(), nbytes, self.max_rows, ids,
                                        C.byref(h)), "ss_llama_create")
        self._h = h
        self._views = {}
        self._cur = 0

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib().ss_llama_destroy(h)
            self._h = None

    # ---- zero-copy views into the engine workspace -------------------------------------------
    def select(self, seq):
        """Address sequence slot ``seq`` (0 <= seq < n_seq) with the single-sequence methods
        below (views, lengths, prefill, generate, kv_gather).  Returns self."""
        check(lib().ss_llama_select(self._h, int(seq)), "ss_llama_select")
        self._cur = int(seq)
        return self

    def _buf(self, which, shape, dtype):
        key = (self._cur, which, tuple(shape), dtype)
        if key not in self._views:
            ptr = lib().ss_llama_buffer(self._h, which)
            base = self._ws.data_ptr()
            off = ptr - base
            n = 1
            for s in shape:
                n *= s
            esz = torch.empty(0, dtype=dtype).element_size()
            self._views[key] = self._ws[off:off + n * esz].view(dtype).view(*shape)
        return self._views[key]

    @property
    def k_cache(self):
        return self._buf(0, (self.n_layers, self.n_heads, self.cache_cap, self.hd), self.dtype)

    @property
    def v_cache(self):
        return self._buf(1, (self.n_layers, self.n_heads, self.cache_cap, self.hd), self.dtype)

    @property
    def gen_ids(self):
        return self._buf(2, (self.max_new,), torch.int32)

    @property
    def hidden_rows(self):
        return self._buf(3, (self.max_new, self.hidden), self.dtype)

    @property
    def logits(self):
        return self._buf(4, (self.vocab,), self.dtype)

    @property
    def state(self):
        return self._buf(5, (8,), torch.int32)

    # ---- lengths / cache management --------------------------------------------------------------
    def lengths(self):
        kv, pos = C.c_int64(), C.c_int64()
        check(lib().ss_llama_get_lengths(self._h, C.byref(kv), C.byref(pos)), "ss_llama_get_lengths")
        return kv.value, pos.value

    def set_lengths(self, kv_len, pos):
        check(lib().ss_llama_set_lengths(self._h, kv_len, pos, ops.stream()), "ss_llama_set_lengths")

    def reset(self):
        self.set_lengths(0, 0)

    def past_key_values(self, length=None):
        """Reference-layout views of the live cache (modeling_llama_xformer.py:239-242)."""
        n = self.lengths()[0] if length is None else length
        k, v = self.k_cache, self.v_cache
        return tuple((k[l, :, :n].unsqueeze(0), v[l, :, :n].unsqueeze(0)) for l in range(self.n_layers))

    def load_past_key_values(self, past, pos=None):
        """Copy an external reference-layout cache into the slab (no-op for our own views)."""
        n = past[0][0].shape[2]
        for l, (k, v) in enumerate(past):
            dk, dv = self.k_cache[l, :, :n], self.v_cache[l, :, :n]
            if k.data_ptr() != dk.data_ptr():
                dk.copy_(k[0])
            if v.data_ptr() != dv.data_ptr():
                dv.copy_(v[0])
        self.set_lengths(n, n if pos is None else pos)

    def kv_gather(self, keep_idx):
        idx = torch.as_tensor(keep_idx, dtype=torch.int32, device=self.device).contiguous()
        check(lib().ss_llama_kv_gather(self._h, idx.data_ptr(), idx.numel(), ops.stream()), "ss_llama_kv_gather")

    # ---- forward paths ---------------------------------------------------------------------------------
    def prefill(self, embeds, pos_ids=None, want_hidden=False):
        """embeds [M, hidden] rows appended after the cached prefix.  Returns the post-final-norm
        hidden rows [M, hidden] if want_hidden; the last row's logits land in ``self.logits``."""
        embeds = embeds.to(device=self.device, dtype=self.dtype).contiguous()
        M = embeds.shape[0]
        hid = torch.empty(M, self.hidden, dtype=self.dtype, device=self.device) if want_hidden else None
        pid = None if pos_ids is None else pos_ids.to(device=self.device, dtype=torch.int32).contiguous()
        if M > 128:      # the four prefill projections of this row-count bucket (tile table: seedstory/tune.py)
            code, Hd, I = ops.dt(self.dtype), self.hidden, self.inter
            for n, k in ((3 * Hd, Hd), (Hd, Hd), (2 * I, Hd), (Hd, I)):
                tune.ensure_gemm(M, n, k, code, 0, self.device)
        check(lib().ss_llama_prefill(self._h, embeds.data_ptr(), M, ops.p(pid), ops.p(hid), ops.stream()),
              "ss_llama_prefill")
        return hid

    def prefill_batch(self, embeds, want_hidden=False):
        """``prefill`` for several sequence slots in ONE sweep of the weights (ss_llama_prefill_batch): ``embeds[b]`` =
        slot b's new rows [M_b, hidden] or None.  Returns the per-slot hidden rows (or None) — each slot's last row's
        logits land in that slot's logits buffer and its lengths advance by M_b."""
        S = self.n_seq
        assert len(embeds) == S
        rows = [0 if e is None else int(e.shape[0]) for e in embeds]
        live = [e.to(device=self.device, dtype=self.dtype) for e in embeds if e is not None and e.shape[0]]
        if not live:
            return [None] * S
        M = sum(rows)
        if M > self.max_rows:                      # engine sized for fewer stacked rows: slot by slot, max_rows at a time
            out, step, keep = [], int(self.max_rows), self._cur
            for b, e in enumerate(embeds):
                if not rows[b]:
                    out.append(None)
                    continue
                self.select(b)
                parts = [self.prefill(e[i:i + step], want_hidden=want_hidden) for i in range(0, rows[b], step)]
                out.append(None if not want_hidden else (parts[0] if len(parts) == 1 else torch.cat([p.clone() for p in parts])))
            self.select(keep)
            return out
        stack = live[0].contiguous() if len(live) == 1 else torch.cat(live, dim=0)
        hid = torch.empty(M, self.hidden, dtype=self.dtype, device=self.device) if want_hidden else None
        if M > 128:
            code, Hd, I = ops.dt(self.dtype), self.hidden, self.inter
            for n, k in ((3 * Hd, Hd), (Hd, Hd), (2 * I, Hd), (Hd, I)):
                tune.ensure_gemm(M, n, k, code, 0, self.device)
        arr = (C.c_int64 * S)(*rows)
        check(lib().ss_llama_prefill_batch(self._h, stack.data_ptr(), arr, ops.p(hid), ops.stream()), "ss_llama_prefill_batch")
        if not want_hidden:
            return [None] * S
        out, r0 = [], 0
        for b in range(S):
            out.append(hid[r0:r0 + rows[b]] if rows[b] else None)
            r0 += rows[b]
        return out

    def generate(self, n_steps, last_prompt_id, forced=None):
        """Greedy decode from the current logits; returns the number of generated tokens."""
        forced = [] if forced is None else [int(t) for t in forced]
        arr = (C.c_int32 * max(1, len(forced)))(*forced)
        n = C.c_int64()
        check(lib().ss_llama_generate(self._h, n_steps, int(last_prompt_id), arr, len(forced), C.byref(n),
                                      ops.stream()), "ss_llama_generate")
        return n.value

    def generate_batch(self, n_steps, last_prompt_ids, forced=None, active=None):
        """Greedy decode of all ``n_seq`` slots in lock-step (one sweep of the weights per token
        for the whole batch).  ``last_prompt_ids[b]``, optional ``forced[b]`` token lists and
        ``active[b]`` flags are per slot; returns the per-slot generated-token counts."""
        S = self.n_seq
        assert len(last_prompt_ids) == S
        forced = [[] for _ in range(S)] if forced is None else [[int(t) for t in (f or [])] for f in forced]
        ld = max(1, max(len(f) for f in forced))
        flat = (C.c_int32 * (S * ld))()
        for b, f in enumerate(forced):
            for i, t in enumerate(f):
                flat[b * ld + i] = t
        nf = (C.c_int64 * S)(*[len(f) for f in forced])
        last = (C.c_int32 * S)(*[int(t) for t in last_prompt_ids])
        act = None if active is None else (C.c_int32 * S)(*[1 if a else 0 for a in active])
        out = (C.c_int64 * S)()
        check(lib().ss_llama_generate_batch(self._h, n_steps, last, flat, ld, nf, act, out, ops.stream()),
              "ss_llama_generate_batch")
        return [int(v) for v in out]

    # ---- image-token block decode ---------------------------------------------------------------------------
    # After ``<img>`` the logits processor FORCES the next 65 tokens (``<img_00000>`` .. ``<img_00063>``, ``</img>``:
    # generation.py:19-31 sets their score to max + 10), so their forwards do not depend on one another's sampled
    # output: the decode loop stops at ``<img>`` (ss_llama_set_stop_id), the forced block is fed as ONE batched
    # continuation — the same layers, the same causal attention, the same hidden rows and KV entries, but the 13.2 GB
    # of weights are streamed once for 66 rows instead of 66 times — and the loop resumes from the block's last logits.
    # lm_head is still applied to every block row (the reference computes those logits; they cannot change a forced
    # token), so no arithmetic of the sequential loop is skipped.
    @staticmethod
    def img_block_enabled():
        """Default ON; ``SEEDSTORY_IMG_BLOCK=0`` or the ``img_block_decode`` tuning knob set to 0 restore the token-by-token loop."""
        import os
        return _lib.get_tuning("img_block_decode", 0 if os.environ.get("SEEDSTORY_IMG_BLOCK", "1") == "0" else 1) != 0

    def set_stop_id(self, token_id):
        check(lib().ss_llama_set_stop_id(self._h, int(token_id)), "ss_llama_set_stop_id")

    def _img_block_plan(self, remaining, forced):
        """(tokens the block contributes, rows to feed) for a slot whose loop has just produced ``<img>``."""
        blk = self.img_ids
        m = min(remaining, len(blk) - 1)                          # tokens the block contributes: blk[1 .. m]
        if forced[:m] != blk[1:m + 1][:len(forced[:m])]:
            raise _lib.SSError("forced tokens contradict the image-token schedule that follows <img>")
        rows = m + 1 if remaining > len(blk) - 1 else m            # </img> is fed only when generation goes on after it
        return m, rows

    def _img_block(self, remaining, forced):
        """The slot's decode loop has just produced ``<img>`` (not fed yet).  Feeds the forced block; returns
        (tokens appended, hidden rows [rows, H], id of the last token fed or None when the block ended the budget)."""
        blk = self.img_ids
        m, rows = self._img_block_plan(remaining, forced)
        ids = torch.tensor(blk[:rows], dtype=torch.int32, device=self.device)
        emb = ops.gather_rows(self.embed, ids)
        step = int(self.max_rows)                                   # rows per prefill call the engine was sized for
        hb = torch.cat([self.prefill(emb[i:i + step], want_hidden=True).clone() for i in range(0, rows, step)]) \
            if rows > step else self.prefill(emb, want_hidden=True)
        if _lib.get_tuning("img_block_logits", 1):
            ops.gemm(hb, self.lm_head)                             # the reference's per-position logits (unused: forced)
        return blk[1:m + 1], hb, (blk[rows - 1] if rows == m + 1 else None)

    def generate_img_block(self, n_steps, last_prompt_id, forced=None):
        """``generate`` with the forced image-token runs batched.  Returns (ids LongTensor [n] on the host side list,
        hidden rows [n - 1, hidden]) — the same tokens / rows the token-by-token loop yields."""
        if len(self.img_ids) < 3:
            n = self.generate(n_steps, last_prompt_id, forced)
            return self.gen_ids[:n].tolist(), self.hidden_rows[:max(n - 1, 0)]
        boi = self.img_ids[0]
        forced = [] if forced is None else [int(t) for t in forced]
        ids, hid, rem, last = [], [], min(int(n_steps), 1 << 30), int(last_prompt_id)
        self.set_stop_id(boi)
        try:
            while rem > 0:
                n = self.generate(rem, last, forced)
                g = self.gen_ids[:n].tolist()
                ids += g
                hid.append(self.hidden_rows[:max(n - 1, 0)].clone())
                forced, rem = forced[n:], rem - n
                if n == 0 or g[-1] != boi or rem == 0:
                    break                                           # EOS, the token limit, or the budget ended at <img>
                toks, hb, last_fed = self._img_block(rem, forced)
                ids += toks
                hid.append(hb)
                forced, rem = forced[len(toks):], rem - len(toks)
                if last_fed is None:
                    break
                last = last_fed
        finally:
            self.set_stop_id(-1)
        return ids, torch.cat(hid)[:max(len(ids) - 1, 0)]

    def generate_batch_img_block(self, n_steps, last_prompt_ids, forced=None):
        """``generate_batch`` with the forced image-token runs batched, per slot.  Returns (ids per slot, hidden rows per
        slot).  Slots are independent: one may be inside its block while another still writes its caption."""
        S = self.n_seq
        if len(self.img_ids) < 3:
            ns = self.generate_batch(n_steps, last_prompt_ids, forced)
            return ([self.select(b).gen_ids[:ns[b]].tolist() for b in range(S)],
                    [self.select(b).hidden_rows[:max(ns[b] - 1, 0)] for b in range(S)])
        boi = self.img_ids[0]
        forced = [[] for _ in range(S)] if forced is None else [[int(t) for t in (f or [])] for f in forced]
        ids, hid = [[] for _ in range(S)], [[] for _ in range(S)]
        rem, last, active = [int(n_steps)] * S, [int(t) for t in last_prompt_ids], [True] * S
        self.set_stop_id(boi)
        try:
            blk = self.img_ids
            while any(active):
                n_call = min(rem[b] for b in range(S) if active[b])
                ns = self.generate_batch(n_call, last, forced, active)
                feed = [None] * S          # rows every slot feeds after this call: ONE stacked continuation (weights once)
                plan = {}
                for b in range(S):
                    if not active[b]:
                        continue
                    n = ns[b]
                    self.select(b)
                    g = self.gen_ids[:n].tolist()
                    ids[b] += g
                    hid[b].append(self.hidden_rows[:max(n - 1, 0)].clone())
                    forced[b], rem[b] = forced[b][n:], rem[b] - n
                    if n == 0 or rem[b] == 0 or g[-1] == self.eos_id:
                        active[b] = False
                    elif g[-1] == boi:      # the forced image-token block of this slot
                        m, rows = self._img_block_plan(rem[b], forced[b])
                        feed[b] = torch.tensor(blk[:rows], dtype=torch.int32, device=self.device)
                        plan[b] = (m, rows)
                    else:       # stopped by this call's common limit: feed its last token, resume from fresh logits
                        feed[b] = torch.tensor([g[-1]], dtype=torch.int32, device=self.device)
                        plan[b] = None
                if plan:
                    hbs = self.prefill_batch([None if f is None else ops.gather_rows(self.embed, f) for f in feed],
                                             want_hidden=True)
                    if _lib.get_tuning("img_block_logits", 1) and any(v is not None for v in plan.values()):
                        # the reference's per-position logits of the block rows (unused: those tokens are forced)
                        ops.gemm(torch.cat([hbs[b] for b, v in plan.items() if v is not None]).contiguous(), self.lm_head)
                    for b, v in plan.items():
                        hid[b].append(hbs[b].clone())
                        if v is None:
                            last[b] = ids[b][-1]
                            continue
                        m, rows = v
                        toks = blk[1:m + 1]
                        ids[b] += toks
                        forced[b], rem[b] = forced[b][len(toks):], rem[b] - len(toks)
                        if rows == m + 1:
                            last[b] = blk[rows - 1]
                        else:
                            active[b] = False
        finally:
            self.set_stop_id(-1)
        return ids, [torch.cat(h)[:max(len(i) - 1, 0)] for h, i in zip(hid, ids)]

    def profile_decode(self, n_tokens=4):
        ms = (C.c_float * 8)()
        by = (C.c_double * 4)()
        check(lib().ss_llama_profile_decode(self._h, n_tokens, ms, by, ops.stream()), "ss_llama_profile_decode")
        return {"gemv_ms": ms[0], "attn_ms": ms[1], "gemv_down_ms": ms[2], "misc_ms": ms[3], "token_ms": ms[4],
                "gemv_bytes": by[0], "gemv_launches": by[1], "gemv_down_bytes": by[2], "gemv_down_launches": by[3]}
