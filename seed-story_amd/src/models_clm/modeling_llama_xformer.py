"""MI355X-native drop-in for the reference's ``src/models_clm/modeling_llama_xformer.py``.

``LlamaForCausalLM`` keeps the reference's surface that the hot path touches:
``from_pretrained(path, low_cpu_mem_usage, torch_dtype)``, HF parameter names
(``model.embed_tokens.weight`` … ``lm_head.weight``), ``get_input_embeddings()``,
``resize_token_embeddings()``, and the mutable attributes ``use_kv_cache_head`` /
``kv_cache_head`` / ``past_key_values`` (reference :676-678) that the drivers poke
(gen_george.py:165, vis_george_sink.py:171-173,244,293).

The decoder stack itself (reference LlamaRMSNorm :97-115, rotary :118-173, LlamaAttention
:217-301 with xformers FMHA, LlamaMLP :176-191, LlamaModel.forward :532-666) runs inside the
native engine (``seedstory.llama.LlamaEngine`` -> ``ss_llama_*``): there is no torch compute
here and no CPU path.
"""
import json
import os

import torch
from torch import nn

from seedstory.llama import LlamaEngine


class LlamaConfig:
    """The handful of transformers.LlamaConfig fields the path needs."""

    def __init__(self, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                 vocab_size=32000, max_position_embeddings=4096, rms_norm_eps=1e-5, eos_token_id=2, bos_token_id=1,
                 **kwargs):
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.vocab_size = vocab_size
        self.max_position_embeddings = max_position_embeddings
        self.rms_norm_eps = rms_norm_eps
        self.eos_token_id = eos_token_id
        self.bos_token_id = bos_token_id
        self.use_cache = True

    @classmethod
    def from_pretrained(cls, path):
        with open(os.path.join(path, "config.json")) as f:
            return cls(**json.load(f))


class _W(nn.Module):
    def __init__(self, o, i):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i), requires_grad=False)


class _Norm(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d), requires_grad=False)


class _Attn(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = _W(h, h), _W(h, h), _W(h, h), _W(h, h)


class _MLP(nn.Module):
    def __init__(self, h, i):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = _W(i, h), _W(i, h), _W(h, i)


class _Layer(nn.Module):
    def __init__(self, h, i):
        super().__init__()
        self.self_attn = _Attn(h)
        self.mlp = _MLP(h, i)
        self.input_layernorm = _Norm(h)
        self.post_attention_layernorm = _Norm(h)


class _Embedding(nn.Module):
    def __init__(self, v, h):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(v, h), requires_grad=False)

    def forward(self, input_ids):
        from seedstory import ops
        shape = tuple(input_ids.shape)
        rows = ops.gather_rows(self.weight.data, input_ids.reshape(-1))
        return rows.view(*shape, self.weight.shape[1])


class _Model(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embed_tokens = _Embedding(cfg.vocab_size, cfg.hidden_size)
        self.layers = nn.ModuleList([_Layer(cfg.hidden_size, cfg.intermediate_size)
                                     for _ in range(cfg.num_hidden_layers)])
        self.norm = _Norm(cfg.hidden_size)


class LlamaForCausalLM(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = _Model(config)
        self.lm_head = _W(config.vocab_size, config.hidden_size)
        self.past_key_values = None
        self.kv_cache_head = None
        self.use_kv_cache_head = True
        self._engine = None
        self._engine_sig = None
        # engine sizing knobs (KV slots, generated-token ring, rows per prefill call)
        self.cache_cap = 2048
        self.max_new = 512
        self.max_prefill_rows = 1280

    # ---- reference surface ------------------------------------------------------------------
    def get_input_embeddings(self):
        return self.model.embed_tokens

    def get_output_embeddings(self):
        return self.lm_head

    def resize_token_embeddings(self, vocab_size):
        """Grow/shrink embed_tokens and lm_head rows (peft_models.py:43-45); new rows keep the
        mean of the old table like HF's default initialiser is NOT reproduced (random there)."""
        for holder in (self.model.embed_tokens, self.lm_head):
            old = holder.weight.data
            new = torch.zeros(vocab_size, old.shape[1], dtype=old.dtype, device=old.device)
            n = min(vocab_size, old.shape[0])
            new[:n] = old[:n]
            holder.weight = nn.Parameter(new, requires_grad=False)
        self.config.vocab_size = vocab_size
        self._engine = None
        return self.model.embed_tokens

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, low_cpu_mem_usage=True, torch_dtype=None, **kwargs):
        """Loads an HF LLaMA folder (config.json + *.safetensors / pytorch_model*.bin)."""
        cfg = LlamaConfig.from_pretrained(pretrained_model_name_or_path)
        model = cls(cfg)
        from seedstory import ckpt as _ckpt
        sd = _ckpt.read_weights(pretrained_model_name_or_path)
        # HF checkpoints carry the RoPE inverse-frequency buffers of older transformers versions; they are derived data
        _ckpt.load_checked(model, {k: v for k, v in sd.items() if not k.endswith("rotary_emb.inv_freq")}, "llama:")
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model

    def init_synthetic(self, seed=0, std=0.02):
        """N(0, 0.02) weights like the reference ``_init_weights`` (:399-408), on the current device."""
        for name, p in self.named_parameters():
            if p.dim() == 1:
                p.data.fill_(1.0)
            elif p.is_cuda:
                p.data.normal_(0.0, std)
            else:
                g = torch.Generator().manual_seed(seed + hash(name) % 10007)
                p.data.copy_(torch.randn(p.shape, generator=g) * std)
        self._engine = None
        return self

    # ---- engine --------------------------------------------------------------------------------
    # ---- generation (HF-4.34 greedy search semantics) -------------------
    def collect_flat_state(self):
        """HF-named flat weights for the engine: unwraps the peft-style children that
        ``get_peft_model_with_resize_embedding`` grafts on (LoRA factors, modules_to_save norms)."""
        flat = {}
        for name, prm in self.named_parameters():
            if ".original_module." in name:
                continue
            name = name.replace(".modules_to_save.default.", ".")
            flat[name] = prm.data
        return flat

    def engine_for_generation(self, img_ids):
        self._img_ids = tuple(img_ids)      # forward() reuses the engine of the last generate() instead of rebuilding
        lora = getattr(self, "_lora_scaling", None)
        p0 = self.lm_head.weight
        sig = (p0.data_ptr(), p0._version, p0.dtype, str(p0.device), tuple(img_ids), self.cache_cap, self.max_new,
               self.max_prefill_rows)
        if self._engine is None or self._engine_sig != sig:
            if not p0.is_cuda:
                raise RuntimeError("LlamaForCausalLM must be moved to the GPU first (no CPU path)")
            c = self.config
            self._engine = LlamaEngine(self.collect_flat_state(), hidden=c.hidden_size,
                                       n_heads=c.num_attention_heads, n_layers=c.num_hidden_layers,
                                       inter=c.intermediate_size, vocab=c.vocab_size, dtype=p0.dtype, device=p0.device,
                                       rms_eps=c.rms_norm_eps, max_pos=c.max_position_embeddings,
                                       cache_cap=self.cache_cap, max_new=self.max_new,
                                       max_prefill_rows=self.max_prefill_rows, img_ids=img_ids,
                                       eos_id=c.eos_token_id, lora_scaling=lora if lora is not None else 2.0)
            self._engine_sig = sig
        return self._engine

    # ---- single forward call (reference :703-794, inference part) ------------------------------------------
    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None):
        """One model call with the reference's argument meaning: ``inputs_embeds`` (or ``input_ids``) [1, q] rows are fed
        after ``past_key_values`` (tuple over layers of (k, v) [1, heads, kv, head_dim], keys post-RoPE); ``position_ids``
        [1, q] default to kv .. kv+q-1; the mask is always the bottom-right causal one (the reference passes
        ``attention_mask=None`` down from ``prepare_inputs_for_generation``, :826,844).  Returns logits for ALL q rows
        (lm_head on every row, :759), the grown cache, and — unlike the reference, which can return every layer's input —
        ``hidden_states = (last,)``: the post-final-norm states, the only entry the path reads (models.py:182-197 uses
        ``hidden_states[-1]``).  Side effects as the reference: ``self.past_key_values`` (:778), ``kv_cache_head``
        (:780-784).  With ``labels`` or a batch of sequences: `_forward_sequences` (loss :761-772, forward only)."""
        if output_attentions:
            raise NotImplementedError("attention probabilities are never materialised (flash attention)")
        eng = self.engine_for_generation(tuple(getattr(self, "_img_ids", ())))
        dev = eng.device
        if inputs_embeds is None:
            inputs_embeds = self.model.embed_tokens(input_ids.to(dev))
        if labels is not None or inputs_embeds.shape[0] != 1:
            if past_key_values is not None:
                raise ValueError("the batched / loss forward takes whole sequences (no past_key_values)")
            return self._forward_sequences(eng, inputs_embeds.to(dev), labels, position_ids, output_hidden_states, return_dict)
        rows = inputs_embeds[0].to(dev)
        q = rows.shape[0]
        if past_key_values is None:
            eng.reset()
            kv = 0
        else:
            eng.load_past_key_values(past_key_values)
            kv = eng.lengths()[0]
        pos = None
        if position_ids is not None:
            pos = position_ids.reshape(-1).to(device=dev, dtype=torch.int32)
        elif kv:
            pos = torch.arange(kv, kv + q, dtype=torch.int32, device=dev)
        hid = eng.prefill(rows, pos_ids=pos, want_hidden=True)                    # [q, H], post final norm
        from seedstory import ops as _ops
        logits = _ops.gemm(hid.contiguous(), self.lm_head.weight).unsqueeze(0)     # [1, q, V]
        pkv = eng.past_key_values()
        self.past_key_values = pkv
        if self.use_kv_cache_head and not self.training:
            n_in = q if input_ids is None else input_ids.shape[1]
            self.kv_cache_head = n_in if self.kv_cache_head is None else self.kv_cache_head + n_in
        out = CausalLMOutputWithPast(logits=logits, past_key_values=pkv,
                                     hidden_states=(hid.unsqueeze(0),) if output_hidden_states else None)
        if return_dict is False:
            return (logits, pkv) + ((out.hidden_states,) if output_hidden_states else ())
        return out


    def _forward_sequences(self, eng, inputs_embeds, labels, position_ids, output_hidden_states, return_dict):
        """The training-side call (reference :703-794 with ``labels``; forward only): ``inputs_embeds``
        [bz, sq, H] are bz independent causal sequences.  ``attention_mask`` plays no role in the outputs — the reference's
        xformers call ignores it too (``attn_bias=LowerTriangularMask()``, :281-295; the additive mask of :274 only
        touches ``attn_weights``, which is discarded), so right-padded rows are ordinary tokens whose labels are -100.
        logits for every row (:759); loss = CrossEntropyLoss over logits[..., :-1, :] / labels[..., 1:] (:761-772), one
        fixed-order reduction on the device, rounded to the model dtype like torch's."""
        from seedstory import ops as _ops
        bz, sq, H = inputs_embeds.shape
        if sq > eng.max_rows:
            raise ValueError("sequence length %d exceeds max_prefill_rows=%d" % (sq, eng.max_rows))
        hid = torch.empty(bz, sq, H, dtype=inputs_embeds.dtype, device=inputs_embeds.device)
        V = self.lm_head.weight.shape[0]
        logits = torch.empty(bz, sq, V, dtype=inputs_embeds.dtype, device=inputs_embeds.device)
        for b in range(bz):
            eng.reset()
            pos = None
            if position_ids is not None:
                pr = position_ids if position_ids.dim() == 1 or position_ids.shape[0] == 1 else position_ids[b]
                pos = pr.reshape(-1).to(device=inputs_embeds.device, dtype=torch.int32)
            hb = eng.prefill(inputs_embeds[b].contiguous(), pos_ids=pos, want_hidden=True)     # [sq, H], post final norm
            hid[b].copy_(hb)
            _ops.gemm(hid[b], self.lm_head.weight, out=logits[b])
        eng.reset()
        loss = None
        if labels is not None:
            labels = labels.to(inputs_embeds.device)
            if labels.shape != (bz, sq):
                raise ValueError("labels must be [batch, sequence]")
            bad = (labels != -100) & ((labels < 0) | (labels >= V))
            if bool(bad.any()):
                raise ValueError("label outside [0, vocab) (and not -100)")
            shift_logits = logits[:, :-1, :].reshape(-1, V).contiguous()
            shift_labels = labels[:, 1:].reshape(-1)
            loss, n_valid = _ops.cross_entropy(shift_logits, shift_labels, -100)
            if float(n_valid) == 0.0:
                loss = torch.full((), float("nan"), device=loss.device)      # CrossEntropyLoss over no target
            loss = loss.to(inputs_embeds.dtype)
        self.past_key_values = None
        out = CausalLMOutputWithPast(logits=logits, past_key_values=None,
                                     hidden_states=(hid,) if output_hidden_states else None, loss=loss)
        if return_dict is False:
            t = (logits, None) + ((out.hidden_states,) if output_hidden_states else ())
            return ((loss,) + t) if loss is not None else t
        return out

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None,
                                      **kwargs):
        """What HF's greedy loop feeds each step (reference :796-852).  With ``use_kv_cache_head`` and a cache, the rows
        ``[kv_cache_head:]`` of the running sequence go in (ids, and embeds when more than one row is left) with the
        matching slice of ``cumsum(mask) - 1`` as positions; otherwise the last token only.  Embeddings are used on the
        first step only.  The mask itself is dropped (the attention is bottom-right causal by construction)."""
        head_mode = self.use_kv_cache_head and not self.training
        cut = self.kv_cache_head if head_mode else -1
        if past_key_values:
            input_ids = input_ids[:, cut:]
            if head_mode and inputs_embeds is not None:
                inputs_embeds = inputs_embeds[:, cut:]
        position_ids = kwargs.get("position_ids", None)
        if attention_mask is not None and position_ids is None:
            position_ids = attention_mask.long().cumsum(-1) - 1
            position_ids.masked_fill_(attention_mask == 0, 1)
            if past_key_values:
                position_ids = position_ids[:, cut:].unsqueeze(-1) if head_mode else position_ids[:, -1].unsqueeze(-1)
        if inputs_embeds is not None and past_key_values is None:
            model_inputs = {"inputs_embeds": inputs_embeds, "input_ids": input_ids}
        elif head_mode and past_key_values is not None and input_ids.shape[1] > 1:
            model_inputs = {"inputs_embeds": inputs_embeds, "input_ids": input_ids}
        else:
            model_inputs = {"input_ids": input_ids}
        model_inputs.update({"position_ids": position_ids, "past_key_values": past_key_values,
                             "use_cache": kwargs.get("use_cache"), "attention_mask": None})
        return model_inputs

    @torch.no_grad()
    def generate(self, input_ids=None, inputs_embeds=None, logits_processor=None, past_key_values=None,
                 max_new_tokens=120, output_hidden_states=True, return_dict_in_generate=True, forced_tokens=None,
                 **unused):
        """Greedy search as ``ContinuousLVLM.generate`` drives it (reference models.py:146-153):
        ``inputs_embeds`` feed step 0, ``input_ids`` is the running sequence, ``do_sample=False``
        (temperature / top_p are inert), the image-token logits processor is applied on device."""
        img_ids = ()
        for proc in (logits_processor or []):
            img_ids = tuple(getattr(proc, "img_ids_list", ()))
        eng = self.engine_for_generation(img_ids)
        dev = eng.device
        input_ids = input_ids.to(dev)
        S = input_ids.shape[1]
        if inputs_embeds is None:
            inputs_embeds = self.model.embed_tokens(input_ids)
        rows = inputs_embeds[0]
        if past_key_values is None:
            eng.reset()
            fed0 = S
            hid0 = eng.prefill(rows, want_hidden=output_hidden_states)
        else:
            head = self.kv_cache_head if (self.use_kv_cache_head and self.kv_cache_head is not None) else S - 1
            eng.load_past_key_values(past_key_values)
            fed0 = S - head
            pos = torch.arange(head, S, dtype=torch.int32, device=dev)      # cumsum(mask)-1 sliced (:811-816)
            hid0 = eng.prefill(rows[head:], pos_ids=pos, want_hidden=output_hidden_states)
            eng.set_lengths(eng.lengths()[0], S)
        if max_new_tokens > eng.max_new:
            raise ValueError("max_new_tokens=%d exceeds the engine's generated-token ring (max_new=%d): raise "
                             "LlamaForCausalLM.max_new before the first generate()" % (max_new_tokens, eng.max_new))
        if img_ids and eng.img_block_enabled():
            # the 65 processor-forced tokens behind <img> run as one batched continuation (seedstory/llama.py)
            gen_list, hr = eng.generate_img_block(max_new_tokens, int(input_ids[0, -1]), forced_tokens)
            n = len(gen_list)
            gen = torch.tensor(gen_list, dtype=torch.long, device=dev)
        else:
            n = eng.generate(max_new_tokens, int(input_ids[0, -1]), forced_tokens)
            gen = eng.gen_ids[:n].to(torch.long)
            hr = eng.hidden_rows[:max(n - 1, 0)]
        sequences = torch.cat([input_ids[0], gen]).unsqueeze(0)
        hidden_states = None
        if output_hidden_states:
            steps = [(hid0.unsqueeze(0),)]
            steps += [(hr[j].view(1, 1, -1),) for j in range(hr.shape[0])]
            hidden_states = tuple(steps)
        self.past_key_values = eng.past_key_values()
        if self.use_kv_cache_head:
            adv = fed0 + max(n - 1, 0)
            self.kv_cache_head = adv if self.kv_cache_head is None else self.kv_cache_head + adv
        return GenerateOutput(sequences=sequences, hidden_states=hidden_states, attentions=None)


class CausalLMOutputWithPast:
    def __init__(self, logits, past_key_values, hidden_states=None, loss=None, attentions=None):
        self.loss, self.logits, self.past_key_values = loss, logits, past_key_values
        self.hidden_states, self.attentions = hidden_states, attentions

    def __getitem__(self, i):
        if isinstance(i, str):        # ModelOutput's dict-style access (models.py:69 reads output_lm['loss'])
            return getattr(self, i)
        return (self.logits, self.past_key_values, self.hidden_states)[i]


class GenerateOutput:
    def __init__(self, sequences, hidden_states, attentions):
        self.sequences = sequences
        self.hidden_states = hidden_states
        self.attentions = attentions
