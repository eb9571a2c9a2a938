"""Id-level story context manager: the drivers' prompt/window bookkeeping restated on token ids
(reference: src/inference/gen_george.py:168-255 does it with string surgery + full re-tokenisation;
src/inference/vis_george_sink.py:214-316 adds the multimodal attention sink on the KV cache).

Two eviction policies when more than ``window`` images are in context:

* ``recompute`` — what gen_george.py does as released (:235-239): cut the prompt through the first
  ``</img>``, drop the oldest image feature, re-prefill everything.
* ``sink`` — the multimodal attention sink as *intended* by vis_george_sink.py:266-295 (SURVEY.md
  Appendix A.5; the released script computes the sliced cache and then discards it, :316): keep in the KV
  cache the first ``n_start`` positions plus, for every evicted image, the windows (-4,+8) around its
  ``<img>`` and (-8,+4) around its ``</img>``; keys keep their original rotary phase (cached post-RoPE),
  new queries get window-relative positions.  Implemented on the engine's KV slab with
  ``LlamaEngine.kv_gather`` (``ss_llama_kv_gather``): no tensor concatenation, no re-prefill.
"""
import re

import torch

from .tokenizer import BOI_TOKEN, EOI_TOKEN, IMG_TOKEN


def sink_keep_indices(n_kv, boi, eoi, sink_len, first, n_start=4, boi_win=(-4, 8), eoi_win=(-8, 4)):
    """KV indices kept by one eviction; ``boi``/``eoi`` index the KV (sink prefix included).
    Returns (indices, new sink length)."""
    keep = list(range(0, n_start if first else sink_len))
    keep += list(range(boi + boi_win[0], boi + boi_win[1]))
    keep += list(range(eoi + eoi_win[0], eoi + eoi_win[1]))
    new_sink = len(keep)
    keep += list(range(eoi + 1, n_kv))
    return keep, new_sink


class StoryContext:
    def __init__(self, bos_id, boi_id, eoi_id, img_placeholder_ids, window=8):
        self.bos, self.boi, self.eoi = bos_id, boi_id, eoi_id
        self.img_tokens = [boi_id] + list(img_placeholder_ids) + [eoi_id]      # '<img>' + 64 placeholders + '</img>'
        self.window = window
        self.ids = []
        self.image_embeds = None
        self.sink_len = 0            # KV entries in front of the live window (sink mode)
        self.n_evicted = 0

    # ---- prompt construction (gen_george.py:168-180, 231) ---------------------------------------------
    def start(self, question_ids, first_image_embeds):
        self.ids = [self.bos] + list(question_ids) + self.img_tokens
        self.image_embeds = first_image_embeds
        self.sink_len = 0
        self.n_evicted = 0

    def append_step(self, text_ids, img_gen_feat):
        """prompt = prompt + text + image_tokens; image_embeds = cat(image_embeds, img_gen_feat) (:224,231)."""
        self.ids = self.ids + list(text_ids) + self.img_tokens
        self.image_embeds = torch.cat([self.image_embeds, img_gen_feat], dim=0)

    def advance(self, out):
        """One story step from a ``ContinuousLVLM.generate`` result: the caption ids in front of the generated
        ``<img>`` are kept VERBATIM (no decode -> regex -> re-tokenise round trip, which is what lets the KV cache of
        the previous step stay valid), then window eviction by recompute.  Returns the number of evicted images."""
        gen = out['generate_ids'].tolist()
        self.append_step(gen[:gen.index(self.boi)] if self.boi in gen else gen, out['img_gen_feat'])
        return self.evict_recompute() if self.over_window() else 0

    def masks(self, device):
        """ids_cmp_mask / embeds_cmp_mask as the drivers build them from <img>/</img> positions (:246-255)."""
        ids = torch.tensor(self.ids, dtype=torch.long)
        b = torch.where(ids == self.boi)[0].tolist()
        e = torch.where(ids == self.eoi)[0].tolist()
        m = torch.zeros(1, len(self.ids), dtype=torch.bool)
        for i in range(self.image_embeds.shape[0]):
            m[0, b[i] + 1:e[i]] = True
        return m.to(device), torch.ones(self.image_embeds.shape[0], dtype=torch.bool, device=device)

    def input_ids(self, device):
        return torch.tensor([self.ids], dtype=torch.long, device=device)

    def over_window(self):
        return self.image_embeds is not None and self.image_embeds.shape[0] > self.window

    # ---- eviction -----------------------------------------------------------------------------------------
    def evict_recompute(self):
        """Drop everything through the first </img> and the oldest image (gen_george.py:235-239)."""
        n = 0
        while self.over_window():
            e = self.ids.index(self.eoi)
            self.ids = [self.bos] + self.ids[e + 1:]          # the driver re-adds BOS after re-tokenising (:243)
            self.image_embeds = self.image_embeds[1:]
            n += 1
        self.n_evicted += n
        return n

    def evict_sink(self, engine, kv_len):
        """Sink eviction on the engine's KV slab. ``kv_len`` = live KV entries (sink prefix + window tokens that
        are already cached).  Returns the new kv_len; ``self.ids`` becomes the surviving window."""
        while self.over_window():
            b = self.ids.index(self.boi) + self.sink_len      # KV coordinates include the sink prefix
            e = self.ids.index(self.eoi) + self.sink_len
            first = self.sink_len == 0
            keep, new_sink = sink_keep_indices(kv_len, b, e, self.sink_len, first)
            engine.kv_gather(keep)
            cut = self.ids.index(self.eoi) + 1
            self.ids = self.ids[cut:]
            self.image_embeds = self.image_embeds[1:]
            kv_len = len(keep)
            self.sink_len = new_sink
            self.n_evicted += 1
        return kv_len


class PromptStory:
    """The driver's prompt bookkeeping EXACTLY as released — on the prompt STRING (gen_george.py:23,168-176,196,
    231-243) — for runs that must reproduce the reference's token stream (``--parity``):

    * the generated text is decoded, scrubbed with ``re.sub(r'\\s*<[^>]*>\\s*', ' ', text).strip()`` (:196) and
      appended as a string, so every step re-tokenises the whole prompt (the round trip is not the identity: spacing
      around the added tokens and sentencepiece's per-segment dummy prefix change the ids);
    * window eviction cuts the string through the first ``</img>`` **plus ``len('[INST]')`` = 6 more characters**
      (:237) — ``instruction_prompt`` is ``'{instruction}'`` (:23), there is no ``[INST]`` in the prompt, so the first six
      characters of the caption that followed the evicted image are dropped as well;
    * ``input_ids = [bos] + encode(prompt, add_special_tokens=False)`` (:233,243).

    Same surface as ``StoryContext`` (``ids``, ``input_ids``, ``masks``, ``image_embeds``, ``advance``); no KV reuse is
    possible in this mode (the ids of the retained window change when the string in front of them is cut)."""

    INST_SKIP = len('[INST]')

    def __init__(self, tokenizer, window=8, num_img_in_tokens=64, instruction_prompt='{instruction}'):
        self.tok = tokenizer
        self.window = window
        self.instruction_prompt = instruction_prompt
        self.image_tokens = BOI_TOKEN + ''.join(IMG_TOKEN.format(i) for i in range(num_img_in_tokens)) + EOI_TOKEN
        self.boi = tokenizer.encode(BOI_TOKEN, add_special_tokens=False)[0]
        self.eoi = tokenizer.encode(EOI_TOKEN, add_special_tokens=False)[0]
        self.prompt = ''
        self.image_embeds = None
        self.n_evicted = 0

    @staticmethod
    def clean(text):
        return re.sub(r'\s*<[^>]*>\s*', ' ', text).strip()

    def start(self, question, first_image_embeds):
        self.prompt = self.instruction_prompt.format_map({'instruction': question + self.image_tokens})
        self.image_embeds = first_image_embeds
        self.n_evicted = 0

    def advance(self, out):
        text = self.clean(out['text'])
        self.image_embeds = torch.cat((self.image_embeds, out['img_gen_feat']), dim=0)      # :224
        self.prompt = self.prompt + text + self.image_tokens                                # :231
        n = 0
        while self.image_embeds.shape[0] > self.window:                                     # :235-239
            e = self.prompt.index(EOI_TOKEN)
            self.prompt = self.prompt[e + len(EOI_TOKEN) + self.INST_SKIP:]
            self.image_embeds = self.image_embeds[1:]
            n += 1
        self.n_evicted += n
        return n

    @property
    def ids(self):
        return [self.tok.bos_token_id] + self.tok.encode(self.prompt, add_special_tokens=False)

    def input_ids(self, device):
        return torch.tensor([self.ids], dtype=torch.long, device=device)

    def masks(self, device):
        ids = torch.tensor(self.ids, dtype=torch.long)
        b = torch.where(ids == self.boi)[0].tolist()
        e = torch.where(ids == self.eoi)[0].tolist()
        m = torch.zeros(1, ids.numel(), dtype=torch.bool)
        for i in range(self.image_embeds.shape[0]):
            m[0, b[i] + 1:e[i]] = True
        return m.to(device), torch.ones(self.image_embeds.shape[0], dtype=torch.bool, device=device)
