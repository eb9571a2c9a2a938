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
import torch


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
