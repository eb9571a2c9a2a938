"""LLaMA sentencepiece tokenizer + the 66 added image tokens.

The reference builds ``transformers.LlamaTokenizer.from_pretrained('pretrained/cvlm_llama2_tokenizer')``
(configs/tokenizer/clm_llama_tokenizer.yaml; transformers==4.34.0 — the *slow*, sentencepiece-backed class in its
default ``legacy=True`` mode) and uses exactly four things of it: ``encode(text, add_special_tokens=False)``
(gen_george.py:104-105,174,233), ``bos_token_id`` (:175), ``decode(ids, skip_special_tokens=False)``
(models.py:156) and ``__call__(prompt, return_tensors='pt').input_ids`` (models.py:119-121).  The transformers in this
image (5.x) replaced that class with a `tokenizers`-backed one whose behaviour around added tokens differs, so the
four calls are restated here directly on the sentencepiece model.  Parity with 4.34 itself is unpinned (4.34 and the
tokenizer folder are both absent); against the INSTALLED transformers 5.15 the ids of every driver-shaped string are
pinned on a LLaMA-like sentencepiece model (tests/test_prompt_cpu.py::test_tokenizer_pinned_on_installed_transformers,
which also states the two places where 5.15 departs from 4.34-slow).  The behaviour below follows 4.34's published
source:

  encode   the text is split on the added tokens (longest match, left to right; nothing stripped around them); every
           other non-empty segment is sentencepiece-encoded on its own — so, in legacy mode, EVERY segment gets the
           dummy-prefix ``▁`` (a text that follows ``</img>`` starts with a ``▁`` piece);  added tokens map to their ids.
  decode   ids are converted to pieces; runs of ordinary pieces are detokenised by sentencepiece (leading dummy space of
           the whole string dropped), added tokens are emitted verbatim, and the parts are joined with single spaces
           (``spaces_between_special_tokens=True``) — which is why the drivers scrub the text with
           ``re.sub(r'\\s*<[^>]*>\\s*', ' ', text)`` (gen_george.py:196).

Host-side string processing: no GPU work, no oracle dependency.
"""
import json
import os

import torch

BOI_TOKEN = "<img>"
EOI_TOKEN = "</img>"
IMG_TOKEN = "<img_{:05d}>"


def image_token_strings(n=64):
    return [BOI_TOKEN] + [IMG_TOKEN.format(i) for i in range(n)] + [EOI_TOKEN]


class _Encoding:
    def __init__(self, ids):
        self.input_ids = ids


class LlamaTokenizer:
    def __init__(self, vocab_file, added_tokens=None, bos_token="<s>", eos_token="</s>", unk_token="<unk>"):
        import sentencepiece as spm
        self.sp_model = spm.SentencePieceProcessor()
        self.sp_model.Load(vocab_file)
        self.vocab_file = vocab_file
        n = self.sp_model.GetPieceSize()
        if added_tokens is None:
            added_tokens = {t: n + i for i, t in enumerate(image_token_strings())}
        elif not isinstance(added_tokens, dict):
            added_tokens = {t: n + i for i, t in enumerate(added_tokens)}
        self.added_tokens_encoder = dict(added_tokens)
        self.added_tokens_decoder = {i: t for t, i in self.added_tokens_encoder.items()}
        self._by_len = sorted(self.added_tokens_encoder, key=len, reverse=True)
        self.bos_token, self.eos_token, self.unk_token = bos_token, eos_token, unk_token
        self.bos_token_id = self.sp_model.PieceToId(bos_token)
        self.eos_token_id = self.sp_model.PieceToId(eos_token)
        self.unk_token_id = self.sp_model.unk_id()
        self.all_special_tokens = [bos_token, eos_token, unk_token]

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kw):
        """A tokenizer folder: ``tokenizer.model`` (+ ``added_tokens.json`` = {token: id})."""
        folder = pretrained_model_name_or_path
        added = None
        p = os.path.join(folder, "added_tokens.json")
        if os.path.exists(p):
            with open(p) as f:
                added = json.load(f)
        return cls(os.path.join(folder, "tokenizer.model"), added_tokens=added)

    def __len__(self):
        return self.sp_model.GetPieceSize() + len(self.added_tokens_encoder)

    # ---- text -> ids -----------------------------------------------------------------------------------
    def _split_on_added(self, text):
        out, cur, i = [], [], 0
        while i < len(text):
            hit = None
            if text[i] == "<":
                for t in self._by_len:
                    if text.startswith(t, i):
                        hit = t
                        break
            if hit is None:
                cur.append(text[i])
                i += 1
            else:
                if cur:
                    out.append(("text", "".join(cur)))
                    cur = []
                out.append(("added", hit))
                i += len(hit)
        if cur:
            out.append(("text", "".join(cur)))
        return out

    def encode(self, text, add_special_tokens=True):
        ids = [self.bos_token_id] if add_special_tokens else []
        for kind, seg in self._split_on_added(text):
            if kind == "added":
                ids.append(self.added_tokens_encoder[seg])
            else:
                ids.extend(self.sp_model.EncodeAsIds(seg))          # dummy prefix per segment (legacy mode)
        return ids

    def __call__(self, text, return_tensors=None, add_special_tokens=True):
        ids = self.encode(text, add_special_tokens=add_special_tokens)
        return _Encoding(torch.tensor([ids], dtype=torch.long) if return_tensors == "pt" else ids)

    # ---- ids -> text -----------------------------------------------------------------------------------
    def decode(self, token_ids, skip_special_tokens=False):
        if isinstance(token_ids, torch.Tensor):
            token_ids = token_ids.tolist()
        parts, run = [], []
        n = self.sp_model.GetPieceSize()

        def flush():
            if run:
                parts.append(self.sp_model.DecodeIds(run))
                run.clear()

        for i in (int(t) for t in token_ids):
            if i >= n:
                flush()
                parts.append(self.added_tokens_decoder[i])
            elif i in (self.bos_token_id, self.eos_token_id):
                if not skip_special_tokens:
                    flush()
                    parts.append(self.bos_token if i == self.bos_token_id else self.eos_token)
            else:
                run.append(i)
        flush()
        return " ".join(parts)
