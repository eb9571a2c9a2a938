"""SURVEY.md §8f row 2 — tokenizer + prompt/window bookkeeping, CPU only.

* ``seedstory.tokenizer.LlamaTokenizer``: the four calls the reference makes on transformers-4.34's slow LLaMA tokenizer,
  on a sentencepiece model trained here (the real ``cvlm_llama2_tokenizer`` folder is not in the image; parity with
  4.34 unpinned — the assertions state the published behaviour: split on added tokens, per-segment dummy prefix,
  space-joined decode).
* ``seedstory.story.PromptStory``: the driver's string surgery (gen_george.py:168-176,196,231-243) against literal
  expected strings, including the ``len('[INST]')`` skip on eviction; and the id-level ``StoryContext`` beside it."""

import pytest
import torch

from seedstory.story import PromptStory, StoryContext
from seedstory.tokenizer import BOI_TOKEN, EOI_TOKEN, LlamaTokenizer, image_token_strings

CORPUS = ["George the monkey looked at the man with the yellow hat.", "The man smiled and opened the door of the house.",
          "They walked to the city zoo and saw a big elephant.", "What happens next in the story?",
          "George climbed a tree and waved at the children below."]


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    import sentencepiece as spm
    d = tmp_path_factory.mktemp("tok")
    open(d / "corpus.txt", "w").write("\n".join(CORPUS * 40))
    spm.SentencePieceTrainer.train(input=str(d / "corpus.txt"), model_prefix=str(d / "tokenizer"), vocab_size=400,
                                   model_type="bpe", bos_id=1, eos_id=2, unk_id=0, pad_id=-1, byte_fallback=True,
                                   character_coverage=1.0, add_dummy_prefix=True, minloglevel=2)
    import json
    json.dump({t: 400 + i for i, t in enumerate(image_token_strings())}, open(d / "added_tokens.json", "w"))
    return LlamaTokenizer.from_pretrained(str(d))


IMG = "".join(image_token_strings())


def test_tokenizer_reference_call_surface(tok):
    assert len(tok) == 466 and tok.bos_token_id == 1 and tok.eos_token_id == 2
    assert tok.encode(BOI_TOKEN, add_special_tokens=False) == [400]                         # gen_george.py:104
    assert tok.encode(EOI_TOKEN, add_special_tokens=False) == [465]                         # :105
    assert tok.encode(IMG, add_special_tokens=False) == list(range(400, 466))               # generation.py:14-17
    q = "What happens next in the story?"
    ids = tok.encode(q + IMG + "George climbed a tree", add_special_tokens=False)
    k = ids.index(400)
    assert ids[:k] == tok.sp_model.EncodeAsIds(q) and ids[k:k + 66] == list(range(400, 466))
    tail = ids[k + 66:]
    assert tail == tok.sp_model.EncodeAsIds("George climbed a tree")                        # its own segment ...
    assert tok.sp_model.IdToPiece(tail[0]).startswith("▁")                             # ... with the dummy prefix
    enc = tok(q, return_tensors="pt")                                                       # models.py:119-121
    assert enc.input_ids.shape[0] == 1 and enc.input_ids[0, 0] == 1 and enc.input_ids[0, 1:].tolist() == ids[:k]
    assert all(i < 400 for i in tok.encode("a < b and <imgx> stay text", add_special_tokens=False))


def test_tokenizer_decode_and_scrub(tok):
    cap = "George climbed a tree"
    gen = tok.sp_model.EncodeAsIds(cap) + list(range(400, 466)) + [2]
    text = tok.decode(gen, skip_special_tokens=False)                                       # models.py:156
    assert text.startswith(cap + " <img> <img_00000> <img_00001>") and text.endswith("<img_00063> </img> </s>")
    assert PromptStory.clean(text) == cap                                                   # gen_george.py:196
    assert tok.decode(torch.tensor(gen), skip_special_tokens=True).endswith("</img>")
    assert tok.decode(tok.encode(cap, add_special_tokens=False)) == cap


def _out(tok, caption, feat_val):
    ids = tok.encode(caption, add_special_tokens=False) + list(range(400, 466))
    return {"text": tok.decode(ids), "generate_ids": torch.tensor(ids), "img_gen_feat": torch.full((1, 4, 8), float(feat_val))}


def test_prompt_story_is_the_drivers_string_surgery(tok):
    q = "What happens next in the story?"
    caps = ["George climbed a tree.", "The man smiled and opened the door.", "They walked to the city zoo."]
    ps = PromptStory(tok, window=2)
    ps.start(q, torch.zeros(1, 4, 8))
    assert ps.prompt == q + IMG                                                             # :168-170
    assert ps.ids == [1] + tok.encode(q + IMG, add_special_tokens=False)                    # :174-175
    assert ps.advance(_out(tok, caps[0], 1)) == 0
    assert ps.prompt == q + IMG + caps[0] + IMG and ps.image_embeds.shape[0] == 2           # :224,231
    assert ps.advance(_out(tok, caps[1], 2)) == 1                                           # 3 images > window 2
    # cut through the first </img> PLUS six characters of the caption behind it (:237): "George" is gone
    assert ps.prompt == caps[0][6:] + IMG + caps[1] + IMG
    assert ps.prompt.startswith(" climbed a tree.")
    assert ps.image_embeds[:, 0, 0].tolist() == [1.0, 2.0]
    assert ps.advance(_out(tok, caps[2], 3)) == 1
    assert ps.prompt == caps[1][6:] + IMG + caps[2] + IMG
    m, e = ps.masks("cpu")
    assert m.sum().item() == 2 * 64 and e.tolist() == [True, True] and m.shape[1] == len(ps.ids)
    ids = ps.ids
    assert ids[0] == 1 and ids.count(400) == 2 and ids.count(465) == 2
    assert not m[0, ids.index(400)] and m[0, ids.index(400) + 1] and not m[0, ids.index(465)]


def test_id_level_context_beside_the_string_one(tok):
    """Same story through both managers: same images in the window and same mask structure; the id-level context keeps
    the generated caption ids verbatim (so the previous step's KV rows stay valid), the string one re-tokenises and
    loses six characters per eviction — the token streams differ exactly there."""
    q = "What happens next in the story?"
    caps = ["George climbed a tree.", "The man smiled and opened the door.", "They walked to the city zoo."]
    img = list(range(400, 466))
    sc = StoryContext(tok.bos_token_id, 400, 465, img[1:-1], window=2)
    ps = PromptStory(tok, window=2)
    first = torch.zeros(1, 4, 8)
    sc.start(tok.encode(q, add_special_tokens=False), first)
    ps.start(q, first)
    assert sc.ids == ps.ids
    o = _out(tok, caps[0], 1)
    assert sc.advance(o) == 0 and ps.advance(o) == 0
    assert sc.ids == ps.ids                      # no eviction yet and this caption survives decode -> scrub -> encode
    o = _out(tok, caps[1], 2)
    assert sc.advance(o) == 1 and ps.advance(o) == 1
    assert torch.equal(sc.image_embeds, ps.image_embeds)
    cap0 = tok.encode(caps[0], add_special_tokens=False)
    assert sc.ids[:1 + len(cap0)] == [1] + cap0                          # whole caption kept
    assert ps.ids[1:1 + 3] != cap0[:3]                                   # "George" eaten by the '[INST]' skip
    assert sc.ids[-67 - len(tok.encode(caps[1], add_special_tokens=False)):] == ps.ids[-67 - len(tok.encode(caps[1], add_special_tokens=False)):]
    assert sc.masks("cpu")[0].sum() == ps.masks("cpu")[0].sum() == 128


def test_synthetic_tokenizer_round_trip():
    from src.inference.gen_george import SyntheticTokenizer
    t = SyntheticTokenizer(1066)
    ids = [17, 530] + t.img + [99]
    s = t.decode(ids)
    assert t.encode(s) == ids
    assert t.encode("hello world" + BOI_TOKEN)[-1] == t.img[0] and len(t.encode("hello world")) == 2
    assert t.encode("hello") == t.encode("hello")


# ---- pin against the INSTALLED transformers (5.15; the reference pins 4.34 — gap stated below) ---------------------------

@pytest.fixture(scope="module")
def llama_like(tmp_path_factory):
    """A sentencepiece model with LLaMA's normaliser settings (identity rule, add_dummy_prefix, NO whitespace removal,
    byte fallback, split digits), the 66 added image tokens, and both tokenizers on it."""
    import json
    import sentencepiece as spm
    transformers = pytest.importorskip("transformers")
    d = tmp_path_factory.mktemp("tokpin")
    open(d / "corpus.txt", "w").write("\n".join((CORPUS + ["[INST] Generate the next scene. [/INST]"]) * 40))
    spm.SentencePieceTrainer.train(input=str(d / "corpus.txt"), model_prefix=str(d / "tokenizer"), vocab_size=400,
                                   model_type="bpe", bos_id=1, eos_id=2, unk_id=0, pad_id=-1, byte_fallback=True,
                                   character_coverage=1.0, add_dummy_prefix=True, normalization_rule_name="identity",
                                   remove_extra_whitespaces=False, split_digits=True, minloglevel=2)
    hf = transformers.LlamaTokenizer.from_pretrained(str(d), legacy=True)
    n0 = len(hf)
    assert hf.add_tokens(image_token_strings(), special_tokens=False) == 66
    json.dump({t: n0 + i for i, t in enumerate(image_token_strings())}, open(d / "added_tokens.json", "w"))
    return hf, LlamaTokenizer.from_pretrained(str(d)), transformers.__version__


def test_tokenizer_pinned_on_installed_transformers(llama_like):
    """``seedstory.tokenizer`` vs ``transformers.LlamaTokenizer`` (legacy=True) on the SAME sentencepiece model + added
    tokens, for the strings the drivers really encode (gen_george.py:168-176,231-239: question + image tokens, then
    ``prompt + text + image_tokens`` with ``text`` regex-scrubbed and stripped, then the window cut).

    Version gap, stated: the reference pins transformers 4.34 whose LlamaTokenizer is the *slow* sentencepiece class; the
    installed 5.15 converts the same model to a `tokenizers` pipeline.  They agree — and this test pins — every id of the
    driver-shaped strings.  They differ in two places, asserted below as the literal 4.34-slow behaviour the product
    follows: (a) a text segment that STARTS WITH A SPACE (``</img> y``): 4.34-slow hands the segment to sentencepiece,
    whose dummy prefix makes it ``▁ ▁y``; 5.15 emits ``▁y``;  (b) ``decode``: 4.34-slow joins added tokens and text runs
    with single spaces (``spaces_between_special_tokens=True``), 5.15 does not."""
    hf, mine, ver = llama_like
    img = "".join(image_token_strings())
    q = "George the monkey looked at the man with the yellow hat."
    t1, t2 = "The man smiled and opened the door.", "They walked to the city zoo!"
    prompts = [q + img, q + img + t1 + img, q + img + t1 + img + t2 + img, "[INST] What happens next? [/INST]" + img,
               (q + img + t1 + img)[len(q + img) + len("[INST]"):],          # the window cut of gen_george.py:236-237
               "George", "What happens next in the story?", img, "<img>", "</img>", "a\nb", "x</img>y", "Ünï ☃ 12345"]
    for p in prompts:
        a, b = hf.encode(p, add_special_tokens=False), mine.encode(p, add_special_tokens=False)
        assert a == b, (ver, p[:40], a[:12], b[:12])
    assert hf.bos_token_id == mine.bos_token_id == 1 and hf.eos_token_id == mine.eos_token_id == 2
    assert hf.encode("<img>", add_special_tokens=False) == mine.encode("<img>", add_special_tokens=False) == [len(hf) - 66]
    assert hf.encode(img, add_special_tokens=False) == list(range(len(hf) - 66, len(hf)))
    # (a) leading-space segment: 4.34-slow semantics (sentencepiece per segment, dummy prefix kept)
    sp = mine.sp_model
    assert mine.encode("</img> y", add_special_tokens=False) == [len(hf) - 1] + sp.EncodeAsIds(" y")
    assert sp.EncodeAsIds(" y")[0] == sp.PieceToId("▁") and len(sp.EncodeAsIds(" y")) == len(sp.EncodeAsIds("y")) + 1
    # (b) decode spacing of 4.34-slow; the drivers scrub it with re.sub(r'\s*<[^>]*>\s*', ' ', .) (gen_george.py:196)
    ids = mine.encode("He ran.<img><img_00001></img>ok", add_special_tokens=False)
    assert mine.decode(ids) == "He ran. <img> <img_00001> </img> ok"
    import re
    assert re.sub(r"\s*<[^>]*>\s*", " ", mine.decode(ids)).strip() == re.sub(r"\s*<[^>]*>\s*", " ", hf.decode(ids)).strip()
