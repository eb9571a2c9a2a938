"""TEST INFRASTRUCTURE ONLY — pins the oracle's greedy loop (``seedstory_oracle.greedy_generate``, a restatement of HF
greedy search as ``ContinuousLVLM.generate`` drives it, src/models_clm/models.py:137-153) on a
REAL ``GenerationMixin.generate`` run.

    python oracle/make_golden_greedy.py          (needs /root/reference for the reference's logits processor)

The reference pins transformers==4.34.0, which is not installable here; the image has transformers 5.15.  What this
script runs is therefore the INSTALLED transformers' ``generate(do_sample=False, output_hidden_states=True,
return_dict_in_generate=True)`` — the same call, same keyword set as models.py:142-153 — on a stock HF
``LlamaForCausalLM`` (tiny config, seeded weights of oracle/synth.py, eager attention) with the reference's REAL
``AutoImageTokenGenerationProcessor`` (src/models_clm/generation.py:9-31) in the ``logits_processor`` list, with BOTH
``input_ids`` and ``inputs_embeds`` supplied as the reference does.  Version gap stated: 5.15 vs 4.34 — the greedy
search contract exercised here (embeds used for the first forward only, ids kept as the running sequence, processor
applied to the last-row scores before argmax, stop on EOS or ``max_new_tokens``, ``hidden_states`` a per-step tuple
whose first element covers the prompt rows) is the part of the API that did not change between them.

Pinned: generated ids (65 processor-forced tokens behind ``<img>`` + a free-running tail), the stop rule (a second run
whose EOS id is a token of the free tail stops at its first occurrence), the per-step hidden-state tuple shapes, and the last-layer rows
(transformers 5.x records the LAST tuple element after the final norm, like modeling_llama_xformer.py:652-656).
Writes ``tests/golden/greedy_hf.safetensors``.
"""
import json
import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402
import seedstory_oracle as O  # noqa: E402
import synth  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
LLAMA = dict(hidden=256, n_heads=2, n_layers=2, inter=512, vocab=320)
IMG_IDS = list(range(320 - 66, 320))
MAX_NEW = 80


class _FakeTok:
    def encode(self, s, add_special_tokens=False):
        return list(IMG_IDS)


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def main():
    import transformers
    from transformers import LlamaConfig, LlamaForCausalLM, LogitsProcessorList
    _, _, gen_mod, _ = ref_shims.import_reference()
    d = LLAMA
    cfg = LlamaConfig(hidden_size=d["hidden"], intermediate_size=d["inter"], num_hidden_layers=d["n_layers"],
                      num_attention_heads=d["n_heads"], num_key_value_heads=d["n_heads"], vocab_size=d["vocab"],
                      max_position_embeddings=4096, rms_norm_eps=1e-5, rope_theta=10000.0, attention_bias=False,
                      tie_word_embeddings=False, bos_token_id=1, eos_token_id=2, pad_token_id=None)
    cfg._attn_implementation = "eager"
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    m = LlamaForCausalLM(cfg).eval()
    missing, unexpected = m.load_state_dict(wd, strict=False)
    assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    dims = O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    proc = gen_mod.AutoImageTokenGenerationProcessor(tokenizer=_FakeTok(), num_img_gen_tokens=64)
    # prompt ends with <img>: the processor forces <img_00000> ... <img_00063> </img>, then the model runs free
    prompt = [1] + synth.randint(60, (11,), 3, 250).tolist() + [IMG_IDS[0]]
    input_ids = torch.tensor([prompt])
    # inputs_embeds differ from embed(input_ids) on 4 rows (as after the image-feature splice, models.py:135): proves
    # that the first forward consumes the EMBEDS while the ids only carry the sequence
    emb = wd["model.embed_tokens.weight"][input_ids].clone()
    emb[0, 3:7] = synth.normal_like(61, (4, d["hidden"]), 0.02)
    out = {}
    meta = dict(LLAMA=LLAMA, IMG_IDS=[IMG_IDS[0], IMG_IDS[-1]], MAX_NEW=MAX_NEW, transformers=transformers.__version__,
                reference_pins="transformers==4.34.0 (absent)", torch=torch.__version__)
    for tag, eos in (("free", 2), ("eos", None)):
        if eos is None:
            eos = int(out["free.generate_ids"][65 + 4])          # the free tail's 5th token becomes EOS: must stop there
            meta["eos_case_id"] = eos
        with torch.no_grad():
            r = m.generate(input_ids=input_ids, inputs_embeds=emb.clone(), output_hidden_states=True,
                           return_dict_in_generate=True, logits_processor=LogitsProcessorList([proc]),
                           max_new_tokens=MAX_NEW, do_sample=False, num_beams=1, eos_token_id=eos, pad_token_id=0)
        seq = r.sequences[0].tolist()
        # sequences carry the prompt ids when input_ids is supplied next to inputs_embeds (models.py:158 slices them off)
        gen = seq[len(prompt):] if seq[:len(prompt)] == prompt else seq
        hs = r.hidden_states
        assert len(hs) == len(gen), (len(hs), len(gen))
        assert len(hs[0]) == d["n_layers"] + 1
        assert hs[0][-1].shape == (1, len(prompt), d["hidden"]) and all(h[-1].shape == (1, 1, d["hidden"]) for h in hs[1:])
        last = torch.cat([h[-1] for h in hs], dim=1)[0, len(prompt):]           # models.py:182-184
        mine_gen, mine_hid, _, _ = O.greedy_generate(wd, dims, input_ids, emb.clone(), IMG_IDS, MAX_NEW, eos_id=eos)
        assert mine_gen == gen, (tag, mine_gen, gen)
        assert gen[:65] == IMG_IDS[1:], "processor-forced chain"
        e = rel(mine_hid, last)
        print("%s: %d tokens (stop: %s), oracle ids == HF ids, hidden rows rel %.3e" %
              (tag, len(gen), "EOS" if gen[-1] == eos else "max_new_tokens", e))
        assert e < 2e-6
        if tag == "eos":       # stops right after the FIRST occurrence of the EOS id in the free run's sequence
            free = out["free.generate_ids"].tolist()
            assert gen[-1] == eos and len(gen) == free.index(eos) + 1 and gen == free[:len(gen)]
        else:
            assert len(gen) == MAX_NEW
        out[tag + ".generate_ids"] = torch.tensor(gen)
        out[tag + ".hidden"] = last.float().contiguous()
    out["input_ids"] = input_ids
    out["inputs_embeds"] = emb
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLD, "greedy_hf.safetensors"))
    with open(os.path.join(GOLD, "greedy_hf.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote greedy_hf.safetensors (%d tensors)" % len(out))


if __name__ == "__main__":
    main()
