"""TEST INFRASTRUCTURE — generates tests/golden/prepare_inputs.json by calling the REAL reference
``LlamaForCausalLM.prepare_inputs_for_generation`` (src/models_clm/modeling_llama_xformer.py:796-852, imported from
/root/reference behind oracle/ref_shims.py) as an unbound function on a stand-in ``self`` carrying the three attributes it
reads (``use_kv_cache_head``, ``kv_cache_head``, ``training``).  Pure host logic (tensor slicing): the fixture pins the
mirror's restatement on every branch.  Run here only (the reference tree does not exist on the GPU box):

    python oracle/make_golden_prepare_inputs.py
"""
import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

CASES = []
for use_head in (True, False):
    for head in (None, 5, 9):
        for past in (False, True):
            for with_embeds in (False, True):
                for with_mask in (False, True):
                    for S in (9, 12):
                        if head is not None and head > S:
                            continue
                        if use_head and past and head is None:
                            continue      # the reference would slice with None: not a state the drivers produce
                        CASES.append(dict(use_head=use_head, head=head, past=past, with_embeds=with_embeds, with_mask=with_mask, S=S))


def build_args(c):
    S = c["S"]
    ids = torch.arange(100, 100 + S).unsqueeze(0)
    emb = (torch.arange(S * 4, dtype=torch.float32).reshape(1, S, 4) if c["with_embeds"] else None)
    mask = torch.ones(1, S, dtype=torch.long) if c["with_mask"] else None
    past = ((torch.zeros(1, 1, 3, 2), torch.zeros(1, 1, 3, 2)),) if c["past"] else None
    return ids, past, mask, emb


def encode(out):
    enc = {}
    for k, v in out.items():
        if isinstance(v, torch.Tensor):
            enc[k] = {"shape": list(v.shape), "values": v.flatten().tolist()}
        elif k == "past_key_values":
            enc[k] = None if v is None else "past"
        else:
            enc[k] = v
    return enc


def main():
    llama_mod, _, _, _ = ref_shims.import_reference()
    fn = llama_mod.LlamaForCausalLM.prepare_inputs_for_generation
    res = []
    for c in CASES:
        me = types.SimpleNamespace(use_kv_cache_head=c["use_head"], kv_cache_head=c["head"], training=False)
        ids, past, mask, emb = build_args(c)
        out = fn(me, ids, past_key_values=past, attention_mask=mask, inputs_embeds=emb, use_cache=True)
        res.append({"case": c, "out": encode(out)})
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "prepare_inputs.json")
    json.dump({"source": "reference prepare_inputs_for_generation, modeling_llama_xformer.py:796-852", "cases": res}, open(path, "w"))
    print("wrote %d cases -> %s (%.1f KiB)" % (len(res), path, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
