"""Cached CPU-oracle truths for the full-size GPU tests (keep `pytest -m gpu` well under its wall-clock limit).

The full-size tests compare the HIP path with the oracle run on the host on the same seeded weights; those host runs (LLaMA-2-7B
32 layers in fp32 and bf16, ViT-G 48 blocks, the 3-step story) cost minutes per suite.  Their OUTPUTS are stored under
tests/golden/*.safetensors, keyed on a checksum of everything they were computed from; a box whose seeded generators draw the same
weights loads them, any other box recomputes them with the oracle exactly as before (and, with SS_WRITE_GOLDEN_DIR=<dir>, writes a
fresh file — oracle/make_golden_mllm_full.py is the committed recipe).  Test infrastructure only: nothing under seed-story_amd/
imports this."""
import hashlib
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def tensors_key(*groups):
    """sha256 over a cheap fingerprint of every tensor of every group (dicts are walked in sorted key order; ints / lists of ints /
    strings are hashed as text): name, shape, dtype, the first 64 elements and a strided sample of ~4096 elements, as float32
    bytes.  Any difference in how a box draws the seeded weights changes the key."""
    h = hashlib.sha256()

    def one(name, t):
        h.update(name.encode())
        if torch.is_tensor(t):
            f = t.detach().flatten()
            h.update(("%s|%s|" % (tuple(t.shape), t.dtype)).encode())
            h.update(f[:64].float().cpu().numpy().tobytes())
            h.update(f[:: max(1, f.numel() // 4096)].float().cpu().numpy().tobytes())
        else:
            h.update(repr(t).encode())

    for gi, g in enumerate(groups):
        if isinstance(g, dict):
            for k in sorted(g):
                one("%d/%s" % (gi, k), g[k])
        else:
            one("%d" % gi, g)
    return h.hexdigest()


def load_or_compute(name, key, compute, note=""):
    """-> (dict of tensors, "loaded" | "computed").  `compute()` returns the dict; every value a tensor."""
    from safetensors import safe_open
    from safetensors.torch import save_file
    path = os.path.join(GOLDEN, name + ".safetensors")
    if os.path.exists(path) and not os.environ.get("SS_IGNORE_TRUTH_CACHE"):
        with safe_open(path, "pt") as f:
            if (f.metadata() or {}).get("key") == key:
                return {k: f.get_tensor(k) for k in f.keys()}, "loaded"
    out = compute()
    dst = os.environ.get("SS_WRITE_GOLDEN_DIR")
    if dst:
        os.makedirs(dst, exist_ok=True)
        save_file({k: v.detach().contiguous().cpu() for k, v in out.items()}, os.path.join(dst, name + ".safetensors"),
                  metadata={"key": key, "generator": "oracle/make_golden_mllm_full.py (tests/test_fulldim_gpu.py, oracle on the host)", "note": note})
    return out, "computed"
