"""Writes the cached CPU-oracle truths of the full-size MLLM-half GPU tests (test infrastructure; see tests/truth_cache.py):

    tests/golden/vitg48_truth.safetensors    Qwen ViT-G, 48 blocks + attn_pool, fp32 and bf16 oracle runs      (test_vit_g_all_48_blocks)
    tests/golden/llama7b_truth.safetensors   LLaMA-2-7B, 32 layers: prefill + continuation + decode rows        (test_llama_7b_all_32_layers)
    tests/golden/story3_truth.safetensors    3 story steps of ContinuousLVLM.generate semantics at real size    (test_story_three_steps_...)

The computations are the ORACLE SIDE of those tests, unchanged (`tests/test_fulldim_gpu.py::{vitg48_truth, llama7b_truth,
story3_truth}` — O.vit_forward / O.llama_forward / O.resampler_forward on seeded weights); no GPU and no reference tree are needed.
Each file is keyed on a fingerprint of the weights and ids it was computed from: a box that draws the same seeded weights loads
it, any other box recomputes the truth on its host as before.  One part per process (the 7B weights take 40 GB of host memory):

    python oracle/make_golden_mllm_full.py            # all three, ~7 min on 8 cores (62 GB of host memory is enough)
    python oracle/make_golden_mllm_full.py llama7b    # one part
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARTS = {"vitg48": "vitg48_truth", "llama7b": "llama7b_truth", "story3": "story3_truth"}


def one(part):
    for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "seed-story_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["SS_WRITE_GOLDEN_DIR"] = os.path.join(ROOT, "tests", "golden")
    os.environ["SS_IGNORE_TRUTH_CACHE"] = "1"            # recompute even when a matching file exists
    import test_fulldim_gpu as T
    t0 = time.time()
    out = getattr(T, PARTS[part])()
    assert out[-1] == "computed"
    print("%s: computed and written in %.0f s" % (PARTS[part], time.time() - t0), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2])
    else:
        for part in (sys.argv[1:] or list(PARTS)):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--one", part])
