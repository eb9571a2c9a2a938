"""Golden vectors for the diffusers attention-processor plug point, produced by the REAL reference classes.

    python oracle/make_golden_attnproc.py        (build container only: reads /root/reference)

Runs ``AttnProcessor`` and ``AttnProcessor2_0`` of ``/root/reference/src/models_ipa/attention_processor.py`` on the stand-in
``Attention`` modules of ``oracle/diffusers_standin.py`` (seeded weights and inputs), asserts that the two reference classes agree
with each other, and stores their outputs (fp32 and bf16 CPU runs) in ``tests/golden/attn_processor.safetensors``.  The product's
``src.models_ipa.attention_processor.AttnProcessor`` is compared with these rows on the GPU (``tests/test_attn_processor.py``);
weights and inputs regenerate from the seeds, only outputs are stored.
"""
import importlib.util
import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import diffusers_standin as S  # noqa: E402

REF = os.path.join(os.environ.get("SEEDSTORY_REFERENCE_ROOT", "/root/reference"), "src", "models_ipa", "attention_processor.py")


def main():
    spec = importlib.util.spec_from_file_location("ref_attention_processor", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {}
    for name in S.CASES:
        m, x, e = S.build(name)
        with torch.no_grad():
            y1 = ref.AttnProcessor()(m, x, encoder_hidden_states=e)
            y2 = ref.AttnProcessor2_0()(m, x, encoder_hidden_states=e)
            d12 = float((y1 - y2).norm() / y1.norm())
            assert d12 < 2e-6, (name, d12)
            mb = S.build(name)[0].to(torch.bfloat16)
            yb = ref.AttnProcessor2_0()(mb, x.to(torch.bfloat16), encoder_hidden_states=None if e is None else e.to(torch.bfloat16))
        rows = slice(None, None, 4) if name.startswith("sdxl") else slice(None)     # every 4th token row of the big cases
        out[name + ".fp32"] = y1[:, rows].contiguous()
        out[name + ".bf16"] = yb[:, rows].contiguous()
        print("%-28s out %s  |AttnProcessor - AttnProcessor2_0| %.1e   bf16 vs fp32 %.2e" %
              (name, tuple(y1.shape), d12, float((yb.float() - y1).norm() / y1.norm())))
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "attn_processor.safetensors")
    save_file(out, path, metadata={"generator": "oracle/make_golden_attnproc.py", "torch": torch.__version__})
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
