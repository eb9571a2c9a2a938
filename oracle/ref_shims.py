"""TEST INFRASTRUCTURE ONLY — not shipped, never imported by the product path.

Import shims that let the *real* SEED-Story reference modules under
``/root/reference`` run on this CPU-only container, so the oracle restatements
in this directory can be pinned against them and golden vectors generated
(see ``oracle/make_golden.py``).  Nothing here is used on the GPU box:
``/root/reference`` does not exist there.

Two third-party packages the reference imports are absent:

* ``xformers.ops`` — used at ``src/models_clm/modeling_llama_xformer.py:44,281-295``
  for ``memory_efficient_attention`` with ``LowerTriangularMask`` /
  ``LowerTriangularFromBottomRightMask``.  Stubbed with exact fp32-softmax
  attention (xformers==0.0.23.post1 published semantics: query i attends keys
  ``j <= i + (kv_len - q_len)`` for the bottom-right mask).
* ``torchvision.transforms`` — imported at ``src/models/qwen_visual.py:19-20`` only
  to build an ``image_transform`` attribute that the forward path never touches.
"""
import math
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("SEEDSTORY_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "models_clm"))


class _LowerTriangularMask:
    pass


class _LowerTriangularFromBottomRightMask:
    pass


def _memory_efficient_attention(query, key, value, attn_bias=None, p=0.0, scale=None):
    """q [B,M,H,K], k/v [B,N,H,K] -> [B,M,H,K]; softmax in fp32, output in q.dtype."""
    B, M, H, K = query.shape
    N = key.shape[1]
    scale = (1.0 / math.sqrt(K)) if scale is None else scale
    q = query.permute(0, 2, 1, 3).float()
    k = key.permute(0, 2, 1, 3).float()
    v = value.permute(0, 2, 1, 3).float()
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if isinstance(attn_bias, _LowerTriangularFromBottomRightMask):
        allow = torch.ones(M, N, dtype=torch.bool).tril(diagonal=N - M)
        s = s.masked_fill(~allow, float("-inf"))
    elif isinstance(attn_bias, _LowerTriangularMask):
        allow = torch.ones(M, N, dtype=torch.bool).tril(diagonal=0)
        s = s.masked_fill(~allow, float("-inf"))
    elif attn_bias is not None:
        raise NotImplementedError(type(attn_bias))
    p_ = torch.softmax(s, dim=-1)
    o = torch.matmul(p_, v)
    return o.permute(0, 2, 1, 3).to(query.dtype)


def install() -> None:
    """Install the stub modules and put the reference root on sys.path (idempotent)."""
    if "xformers" not in sys.modules:
        xf = types.ModuleType("xformers")
        ops = types.ModuleType("xformers.ops")
        fmha = types.ModuleType("xformers.ops.fmha")
        attn_bias = types.ModuleType("xformers.ops.fmha.attn_bias")
        attn_bias.LowerTriangularMask = _LowerTriangularMask
        attn_bias.LowerTriangularFromBottomRightMask = _LowerTriangularFromBottomRightMask
        fmha.attn_bias = attn_bias
        ops.fmha = fmha
        ops.LowerTriangularMask = _LowerTriangularMask
        ops.memory_efficient_attention = _memory_efficient_attention
        xf.ops = ops
        sys.modules["xformers"] = xf
        sys.modules["xformers.ops"] = ops
        sys.modules["xformers.ops.fmha"] = fmha
        sys.modules["xformers.ops.fmha.attn_bias"] = attn_bias
    if "torchvision" not in sys.modules:
        # transformers probes `importlib.util.find_spec("torchvision")` at import time and chokes on
        # a spec-less stub, so make sure it is imported BEFORE the stub goes in.
        import transformers  # noqa: F401
        import transformers.modeling_utils  # noqa: F401
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")

        class _Passthrough:
            def __init__(self, *a, **k):
                pass

            def __call__(self, x):
                return x

        class _InterpolationMode:
            BICUBIC = "bicubic"
            BILINEAR = "bilinear"

        for name in ("Compose", "Resize", "ToTensor", "Normalize", "CenterCrop"):
            setattr(tr, name, _Passthrough)
        tr.InterpolationMode = _InterpolationMode
        tv.transforms = tr
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tr
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def import_reference():
    """Returns (modeling_llama_xformer, qwen_visual, generation, ipa_resampler) reference modules."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    install()
    # The reference's top-level package is called `src`; our drop-in mirror uses the same
    # dotted names, so make sure the reference one wins inside this process.
    for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        mod = sys.modules[k]
        f = getattr(mod, "__file__", None) or ""
        if REFERENCE_ROOT not in f:
            del sys.modules[k]
    import importlib
    llama = importlib.import_module("src.models_clm.modeling_llama_xformer")
    qwen = importlib.import_module("src.models.qwen_visual")
    gen = importlib.import_module("src.models_clm.generation")
    ipa = importlib.import_module("src.models_ipa.resampler")
    return llama, qwen, gen, ipa
