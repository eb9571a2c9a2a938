"""Minimal ``hydra.utils.instantiate`` / ``OmegaConf.load`` stand-in on PyYAML.

The reference instantiates every object from a one-file-per-object YAML with a ``_target_``
dotted path and constructor overrides (src/inference/gen_george.py:39-71); hydra-core and
omegaconf are not installed in this image, and this ~60-line resolver is all the path needs.
Nested dicts carrying ``_target_`` are instantiated depth-first unless the callee wants the raw
config (``_partial_``/``_convert_`` keys are accepted and ignored like hydra's defaults).
"""
import importlib

import yaml


class DictConfig(dict):
    """Marker type: a config node (what omegaconf would hand to the callee)."""


def load(path):
    with open(path) as f:
        return _wrap(yaml.safe_load(f))


def _wrap(node):
    if isinstance(node, dict):
        return DictConfig({k: _wrap(v) for k, v in node.items()})
    if isinstance(node, list):
        return [_wrap(v) for v in node]
    return node


# third-party targets of the reference YAMLs that this package provides itself (see the modules' headers for why)
_ALIASES = {"transformers.LlamaTokenizer.from_pretrained": "seedstory.tokenizer.LlamaTokenizer.from_pretrained"}


def locate(path):
    path = _ALIASES.get(path, path)
    parts = path.split(".")
    for i in range(len(parts), 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:i]))
        except ImportError:
            continue
        for name in parts[i:]:
            obj = getattr(obj, name)
        return obj
    raise ImportError("cannot locate %r" % path)


# callables that take raw config nodes for these keyword names (they instantiate lazily themselves,
# like the reference's get_peft_model_with_resize_embedding does with `model`, peft_models.py:36-37)
_RAW_KWARGS = {"src.models_clm.peft_models.get_peft_model_with_resize_embedding": {"model", "peft_config"}}


def instantiate(cfg, *args, **overrides):
    if not isinstance(cfg, dict) or "_target_" not in cfg:
        raise TypeError("instantiate() needs a config with a _target_")
    target = cfg["_target_"]
    raw = _RAW_KWARGS.get(target, set())
    kwargs = {}
    for k, v in cfg.items():
        if k in ("_target_", "_convert_", "_partial_", "_recursive_"):
            continue
        if isinstance(v, dict) and "_target_" in v and k not in raw:
            v = instantiate(v)
        kwargs[k] = v
    kwargs.update(overrides)
    return locate(target)(*args, **kwargs)
