"""Drop-in for the reference's ``src/models_clm/peft_models.py`` (peft==0.4.0 is absent here).

``get_peft_model_with_resize_embedding(model, peft_config, vocab_size, torch_dtype)`` keeps the
reference signature (reference :21-66) and returns a wrapper with the attribute / state-dict
layout of ``PeftModelForCausalLM`` as the agent checkpoints expect it:

    base_model.model.model.layers.N.self_attn.q_proj.weight
    base_model.model.model.layers.N.self_attn.q_proj.lora_A.default.weight   [r, in]
    base_model.model.model.layers.N.self_attn.q_proj.lora_B.default.weight   [out, r]
    base_model.model.model.layers.N.input_layernorm.original_module.weight
    base_model.model.model.layers.N.input_layernorm.modules_to_save.default.weight

LoRA semantics (configs/clm_models/llama2chat7b_lora.yaml:7-26): y = Wx + (alpha/r) B A x on
q,k,v,o,gate,up,down; dropout inert in eval.  On the MI355X path the factors are merged into the
weights once, in fp32 on the device, when the engine is (re)built — decode streams 13.2 GB of
weights per token, the 14 extra rank-16 GEMVs per layer of the unmerged form are pure overhead.
"""
import copy

import torch
from torch import nn

from seedstory import instantiate as _inst

_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16}


class _Holder(nn.Module):
    def __init__(self, o, i, dtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(o, i, dtype=dtype, device=device), requires_grad=False)


class _ModulesToSave(nn.Module):
    """peft ModulesToSaveWrapper naming: original_module + modules_to_save['default']."""

    def __init__(self, original):
        super().__init__()
        self.original_module = original
        self.modules_to_save = nn.ModuleDict({"default": copy.deepcopy(original)})     # peft: deepcopy of the module

    @property
    def weight(self):
        return self.modules_to_save["default"].weight


class _LoraModel(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.model, name)


class PeftModelForCausalLM(nn.Module):
    def __init__(self, model, peft_config):
        super().__init__()
        r = int(peft_config.get("r", 16))
        alpha = float(peft_config.get("lora_alpha", 32))
        targets = list(peft_config.get("target_modules", []))
        to_save = list(peft_config.get("modules_to_save", []) or [])
        model._lora_scaling = alpha / r
        for mod_name, mod in list(model.named_modules()):
            leaf = mod_name.split(".")[-1]
            if leaf in targets and hasattr(mod, "weight") and mod.weight.dim() == 2:
                o, i = mod.weight.shape
                dt, dev = mod.weight.dtype, mod.weight.device
                mod.lora_A = nn.ModuleDict({"default": _Holder(r, i, dt, dev)})
                mod.lora_B = nn.ModuleDict({"default": _Holder(o, r, dt, dev)})  # zeros: peft's init
        for mod_name, mod in list(model.named_modules()):
            for child_name, child in list(mod.named_children()):
                if child_name in to_save and not isinstance(child, _ModulesToSave):
                    setattr(mod, child_name, _ModulesToSave(child))
        self.base_model = _LoraModel(model)
        self.peft_config = {"default": dict(peft_config)}
        model._engine = None

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.base_model, name)

    def get_input_embeddings(self):
        return self.base_model.model.get_input_embeddings()

    def get_output_embeddings(self):
        return self.base_model.model.get_output_embeddings()

    def generate(self, **kwargs):
        return self.base_model.model.generate(**kwargs)

    def print_trainable_parameters(self):
        n = sum(p.numel() for k, p in self.named_parameters() if "lora_" in k or "modules_to_save" in k)
        print("adapter params: %d" % n)


def get_peft_model_with_resize_embedding(model, peft_config=None, model_id=None, vocab_size=None,
                                         torch_dtype="bf16"):
    torch_dtype = _DTYPES.get(torch_dtype, torch_dtype if isinstance(torch_dtype, torch.dtype) else torch.float32)
    if isinstance(model, dict) and "_target_" in model:
        model = _inst.instantiate(model, torch_dtype=torch_dtype)
    assert (peft_config is None) + (model_id is None) == 1
    if vocab_size is not None:
        print(f"Length of tokenizer and resize embedding: {vocab_size}")
        model.resize_token_embeddings(vocab_size)
    if peft_config is None:
        return PeftModel.from_pretrained(model=model, model_id=model_id)
    cfg = {k: v for k, v in dict(peft_config).items() if not k.startswith("_")}
    return PeftModelForCausalLM(model, cfg)


class PeftModel:
    """``PeftModel.from_pretrained(model, model_id)`` of peft==0.4.0 for a LOCAL adapter folder (reference
    peft_models.py:63): adapter_config.json gives r / lora_alpha / target_modules / modules_to_save, adapter_model.bin the
    factors with the adapter name stripped from the keys."""

    @staticmethod
    def from_pretrained(model, model_id, adapter_name="default", **kwargs):
        from seedstory import ckpt as _ckpt
        cfg, sd = _ckpt.read_peft_adapter(model_id)
        if cfg.get("peft_type", "LORA") != "LORA":
            raise ValueError("only LoRA adapters are on this path, got %r" % cfg.get("peft_type"))
        pm = PeftModelForCausalLM(model, cfg)
        missing, unexpected = pm.load_state_dict(sd, strict=False)
        # an adapter file holds ONLY adapter tensors: base weights are "missing" by construction
        missing = [k for k in missing if "lora_" in k or "modules_to_save" in k]
        print("peft adapter, missing keys: ", len(missing), "unexpected keys:", len(unexpected))
        pm.load_report = {"missing": missing, "unexpected": list(unexpected)}
        if missing or unexpected:
            raise KeyError("adapter %s does not match the model: missing %s unexpected %s"
                           % (model_id, missing[:4], list(unexpected)[:4]))
        model._engine = None
        return pm


def get_model_with_resize_embedding(model, vocab_size=None, torch_dtype="bf16"):
    torch_dtype = _DTYPES.get(torch_dtype, torch.float32)
    if isinstance(model, dict) and "_target_" in model:
        model = _inst.instantiate(model, torch_dtype=torch_dtype)
    if vocab_size is not None:
        model.resize_token_embeddings(vocab_size)
    return model
