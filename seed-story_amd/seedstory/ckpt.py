"""Checkpoint file formats of the path — what is on disk and how it lands in the modules.

  agent ``pytorch_model.bin``      torch pickle; keys ``llm.base_model.model.model.layers.N.…`` (peft wrapper prefix),
                                   ``….q_proj.lora_A.default.weight`` / ``lora_B.default.weight``,
                                   ``….input_layernorm.original_module.weight`` + ``….modules_to_save.default.weight``,
                                   ``input_resampler.*``, ``output_resampler.*``            (reference models.py:223-230)
  ``qwen_vit_G.pt``                torch pickle of Qwen-VL's ``transformer.visual`` state dict    (qwen_visual.py:413-422)
  de-tokenizer ``pytorch_model.bin``  torch pickle; ``unet.*`` (diffusers names) + ``resampler.*``  (adapter_modules.py:350-357)
  SDXL-base diffusers folder       ``unet/``, ``vae/`` (config.json + diffusion_pytorch_model.safetensors), ``scheduler/``
  HF LLaMA folder                  config.json + (sharded) ``*.safetensors`` / ``pytorch_model-*.bin``
  peft adapter folder              adapter_config.json + adapter_model.bin (``.default`` AND ``modules_to_save.`` stripped
                                   from the keys by peft's save path)

The reference loads all of them with ``strict=False`` and prints only COUNTS, which hides key mismatches; `load_checked`
keeps that behaviour (same print) but records the names on the module (``model.load_report``) and warns with the first
few, so a wrong layout is visible.
"""
import json
import os
import warnings

import torch


def read_weights(path):
    """A state dict from one file (.safetensors / torch pickle) or every weight shard in a folder."""
    if os.path.isdir(path):
        sd = {}
        for fn in sorted(os.listdir(path)):
            if fn.endswith(".safetensors") or (fn.endswith(".bin") and fn.startswith(("pytorch_model", "adapter_model"))):
                sd.update(read_weights(os.path.join(path, fn)))
        return sd
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu")


def load_checked(model, state_dict, tag, expect_unexpected=()):
    """``load_state_dict(strict=False)`` + the reference's count print + a record of the names."""
    missing, unexpected = model.load_state_dict(state_dict, strict=False)
    unexpected = [k for k in unexpected if not k.startswith(tuple(expect_unexpected))] if expect_unexpected else list(unexpected)
    print("%s missing keys: " % tag, len(missing), "unexpected keys:", len(unexpected))
    model.load_report = {"missing": list(missing), "unexpected": list(unexpected)}
    if missing or unexpected:
        warnings.warn("%s: %d missing / %d unexpected keys, e.g. missing %s unexpected %s"
                      % (tag, len(missing), len(unexpected), list(missing)[:3], list(unexpected)[:3]))
    return model.load_report


def peft_adapter_to_wrapper_keys(sd, adapter_name="default", modules_to_save=None):
    """Keys of a SAVED peft==0.4.0 adapter back to the wrapper's live names, the way ``set_peft_model_state_dict`` does it.

    ``get_peft_model_state_dict`` strips BOTH ``modules_to_save.`` and ``.<adapter_name>`` on save, so on disk a LoRA factor
    is ``….q_proj.lora_A.weight`` and a ``modules_to_save`` copy is just ``….input_layernorm.weight``.  Loading re-inserts
    ``<module>.modules_to_save.<adapter>`` for the FIRST name of ``config.modules_to_save`` that occurs in the key
    (substring match, list order — ``norm`` is also a substring of ``input_layernorm``, which is why order matters) and
    ``.<adapter>`` behind ``lora_A`` / ``lora_B`` / ``lora_embedding_*``."""
    out = {}
    mts = list(modules_to_save or ())
    for k, v in sd.items():
        if mts and "modules_to_save" not in k and "lora_" not in k:
            for name in mts:
                if name in k:
                    k = k.replace(name, "%s.modules_to_save.%s" % (name, adapter_name))
                    break
        elif ".modules_to_save." in k and (".modules_to_save." + adapter_name + ".") not in k:
            k = k.replace(".modules_to_save.", ".modules_to_save." + adapter_name + ".")      # tolerated older layout
        if "lora_" in k:
            suffix = k.split("lora_")[1]
            if "." in suffix:
                rest = ".".join(suffix.split(".")[1:])
                if not rest.startswith(adapter_name + "."):
                    k = k[:len(k) - len(rest)] + adapter_name + "." + rest
            else:
                k = k + "." + adapter_name
        out[k] = v
    return out


def read_peft_adapter(folder):
    with open(os.path.join(folder, "adapter_config.json")) as f:
        cfg = json.load(f)
    return cfg, peft_adapter_to_wrapper_keys(read_weights(folder), modules_to_save=cfg.get("modules_to_save"))
