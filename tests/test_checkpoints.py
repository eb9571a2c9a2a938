"""SURVEY.md §8f row 1 — checkpoint / weight FILE formats of the path, loaded from disk in the layouts the reference
ships them in (synthetic tensors, real key layouts, written by this test into tmp_path):

  agent ``pytorch_model.bin``         models.py:223-230       llm.base_model.model.… + lora_A.default / modules_to_save
  peft adapter folder                 peft_models.py:63       adapter_config.json + adapter_model.bin (name-stripped keys)
  HF LLaMA folder                     llama2chat7b_lora.yaml  config.json + sharded safetensors (+ legacy inv_freq buffers)
  ``qwen_vit_G.pt``                   qwen_visual.py:413-422  Qwen ``transformer.visual`` state dict
  de-tokenizer ``pytorch_model.bin``  adapter_modules.py:350  unet.* + resampler.*
  SDXL-base diffusers folder          gen_george.py:40-47     unet/ vae/ scheduler/ with config.json

CPU tests: every tensor of the file lands in the module (no missing / unexpected key, values equal).  GPU tests: the
module loaded FROM THE FILE computes what the oracle computes from the same tensors (LoRA unmerged in the oracle, merged
at load on the device)."""
import json

import pytest
import torch

import sdxl_oracle as S
import seedstory_oracle as O
import synth

DEV = "cuda:0"
LORA_CFG = {"peft_type": "LORA", "r": 16, "lora_alpha": 32, "lora_dropout": 0.05, "task_type": "CAUSAL_LM",
            "target_modules": ["q_proj", "v_proj", "k_proj", "o_proj", "gate_proj", "down_proj", "up_proj"],
            "modules_to_save": ["input_layernorm", "post_attention_layernorm", "norm"]}
PFX = "base_model.model."


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _img_ids(meta):
    lo, hi = meta["IMG_IDS"]
    return list(range(lo, hi + 1))


class _Tok:
    def __init__(self, ids):
        self.ids = ids

    def encode(self, s, add_special_tokens=False):
        return [self.ids[0]] if s == "<img>" else [self.ids[-1]] if s == "</img>" else list(self.ids)

    def decode(self, ids, skip_special_tokens=False):
        return " ".join(str(int(i)) for i in ids)


def peft_layout(wd, adapter_name="default", with_base=True, with_original=True):
    """Oracle (HF-named, ``lora_A.weight``) llama weights -> live peft-wrapper key layout."""
    out = {}
    for k, v in wd.items():
        if ".lora_" in k:
            out[PFX + k.replace(".weight", ".%s.weight" % adapter_name)] = v
        elif k.endswith("layernorm.weight") or k == "model.norm.weight":
            base = PFX + k[:-len(".weight")]
            out[base + ".modules_to_save.%s.weight" % adapter_name] = v
            if with_original:      # the frozen pre-training norm: present in the file, NOT what the adapter computes with
                out[base + ".original_module.weight"] = torch.ones_like(v)
        elif with_base:
            out[PFX + k] = v
    return out


def _tiny_llama(meta, vocab=None):
    from src.models_clm.modeling_llama_xformer import LlamaConfig, LlamaForCausalLM
    d = meta["LLAMA"]
    cfg = LlamaConfig(hidden_size=d["hidden"], intermediate_size=d["inter"], num_hidden_layers=d["n_layers"],
                      num_attention_heads=d["n_heads"], vocab_size=vocab or d["vocab"])
    return LlamaForCausalLM(cfg), d


def _agent_from_file(meta, path):
    from src.models.qwen_visual import Resampler
    from src.models_clm.models import ContinuousLVLM
    from src.models_clm.peft_models import get_peft_model_with_resize_embedding
    llm, d = _tiny_llama(meta, vocab=d_vocab_before_resize(meta))
    pm = get_peft_model_with_resize_embedding(llm, peft_config=dict(LORA_CFG), vocab_size=d["vocab"], torch_dtype="fp32")
    rin = Resampler(grid_size=meta["RES_IN"]["grid"], embed_dim=256, num_heads=2, kv_dim=256)
    rout = Resampler(grid_size=meta["RES_OUT"]["grid"], embed_dim=256, num_heads=2, kv_dim=256)
    return ContinuousLVLM.from_pretrained(llm=pm, input_resampler=rin, output_resampler=rout, pretrained_model_path=path)


def d_vocab_before_resize(meta):
    return meta["LLAMA"]["vocab"] - 66          # the 66 added image tokens (llama2chat7b_lora.yaml:29)


def _agent_weights(meta):
    d = meta["LLAMA"]
    wd = synth.llama_weights(12, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], lora_r=16)
    wd.update(synth.resampler_weights(21, "input_resampler.", meta["RES_IN"]["grid"], 256))
    wd.update(synth.resampler_weights(22, "output_resampler.", meta["RES_OUT"]["grid"], 256))
    return wd


def _write_agent_file(meta, tmp_path):
    wd = _agent_weights(meta)
    llama = {k: v for k, v in wd.items() if not k.startswith(("input_resampler.", "output_resampler."))}
    ck = {"llm." + k: v for k, v in peft_layout(llama).items()}
    ck.update({k: v for k, v in wd.items() if k.startswith(("input_resampler.", "output_resampler."))})
    path = str(tmp_path / "pytorch_model.bin")
    torch.save(ck, path)
    return wd, ck, path


# ------------------------------------------------------------------------------------------------------------
# CPU: key layouts
# ------------------------------------------------------------------------------------------------------------


def test_agent_checkpoint_file_lands_completely(golden, tmp_path):
    g, meta = golden
    wd, ck, path = _write_agent_file(meta, tmp_path)
    agent = _agent_from_file(meta, path)
    assert agent.load_report == {"missing": [], "unexpected": []}
    sd = agent.state_dict()
    assert set(sd) == set(ck)
    for k in ck:
        assert torch.equal(sd[k], ck[k]), k
    # the flat view the engine is built from: adapter norms (not the frozen originals), LoRA factors by projection
    flat = agent.llm.base_model.model.collect_flat_state()
    assert torch.equal(flat["model.layers.1.input_layernorm.weight"], wd["model.layers.1.input_layernorm.weight"])
    assert torch.equal(flat["model.norm.weight"], wd["model.norm.weight"])
    assert torch.equal(flat["model.layers.0.mlp.up_proj.lora_B.default.weight"], wd["model.layers.0.mlp.up_proj.lora_B.weight"])
    assert not any("original_module" in k for k in flat)
    # a wrong layout is reported by name, not only counted
    bad = dict(ck)
    bad["llm.base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight"] = bad.pop(
        "llm.base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight")
    torch.save(bad, path)
    with pytest.warns(UserWarning, match="1 missing / 1 unexpected"):
        agent2 = _agent_from_file(meta, path)
    assert agent2.load_report["missing"] == ["llm.base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight"]


def test_modules_to_save_wrapper_is_a_deepcopy():
    """peft's ModulesToSaveWrapper deep-copies the wrapped module whatever its constructor looks like."""
    from torch import nn
    from src.models_clm.peft_models import _ModulesToSave

    class Odd(nn.Module):
        def __init__(self, a, b, *, flag):
            super().__init__()
            self.weight = nn.Parameter(torch.full((a, b), 3.0))
            self.flag = flag

    w = _ModulesToSave(Odd(2, 3, flag="x"))
    assert w.modules_to_save["default"].flag == "x" and torch.equal(w.weight, torch.full((2, 3), 3.0))
    assert w.weight.data_ptr() != w.original_module.weight.data_ptr()


def test_peft_adapter_folder(golden, tmp_path):
    """``get_peft_model_with_resize_embedding(model_id=folder)`` -> ``PeftModel.from_pretrained`` (peft_models.py:63)."""
    from src.models_clm.peft_models import get_peft_model_with_resize_embedding
    g, meta = golden
    d = meta["LLAMA"]
    wd = synth.llama_weights(12, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], lora_r=16)
    live = peft_layout(wd, with_base=False, with_original=False)
    # peft 0.4.0 get_peft_model_state_dict: `key.replace("modules_to_save.", "")` then `.replace(".default", "")`
    saved = {k.replace("modules_to_save.", "").replace(".default", ""): v for k, v in live.items()}
    assert any(k.endswith("input_layernorm.weight") for k in saved) and not any("modules_to_save" in k for k in saved)
    folder = tmp_path / "adapter"
    folder.mkdir()
    json.dump(LORA_CFG, open(folder / "adapter_config.json", "w"))
    torch.save(saved, str(folder / "adapter_model.bin"))
    llm, _ = _tiny_llama(meta)
    pm = get_peft_model_with_resize_embedding(llm, model_id=str(folder), torch_dtype="fp32")
    assert pm.load_report == {"missing": [], "unexpected": []}
    sd = pm.state_dict()
    for k, v in live.items():
        assert torch.equal(sd[k], v), k
    assert llm._lora_scaling == 2.0
    # an adapter for a different architecture is an error, not a silent partial load
    saved.pop(next(iter(saved)))
    torch.save(saved, str(folder / "adapter_model.bin"))
    with pytest.raises(KeyError):
        get_peft_model_with_resize_embedding(_tiny_llama(meta)[0], model_id=str(folder), torch_dtype="fp32")


def _write_hf_llama_folder(meta, tmp_path, wd):
    from safetensors.torch import save_file
    d = meta["LLAMA"]
    folder = tmp_path / "Llama-tiny-hf"
    folder.mkdir()
    json.dump({"architectures": ["LlamaForCausalLM"], "hidden_size": d["hidden"], "intermediate_size": d["inter"],
               "num_hidden_layers": d["n_layers"], "num_attention_heads": d["n_heads"], "vocab_size": d["vocab"],
               "rms_norm_eps": 1e-5, "max_position_embeddings": 4096, "bos_token_id": 1, "eos_token_id": 2,
               "hidden_act": "silu", "torch_dtype": "float16"}, open(folder / "config.json", "w"))
    keys = sorted(wd)
    half = len(keys) // 2
    shard1 = {k: wd[k].contiguous() for k in keys[:half]}
    shard2 = {k: wd[k].contiguous() for k in keys[half:]}
    shard2["model.layers.0.self_attn.rotary_emb.inv_freq"] = torch.ones(d["hidden"] // d["n_heads"] // 2)   # legacy buffer
    save_file(shard1, str(folder / "model-00001-of-00002.safetensors"))
    save_file(shard2, str(folder / "model-00002-of-00002.safetensors"))
    return str(folder)


def test_hf_llama_folder_sharded_safetensors(golden, tmp_path):
    from src.models_clm.modeling_llama_xformer import LlamaForCausalLM
    g, meta = golden
    d = meta["LLAMA"]
    wd = synth.llama_weights(11, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    m = LlamaForCausalLM.from_pretrained(_write_hf_llama_folder(meta, tmp_path, wd), low_cpu_mem_usage=True,
                                         torch_dtype=torch.bfloat16)
    assert m.load_report == {"missing": [], "unexpected": []}
    assert m.config.num_hidden_layers == d["n_layers"] and m.lm_head.weight.dtype == torch.bfloat16
    sd = m.state_dict()
    assert set(sd) == set(wd)
    assert torch.equal(sd["model.layers.1.mlp.down_proj.weight"], wd["model.layers.1.mlp.down_proj.weight"].to(torch.bfloat16))


def _write_sdxl_folder(tmp_path, legacy_vae_attn=False):
    from safetensors.torch import save_file
    root = tmp_path / "stable-diffusion-xl-base-1.0"
    cu, cv = S.TINY_UNET, S.TINY_VAE
    uw = S.synth_weights(S.unet_shapes(cu), 1)
    vw = S.synth_weights(S.vae_decoder_shapes(cv), 2)
    (root / "unet").mkdir(parents=True)
    (root / "vae").mkdir()
    (root / "scheduler").mkdir()
    json.dump({"_class_name": "UNet2DConditionModel", "in_channels": 4, "out_channels": 4,
               "block_out_channels": list(cu["block_out_channels"]), "layers_per_block": 2,
               "down_block_types": ["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
               "up_block_types": ["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"],
               "transformer_layers_per_block": [1, 1, 2], "attention_head_dim": list(cu["num_heads"]),
               "cross_attention_dim": 128, "addition_embed_type": "text_time", "addition_time_embed_dim": 32,
               "projection_class_embeddings_input_dim": 80 + 6 * 32, "norm_num_groups": 32, "use_linear_projection": True},
              open(root / "unet" / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in uw.items()}, str(root / "unet" / "diffusion_pytorch_model.safetensors"))
    vfile = {k: v.contiguous() for k, v in vw.items()}
    if legacy_vae_attn:                 # SDXL-base's VAE file predates the to_q / to_k / to_v / to_out.0 naming
        ren = {".to_q.": ".query.", ".to_k.": ".key.", ".to_v.": ".value.", ".to_out.0.": ".proj_attn."}
        for k in list(vfile):
            for a, b in ren.items():
                if a in k and "mid_block.attentions" in k:
                    vfile[k.replace(a, b)] = vfile.pop(k)
    vfile["encoder.conv_in.weight"] = torch.zeros(32, 3, 3, 3)            # decode-only module: encoder half is ignored
    vfile["quant_conv.weight"] = torch.zeros(8, 8, 1, 1)
    json.dump({"_class_name": "AutoencoderKL", "latent_channels": 4, "out_channels": 3, "in_channels": 3,
               "block_out_channels": list(cv["block_out_channels"]), "layers_per_block": 2, "norm_num_groups": 32,
               "scaling_factor": 0.13025, "force_upcast": True}, open(root / "vae" / "config.json", "w"))
    save_file(vfile, str(root / "vae" / "diffusion_pytorch_model.safetensors"))
    json.dump({"_class_name": "EulerDiscreteScheduler", "_diffusers_version": "0.19.0.dev0", "beta_end": 0.012,
               "beta_schedule": "scaled_linear", "beta_start": 0.00085, "clip_sample": False, "interpolation_type": "linear",
               "num_train_timesteps": 1000, "prediction_type": "epsilon", "sample_max_value": 1.0, "set_alpha_to_one": False,
               "skip_prk_steps": True, "steps_offset": 1, "timestep_spacing": "leading", "trained_betas": None,
               "use_karras_sigmas": False}, open(root / "scheduler" / "scheduler_config.json", "w"))
    return str(root), uw, vw


@pytest.mark.parametrize("legacy", [False, True])
def test_sdxl_diffusers_folder(tmp_path, legacy):
    from seedstory.diffusion import AutoencoderKL, EulerDiscreteScheduler, UNet2DConditionModel
    root, uw, vw = _write_sdxl_folder(tmp_path, legacy_vae_attn=legacy)
    unet = UNet2DConditionModel.from_pretrained(root, subfolder="unet")
    assert unet.load_report == {"missing": [], "unexpected": []}
    assert {k: v for k, v in unet.cfg.items() if k in S.TINY_UNET} == S.TINY_UNET
    sd = unet.state_dict()
    assert set(sd) == set(uw) and all(torch.equal(sd[k], uw[k]) for k in uw)
    vae = AutoencoderKL.from_pretrained(root, subfolder="vae")
    assert vae.load_report == {"missing": [], "unexpected": []}          # encoder.* / quant_conv.* are expected leftovers
    assert vae.config.scaling_factor == 0.13025 and vae.config.force_upcast is True
    sd = vae.state_dict()
    assert set(sd) == set(vw) and all(torch.equal(sd[k], vw[k]) for k in vw)
    sch = EulerDiscreteScheduler.from_pretrained(root, subfolder="scheduler")
    sch.set_timesteps(50)
    ts, sig, init = S.euler_schedule(50)
    assert [int(t) for t in sch.timesteps] == [int(t) for t in ts]
    assert abs(float(sch.sigmas[0]) - float(sig[0])) < 1e-4 * float(sig[0]) and abs(sch.init_noise_sigma - init) < 1e-4 * init


def _vit_kwargs(c):
    return dict(image_size=c["image"], patch_size=c["patch"], width=c["width"], layers=c["layers"], heads=c["heads"],
                mlp_ratio=c["mlp_width"] / c["width"], n_queries=c["n_queries"], output_dim=c["out_dim"])


def _write_detokenizer_file(meta, tmp_path):
    cx = meta["XLV2"]
    uw = S.synth_weights(S.unet_shapes(S.TINY_UNET), 1)
    xw = synth.resampler_xlv2_weights(41, **cx)
    ck = {"unet." + k: v for k, v in uw.items()}
    ck.update({"resampler." + k: v for k, v in xw.items()})
    path = str(tmp_path / "detok_pytorch_model.bin")
    torch.save(ck, path)
    return ck, uw, xw, path


def test_qwen_vit_and_detokenizer_files(golden, tmp_path):
    from seedstory.diffusion import UNet2DConditionModel
    from src.models.qwen_visual import VisionTransformerWithAttnPool
    from src.models_ipa.adapter_modules import SDXLAdapter
    from src.models_ipa.resampler import ResamplerXLV2
    g, meta = golden
    c = meta["VIT"]
    vwd = synth.vit_weights(31, c["width"], c["layers"], c["heads"], c["mlp_width"], c["patch"], c["out_dim"], c["n_queries"])
    p = str(tmp_path / "qwen_vit_G.pt")
    torch.save(vwd, p)
    vit = VisionTransformerWithAttnPool.from_pretrained(pretrained_model_path=p, **_vit_kwargs(c))
    assert vit.load_report == {"missing": [], "unexpected": []}
    sd = vit.state_dict()
    assert all(torch.equal(sd[k], vwd[k]) for k in vwd)
    ck, uw, xw, path = _write_detokenizer_file(meta, tmp_path)
    adapter = SDXLAdapter.from_pretrained(unet=UNet2DConditionModel(S.TINY_UNET), resampler=ResamplerXLV2(**meta["XLV2"]),
                                          pretrained_model_path=path)
    assert adapter.load_report == {"missing": [], "unexpected": []}
    sd = adapter.state_dict()
    assert set(sd) == set(ck) and all(torch.equal(sd[k], ck[k]) for k in ck)


# ------------------------------------------------------------------------------------------------------------
# GPU: a module loaded from the file computes what the oracle computes from the same tensors
# ------------------------------------------------------------------------------------------------------------


@pytest.mark.gpu
def test_agent_file_generate_parity(golden, tmp_path):
    """Agent checkpoint from disk (LoRA factors + modules_to_save norms) -> ContinuousLVLM.generate vs the oracle run
    UNMERGED (y = Wx + 2·B(Ax), peft semantics) on the adapter norms: ids exact, img_gen_feat within 1e-4 (fp32)."""
    g, meta = golden
    wd, ck, path = _write_agent_file(meta, tmp_path)
    agent = _agent_from_file(meta, path).eval().to(DEV)
    llm = agent.llm.base_model.model
    llm.cache_cap, llm.max_new, llm.max_prefill_rows = 256, 128, 64
    llm.use_kv_cache_head = False                                       # gen_george.py:165 through the wrappers
    d = meta["LLAMA"]
    dims = O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"])
    input_ids = g["gen.input_ids"]
    n_in = meta["RES_IN"]["grid"] ** 2
    mask = torch.zeros_like(input_ids, dtype=torch.bool)
    mask[0, 14:14 + n_in] = True
    forced = g["gen.forced"].tolist()
    ref = O.lvlm_generate(wd, dims, input_ids, g["gen.image_embeds"], torch.tensor([True]), mask, _img_ids(meta),
                          max_new_tokens=90, forced=forced, n_heads_resampler=2, lora_scaling=2.0)
    out = agent.generate(tokenizer=_Tok(_img_ids(meta)), input_ids=input_ids, image_embeds=g["gen.image_embeds"].to(DEV),
                         embeds_cmp_mask=torch.tensor([True]), ids_cmp_mask=mask, max_new_tokens=90,
                         num_img_gen_tokens=64, forced_tokens=forced)
    assert out["generate_ids"].tolist() == ref["generate_ids"]
    assert out["has_img_output"] and rel(out["img_gen_feat"], ref["img_gen_feat"]) < 1e-4
    # the frozen original norms in the file (all ones) must NOT be what ran: the oracle on them is far away
    wd_orig = {k: (torch.ones_like(v) if k.endswith("norm.weight") else v) for k, v in wd.items()}
    far = O.lvlm_generate(wd_orig, dims, input_ids, g["gen.image_embeds"], torch.tensor([True]), mask, _img_ids(meta),
                          max_new_tokens=90, forced=forced, n_heads_resampler=2, lora_scaling=2.0)
    assert rel(out["img_gen_feat"], far["img_gen_feat"]) > 1e-2


@pytest.mark.gpu
def test_hf_llama_folder_and_adapter_folder_parity(golden, tmp_path):
    """HF folder (sharded safetensors) + peft adapter folder -> prefill hidden / logits vs the unmerged oracle."""
    from src.models_clm.modeling_llama_xformer import LlamaForCausalLM
    from src.models_clm.peft_models import get_peft_model_with_resize_embedding
    g, meta = golden
    d = meta["LLAMA"]
    wd = synth.llama_weights(12, d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"], lora_r=16)
    base = {k: v for k, v in wd.items() if ".lora_" not in k}
    base_file = {k: (torch.ones_like(v) if k.endswith("norm.weight") else v) for k, v in base.items()}
    hf = _write_hf_llama_folder(meta, tmp_path, base_file)
    folder = tmp_path / "adapter"
    folder.mkdir()
    json.dump(LORA_CFG, open(folder / "adapter_config.json", "w"))
    torch.save({k.replace(".default.", "."): v for k, v in peft_layout(wd, with_base=False, with_original=False).items()},
               str(folder / "adapter_model.bin"))
    llm = LlamaForCausalLM.from_pretrained(hf, torch_dtype=torch.float32)
    pm = get_peft_model_with_resize_embedding(llm, model_id=str(folder), torch_dtype="fp32")
    pm = pm.to(DEV)
    llm.cache_cap, llm.max_new, llm.max_prefill_rows = 128, 16, 64
    ids = synth.randint(51, (1, 33), 3, d["vocab"] - 70)
    emb = wd["model.embed_tokens.weight"]
    eng = llm.engine_for_generation(tuple(_img_ids(meta)))
    eng.reset()
    hid = eng.prefill(emb[ids[0]].to(DEV), want_hidden=True)
    logits_o, last_o, _ = O.llama_forward(wd, O.LlamaDims(d["hidden"], d["n_heads"], d["n_layers"], d["inter"], d["vocab"]),
                                          emb[ids], torch.arange(33).unsqueeze(0), None, 2.0)
    assert rel(hid, last_o[0]) < 1e-4
    assert rel(eng.logits, logits_o[0, -1]) < 1e-4


@pytest.mark.gpu
def test_detokenizer_and_sdxl_folder_parity(golden, tmp_path):
    """UNet from the de-tokenizer file (``unet.*`` overrides the SDXL-base UNet it was fine-tuned from), VAE from the
    diffusers folder in its legacy attention naming, ViT from qwen_vit_G.pt: each vs the oracle on the same tensors."""
    from seedstory.diffusion import AutoencoderKL, UNet2DConditionModel
    from src.models.qwen_visual import VisionTransformerWithAttnPool
    from src.models_ipa.adapter_modules import SDXLAdapter
    from src.models_ipa.resampler import ResamplerXLV2
    g, meta = golden
    root, uw_base, vw = _write_sdxl_folder(tmp_path, legacy_vae_attn=True)
    ck, uw, xw, path = _write_detokenizer_file(meta, tmp_path)
    # make the fine-tuned UNet differ from the base one so the override is observable
    ck = {k: (v * 1.25 if k.startswith("unet.") and v.dim() > 1 else v) for k, v in ck.items()}
    torch.save(ck, path)
    uw = {k[len("unet."):]: v for k, v in ck.items() if k.startswith("unet.")}
    unet = UNet2DConditionModel.from_pretrained(root, subfolder="unet")
    adapter = SDXLAdapter.from_pretrained(unet=unet, resampler=ResamplerXLV2(**meta["XLV2"]), pretrained_model_path=path)
    adapter = adapter.to(DEV).eval()
    c = S.TINY_UNET
    x = synth.normal_like(5, (2, 4, 16, 16), 1.0)
    ctx = synth.normal_like(6, (2, 8, 128), 1.0)
    pooled = synth.normal_like(7, (2, 80), 1.0)
    tid = torch.tensor([[128, 128, 0, 0, 128, 128]] * 2, dtype=torch.float32)
    ref = S.unet_forward(uw, c, x, torch.tensor(801.0), ctx, pooled, tid)
    y = adapter.unet(x.to(DEV), 801.0, ctx.to(DEV), added_cond_kwargs={"text_embeds": pooled.to(DEV), "time_ids": tid}).sample
    assert rel(y, ref) < 2e-4
    assert rel(y, S.unet_forward(uw_base, c, x, torch.tensor(801.0), ctx, pooled, tid)) > 1e-2
    vae = AutoencoderKL.from_pretrained(root, subfolder="vae").to(DEV)
    lat = synth.normal_like(20, (1, 4, 12, 12), 1.0)
    yv = vae.decode((lat / 0.13025).to(DEV)).sample
    assert rel(yv, S.vae_decode(vw, S.TINY_VAE, lat)) < 2e-4
    cvit = meta["VIT"]
    vwd = synth.vit_weights(31, cvit["width"], cvit["layers"], cvit["heads"], cvit["mlp_width"], cvit["patch"],
                            cvit["out_dim"], cvit["n_queries"])
    p = str(tmp_path / "qwen_vit_G.pt")
    torch.save(vwd, p)
    vit = VisionTransformerWithAttnPool.from_pretrained(pretrained_model_path=p, **_vit_kwargs(cvit)).to(DEV)
    assert rel(vit(g["vit.x"].to(DEV)), g["vit.y"]) < 1e-4
