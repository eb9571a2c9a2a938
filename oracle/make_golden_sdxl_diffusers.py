"""TEST INFRASTRUCTURE ONLY — closes the "parity unpinned" status of ``oracle/sdxl_oracle.py`` on any box that HAS
diffusers (this image does not: ``import diffusers`` fails, and there is no network).

    python oracle/make_golden_sdxl_diffusers.py            # tiny configuration: every tensor  (seconds)
    python oracle/make_golden_sdxl_diffusers.py --full     # + SDXL-base: eps rows, 2 Euler+CFG steps, a VAE crop (≈ 10 min of CPU)

writes ``tests/golden/sdxl_diffusers.safetensors`` (+ ``.json``: diffusers / torch versions, the configs, the seeds).
``tests/test_sdxl_pin.py`` pins ``sdxl_oracle`` on that file when it exists and otherwise SKIPS with the reason
"parity unpinned: fixture absent".

What is run is the REAL library behind the reference's call sites:
  * ``UNet2DConditionModel`` / ``AutoencoderKL`` / ``EulerDiscreteScheduler`` — ``/root/reference/src/inference/gen_george.py:10,60-64``
    (``from_pretrained(..., subfolder=...)`` there; here ``from_config`` with the published SDXL-base config keys and
    the seeded weights of ``sdxl_oracle.synth_weights`` loaded with ``load_state_dict(strict=True)``);
  * ``StableDiffusionXLPipeline.__call__`` as ``SDXLAdapter.generate`` drives it —
    ``/root/reference/src/models_ipa/adapter_modules.py:369-375,455-466``: text-encoder-less pipeline,
    ``prompt_embeds`` / ``negative_prompt_embeds`` / pooled embeds passed in, ``guidance_scale`` 7.5.  The pipeline
    call itself is reproduced step by step with the real scheduler and UNet objects (``scale_model_input`` →
    UNet on the [negative; positive] batch with ``added_cond_kwargs`` → guidance → ``scheduler.step``) so the fixture
    does not depend on which pipeline arguments a diffusers version insists on; with ``--pipeline`` the real
    ``StableDiffusionXLPipeline`` object is ALSO called (``output_type="latent"``) and must agree.
Inputs are ``oracle/synth.py`` tensors (pure functions of seed and index), so only outputs are stored.
"""
import argparse
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)

import sdxl_oracle as S  # noqa: E402
import sdxl_modules as M  # noqa: E402  (only for the published-key config dictionaries)
import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "sdxl_diffusers")

SEEDS = dict(unet=1, vae=2, x=5, ctx=6, pooled=7, noise=8, z=9)


def unet_config(c):
    k = M.config_from_oracle(c)
    return dict(sample_size=128, in_channels=k["in_channels"], out_channels=k["out_channels"],
                down_block_types=tuple(k["down_block_types"]), up_block_types=tuple(k["up_block_types"]),
                block_out_channels=tuple(k["block_out_channels"]), layers_per_block=k["layers_per_block"],
                transformer_layers_per_block=tuple(k["transformer_layers_per_block"]),
                attention_head_dim=tuple(k["attention_head_dim"]), cross_attention_dim=k["cross_attention_dim"],
                addition_embed_type="text_time", addition_time_embed_dim=k["addition_time_embed_dim"],
                projection_class_embeddings_input_dim=k["projection_class_embeddings_input_dim"],
                use_linear_projection=True, norm_num_groups=k["norm_num_groups"], norm_eps=1e-5, flip_sin_to_cos=True,
                freq_shift=0, act_fn="silu", downsample_padding=1, mid_block_type="UNetMidBlock2DCrossAttn",
                upcast_attention=False, resnet_time_scale_shift="default")


def vae_config(c):
    n = len(c["block_out_channels"])
    return dict(in_channels=3, out_channels=c["out_channels"], down_block_types=("DownEncoderBlock2D",) * n,
                up_block_types=("UpDecoderBlock2D",) * n, block_out_channels=tuple(c["block_out_channels"]),
                layers_per_block=c["layers_per_block"], act_fn="silu", latent_channels=c["latent_channels"],
                norm_num_groups=c["norm_groups"], sample_size=1024, scaling_factor=c["scaling_factor"], force_upcast=True)


SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
             timestep_spacing="leading", steps_offset=1, prediction_type="epsilon", interpolation_type="linear",
             use_karras_sigmas=False)


STANDIN = False      # --standin: exercise this script's own data flow without diffusers (see _standin_build)


class _Out:
    def __init__(self, t):
        self.sample = t
        self.prev_sample = t


def _standin_build(c_unet, c_vae):
    """NOT a pin: objects with the call signatures of the diffusers classes used below, backed by oracle/sdxl_modules.py, so that the
    recipe (weight loading, loop, fixture keys) and tests/test_sdxl_pin.py's consumer can be run end to end in an image without
    diffusers (tests/test_sdxl_pin.py::test_pin_recipe_runs_end_to_end_with_standin).  The fixture it writes is marked
    `"diffusers": "STANDIN"` and must never be committed as tests/golden/sdxl_diffusers.*."""
    import types
    net = M.UNet2DConditionModel(M.config_from_oracle(c_unet)).eval()
    net.load_state_dict(S.synth_weights(S.unet_shapes(c_unet), SEEDS["unet"]), strict=True)
    dec = M.AutoencoderKLDecoder(M.vae_config_from_oracle(c_vae)).eval()
    dec.load_state_dict(S.synth_weights(S.vae_decoder_shapes(c_vae), SEEDS["vae"]), strict=True)

    class U:
        def __call__(self, x, t, encoder_hidden_states=None, added_cond_kwargs=None):
            return _Out(net(x, t, encoder_hidden_states, added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]))

    class V:
        config = types.SimpleNamespace(scaling_factor=c_vae["scaling_factor"])

        def decode(self, z):
            return _Out(dec.decode(z))

    class Sch:
        def set_timesteps(self, n):
            ts, sig, init = M.euler_tables(n)
            self.timesteps = torch.tensor(ts, dtype=torch.float32)
            self.sigmas = torch.tensor(sig, dtype=torch.float32)
            self.init_noise_sigma = init
            self._i = {float(t): k for k, t in enumerate(ts)}

        def scale_model_input(self, x, t):
            s_ = float(self.sigmas[self._i[float(t)]])
            return x / (s_ * s_ + 1.0) ** 0.5

        def step(self, eps, t, x):
            k = self._i[float(t)]
            return _Out(x + eps * (float(self.sigmas[k + 1]) - float(self.sigmas[k])))
    return U(), V(), Sch()


def build(c_unet, c_vae):
    if STANDIN:
        return _standin_build(c_unet, c_vae)
    from diffusers import AutoencoderKL, EulerDiscreteScheduler, UNet2DConditionModel
    unet = UNet2DConditionModel.from_config(unet_config(c_unet)).eval()
    unet.load_state_dict(S.synth_weights(S.unet_shapes(c_unet), SEEDS["unet"]), strict=True)
    vae = AutoencoderKL.from_config(vae_config(c_vae)).eval()
    # decoder half only (the encoder is never on the path): strict over the decoder + post_quant_conv keys
    dec = S.synth_weights(S.vae_decoder_shapes(c_vae), SEEDS["vae"])
    want = {k for k in vae.state_dict() if k.startswith("decoder.") or k.startswith("post_quant_conv.")}
    assert want == set(dec), sorted(want ^ set(dec))[:8]
    vae.load_state_dict(dec, strict=False)
    return unet, vae, EulerDiscreteScheduler(**SCHED)


def inputs(c, hw, tokens):
    x = synth.normal_like(SEEDS["x"], (2, 4, hw, hw), 1.0)
    ctx = synth.normal_like(SEEDS["ctx"], (2, tokens, c["cross_attention_dim"]), 1.0)
    pooled = synth.normal_like(SEEDS["pooled"], (2, c["pooled_dim"]), 1.0)
    return x, ctx, pooled


@torch.no_grad()
def euler_cfg(unet, sched, noise, ctx, pooled, size, steps, n_run, guidance=7.5):
    """The denoising loop of StableDiffusionXLPipeline.__call__ with the real scheduler / UNet objects."""
    sched.set_timesteps(steps)
    lat = noise * sched.init_noise_sigma
    ids = torch.tensor([[size, size, 0, 0, size, size]] * 2, dtype=noise.dtype)
    for t in sched.timesteps[:n_run]:
        xin = sched.scale_model_input(torch.cat([lat] * 2), t)
        eps = unet(xin, t, encoder_hidden_states=ctx, added_cond_kwargs={"text_embeds": pooled, "time_ids": ids}).sample
        e_neg, e_pos = eps.chunk(2)
        lat = sched.step(e_neg + guidance * (e_pos - e_neg), t, lat).prev_sample
    return lat


@torch.no_grad()
def run(tag, c_unet, c_vae, hw, tokens, out, full):
    unet, vae, sched = build(c_unet, c_vae)
    x, ctx, pooled = inputs(c_unet, hw, tokens)
    size = hw * 8
    ids = torch.tensor([[size, size, 0, 0, size, size]] * 2, dtype=torch.float32)
    eps = unet(x, torch.tensor(801.0), encoder_hidden_states=ctx,
               added_cond_kwargs={"text_embeds": pooled, "time_ids": ids}).sample
    out[tag + ".unet_eps_t801"] = eps if not full else eps[:, :, :8].contiguous()     # full: 8 rows of every channel
    noise = synth.normal_like(SEEDS["noise"], (1, 4, hw, hw), 1.0)
    # row 0 of ctx / pooled plays the negative branch, row 1 the positive one
    lat = euler_cfg(unet, sched, noise, ctx, pooled, size, steps=30, n_run=2)
    out[tag + ".latents_after_2_of_30"] = lat
    sched.set_timesteps(30)
    out[tag + ".timesteps30"] = sched.timesteps.to(torch.float32).clone()
    out[tag + ".sigmas30"] = sched.sigmas.to(torch.float32).clone()
    out[tag + ".init_noise_sigma30"] = torch.tensor([float(sched.init_noise_sigma)])
    zhw = hw if not full else 32                                                        # full: a 256² crop
    z = synth.normal_like(SEEDS["z"], (1, 4, zhw, zhw), 1.0)
    img = vae.decode(z / vae.config.scaling_factor).sample
    out[tag + ".vae_image"] = img
    if STANDIN:
        u8 = (img / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
    else:
        from diffusers.image_processor import VaeImageProcessor
        u8 = VaeImageProcessor(vae_scale_factor=8).postprocess(img, output_type="np")
    out[tag + ".vae_u8"] = torch.from_numpy((u8 * 255).round().astype("uint8"))
    return unet, vae, sched


def main(argv=None):
    global STANDIN
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also run the SDXL-base configuration (2.57 B parameters, fp32 on CPU)")
    ap.add_argument("--pipeline", action="store_true", help="also call the real StableDiffusionXLPipeline object (tiny config)")
    ap.add_argument("--standin", action="store_true", help="self-test of this script without diffusers (NOT a pin; needs --out)")
    ap.add_argument("--out", default=OUT, help="output path without extension (default: tests/golden/sdxl_diffusers)")
    a = ap.parse_args(argv)
    STANDIN = bool(a.standin)
    if STANDIN:
        assert a.out != OUT and not a.pipeline, "--standin writes a self-test file: give it its own --out, and no --pipeline"
        import types
        diffusers = types.SimpleNamespace(__version__="STANDIN")
    else:
        import diffusers
    out = {}
    unet, vae, sched = run("tiny", S.TINY_UNET, S.TINY_VAE, 8, 8, out, False)
    if a.pipeline:
        from diffusers import StableDiffusionXLPipeline
        pipe = StableDiffusionXLPipeline(unet=unet, vae=vae, scheduler=sched, tokenizer=None, tokenizer_2=None,
                                         text_encoder=None, text_encoder_2=None)     # adapter_modules.py:369-375
        _, ctx, pooled = inputs(S.TINY_UNET, 8, 8)
        noise = synth.normal_like(SEEDS["noise"], (1, 4, 8, 8), 1.0)
        lat = pipe(prompt_embeds=ctx[1:], negative_prompt_embeds=ctx[:1], pooled_prompt_embeds=pooled[1:],
                   negative_pooled_prompt_embeds=pooled[:1], num_inference_steps=30, guidance_scale=7.5, height=64, width=64,
                   latents=noise.clone(), output_type="latent").images              # adapter_modules.py:455-466
        out["tiny.pipeline_latents_30"] = lat
    if a.full:
        run("full", S.SDXL_BASE_UNET, S.SDXL_BASE_VAE, 128, 77, out, True)
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in out.items()}, a.out + ".safetensors")
    with open(a.out + ".json", "w") as f:
        json.dump(dict(diffusers=diffusers.__version__, torch=torch.__version__, seeds=SEEDS, full=a.full,
                       pipeline=a.pipeline, tensors={k: list(v.shape) for k, v in out.items()}), f, indent=1)
    print("wrote", a.out + ".safetensors", "with", len(out), "tensors; diffusers", diffusers.__version__)


if __name__ == "__main__":
    main()
