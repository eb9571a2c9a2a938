"""MI355X-native drop-in for ``SDXLAdapter`` of the reference's ``src/models_ipa/adapter_modules.py``
(:281-468) — the de-tokenizer that renders a 1024x1024 image from the regressed 256x4096 image
feature.  Same constructor / ``from_pretrained`` / ``init_pipe`` / ``get_image_embeds`` / ``generate``
signatures and return values; the UNet, VAE, scheduler and pipeline objects are the HIP-backed
classes of ``seedstory.diffusion`` (diffusers is absent).  The other adapter classes of the reference
file (IPAdapterSD*, SDXLText2ImageAndEditAdapter, SD21..., SDXLAdapterWithLatentImage) are not on the
story path and are not provided.

One deliberate optimisation with identical output: the reference re-encodes an all-zeros 448x448
image with the full ViT-G for the negative branch on EVERY call (:406-414); the result is a constant
of the weights, so it is computed once per (encoder weights, size, dtype) and cached.
"""
import itertools

import torch
from torch import nn

from seedstory.diffusion import StableDiffusionXLPipeline


def compute_time_ids(original_size, crops_coords_top_left, target_resolution):
    target_size = (target_resolution, target_resolution)
    return torch.tensor([list(original_size + crops_coords_top_left + target_size)])


class SDXLAdapter(nn.Module):

    def __init__(self, unet, resampler, full_ft=False) -> None:
        super().__init__()
        self.unet = unet
        self.resampler = resampler
        self.full_ft = full_ft
        self._neg_cache = {}

    def params_to_opt(self):
        return itertools.chain(self.resampler.parameters(), [])

    @torch.no_grad()
    def forward(self, noisy_latents, timesteps, image_embeds, text_embeds, noise, time_ids):
        """The training-side forward of the reference (adapter_modules.py:330-343), FORWARD ONLY (no autograd; SURVEY §8
        row f4): conditioning through the resampler, one UNet call on the noised latents at per-sample ``timesteps``,
        ``F.mse_loss(noise_pred.float(), noise.float())`` as one fixed-order device reduction."""
        from seedstory import ops
        image_embeds, pooled_image_embeds = self.resampler(image_embeds)
        unet_added_conditions = {"time_ids": time_ids, 'text_embeds': pooled_image_embeds}
        noise_pred = self.unet(noisy_latents, timesteps, image_embeds, added_cond_kwargs=unet_added_conditions).sample
        a, b = noise_pred, noise.to(noise_pred.device)
        if a.dtype != b.dtype:
            a, b = a.float(), b.float()
        loss = ops.mse_loss(a.contiguous(), b.contiguous())
        return {'total_loss': loss, 'noise_pred': noise_pred}

    def encode_image_embeds(self, image_embeds):
        return self.resampler(image_embeds)

    @classmethod
    def from_pretrained(cls, unet, resampler, pretrained_model_path=None, **kwargs):
        model = cls(unet=unet, resampler=resampler, **kwargs)
        if pretrained_model_path is not None:
            from seedstory import ckpt as _ckpt
            _ckpt.load_checked(model, _ckpt.read_weights(pretrained_model_path), 'detokenizer,')
        return model

    def init_pipe(self, vae, scheduler, visual_encoder, image_transform, discrete_model=None, dtype=torch.float16,
                  device='cuda'):
        self.device = device
        self.dtype = dtype
        self.sdxl_pipe = StableDiffusionXLPipeline(tokenizer=None, tokenizer_2=None, text_encoder=None,
                                                   text_encoder_2=None, vae=vae, unet=self.unet, scheduler=scheduler)
        self.visual_encoder = visual_encoder.to(self.device, dtype=self.dtype)
        self.discrete_model = discrete_model.to(self.device, dtype=self.dtype) if discrete_model is not None else None
        self.image_transform = image_transform

    def _negative_embeds(self, image_size, like):
        p0 = next(self.visual_encoder.parameters())
        key = (image_size, like.dtype, str(like.device), p0.data_ptr(), p0._version)
        if key not in self._neg_cache:
            zeros = torch.zeros(1, 3, image_size, image_size, device=like.device, dtype=like.dtype)
            self._neg_cache = {key: self.visual_encoder(zeros)}        # reference :406-414, hoisted
        return self._neg_cache[key]

    @torch.inference_mode()
    def get_image_embeds(self, image_pil=None, image_tensor=None, image_embeds=None, return_negative=True,
                         image_size=448):
        assert int(image_pil is not None) + int(image_tensor is not None) + int(image_embeds is not None) == 1
        if image_pil is not None:
            image_tensor = self.image_transform(image_pil).unsqueeze(0).to(self.device, dtype=self.dtype)
        if image_tensor is not None:
            if return_negative:
                image_tensor = torch.cat([image_tensor, torch.zeros_like(image_tensor)], dim=0)
            image_embeds = self.visual_encoder(image_tensor)
        elif return_negative:
            neg = self._negative_embeds(image_size, image_embeds)
            image_embeds = torch.cat([image_embeds, neg.expand(image_embeds.shape[0], -1, -1)], dim=0)
        if self.discrete_model is not None:
            image_embeds = self.discrete_model.encode_image_embeds(image_embeds)
        image_embeds, pooled_image_embeds = self.encode_image_embeds(image_embeds)
        if return_negative:
            image_embeds, image_embeds_neg = image_embeds.chunk(2)
            pooled_image_embeds, pooled_image_embeds_neg = pooled_image_embeds.chunk(2)
        else:
            image_embeds_neg = None
            pooled_image_embeds_neg = None
        return image_embeds, image_embeds_neg, pooled_image_embeds, pooled_image_embeds_neg

    def generate(self, image_pil=None, image_tensor=None, image_embeds=None, seed=42, height=1024, width=1024,
                 guidance_scale=7.5, num_inference_steps=30, input_image_size=448, **kwargs):
        pos, neg, pooled_pos, pooled_neg = self.get_image_embeds(image_pil=image_pil, image_tensor=image_tensor,
                                                                 image_embeds=image_embeds, return_negative=True,
                                                                 image_size=input_image_size)
        generator = torch.Generator(self.device).manual_seed(seed) if seed is not None else None
        if pos.shape[0] > 1 and "latents" not in kwargs:
            # several stories rendered together: every image starts from the noise the reference draws for a
            # single call with this seed (:455), so each equals its own batch-1 generate()
            one = torch.randn((1, 4, height // 8, width // 8), generator=generator, device=pos.device, dtype=pos.dtype)
            kwargs["latents"] = one.repeat(pos.shape[0], 1, 1, 1)
        return self.sdxl_pipe(prompt_embeds=pos.contiguous(), negative_prompt_embeds=neg.contiguous(),
                              pooled_prompt_embeds=pooled_pos.contiguous(),
                              negative_pooled_prompt_embeds=pooled_neg.contiguous(), guidance_scale=guidance_scale,
                              num_inference_steps=num_inference_steps, generator=generator, height=height, width=width,
                              **kwargs).images
