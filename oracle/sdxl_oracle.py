"""TEST INFRASTRUCTURE ONLY — CPU restatement of the SDXL de-tokenizer the reference drives through
diffusers (``StableDiffusionXLPipeline`` built text-encoder-less at
``src/models_ipa/adapter_modules.py:369-375`` and called at ``:455-466``; UNet / VAE / scheduler
instantiated at ``src/inference/gen_george.py:60-64``).

**Parity unpinned**: diffusers is absent from the reference tree and from this image (unpinned in
``requirements.txt:6``), and the reference has no test that pins any number at this boundary.  The
functions below restate the *published* SDXL-base architecture and the EulerDiscrete / pipeline
semantics on a flat weight dict with the diffusers checkpoint key
names, so real checkpoints drop in.  Structural pin: ``unet_param_count(SDXL_BASE_UNET)`` reproduces
the published 2,566,942,084 UNet parameters (checked in tests/test_sdxl_oracle.py).
Never imported by the product path.
"""
import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

W = Dict[str, torch.Tensor]

SDXL_BASE_UNET = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
                      transformer_layers=(0, 2, 10), num_heads=(5, 10, 20), cross_attention_dim=2048,
                      addition_time_embed_dim=256, pooled_dim=1280, norm_groups=32)
SDXL_BASE_VAE = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                     norm_groups=32, scaling_factor=0.13025)
TINY_UNET = dict(in_channels=4, out_channels=4, block_out_channels=(64, 128, 256), layers_per_block=2,
                 transformer_layers=(0, 1, 2), num_heads=(1, 2, 4), cross_attention_dim=128,
                 addition_time_embed_dim=32, pooled_dim=80, norm_groups=32)
TINY_VAE = dict(latent_channels=4, out_channels=3, block_out_channels=(32, 64, 64, 64), layers_per_block=2,
                norm_groups=32, scaling_factor=0.13025)


# ---- shapes / synthetic weights ---------------------------------------------------------------------

def unet_shapes(c) -> Dict[str, tuple]:
    """Every parameter of the UNet2DConditionModel (diffusers key -> shape) for config c."""
    s = {}
    boc = c["block_out_channels"]
    temb = boc[0] * 4
    xdim = c["cross_attention_dim"]

    def lin(n, o, i, bias=True):
        s[n + ".weight"] = (o, i)
        if bias:
            s[n + ".bias"] = (o,)

    def conv(n, o, i, k):
        s[n + ".weight"] = (o, i, k, k)
        s[n + ".bias"] = (o,)

    def norm(n, ch):
        s[n + ".weight"] = (ch,)
        s[n + ".bias"] = (ch,)

    def resnet(n, i, o):
        norm(n + ".norm1", i); conv(n + ".conv1", o, i, 3); lin(n + ".time_emb_proj", o, temb)
        norm(n + ".norm2", o); conv(n + ".conv2", o, o, 3)
        if i != o:
            conv(n + ".conv_shortcut", o, i, 1)

    def transformer(n, ch, layers):
        norm(n + ".norm", ch); lin(n + ".proj_in", ch, ch)
        for k in range(layers):
            b = n + ".transformer_blocks.%d" % k
            norm(b + ".norm1", ch)
            for p in ("to_q", "to_k", "to_v"):
                lin(b + ".attn1." + p, ch, ch, bias=False)
            lin(b + ".attn1.to_out.0", ch, ch)
            norm(b + ".norm2", ch)
            lin(b + ".attn2.to_q", ch, ch, bias=False)
            lin(b + ".attn2.to_k", ch, xdim, bias=False)
            lin(b + ".attn2.to_v", ch, xdim, bias=False)
            lin(b + ".attn2.to_out.0", ch, ch)
            norm(b + ".norm3", ch)
            lin(b + ".ff.net.0.proj", 8 * ch, ch)
            lin(b + ".ff.net.2", ch, 4 * ch)
        lin(n + ".proj_out", ch, ch)

    conv("conv_in", boc[0], c["in_channels"], 3)
    lin("time_embedding.linear_1", temb, boc[0]); lin("time_embedding.linear_2", temb, temb)
    lin("add_embedding.linear_1", temb, 6 * c["addition_time_embed_dim"] + c["pooled_dim"])
    lin("add_embedding.linear_2", temb, temb)
    L = c["layers_per_block"]
    ch = boc[0]
    skips = [ch]
    for i, o in enumerate(boc):
        for j in range(L):
            resnet("down_blocks.%d.resnets.%d" % (i, j), ch, o)
            ch = o
            if c["transformer_layers"][i]:
                transformer("down_blocks.%d.attentions.%d" % (i, j), o, c["transformer_layers"][i])
            skips.append(ch)
        if i < len(boc) - 1:
            conv("down_blocks.%d.downsamplers.0.conv" % i, o, o, 3)
            skips.append(ch)
    resnet("mid_block.resnets.0", ch, ch)
    transformer("mid_block.attentions.0", ch, c["transformer_layers"][-1])
    resnet("mid_block.resnets.1", ch, ch)
    for i, o in enumerate(reversed(boc)):
        tl = list(reversed(c["transformer_layers"]))[i]
        for j in range(L + 1):
            sk = skips.pop()
            resnet("up_blocks.%d.resnets.%d" % (i, j), ch + sk, o)
            ch = o
            if tl:
                transformer("up_blocks.%d.attentions.%d" % (i, j), o, tl)
        if i < len(boc) - 1:
            conv("up_blocks.%d.upsamplers.0.conv" % i, o, o, 3)
    norm("conv_norm_out", ch)
    conv("conv_out", c["out_channels"], ch, 3)
    return s


def vae_decoder_shapes(c) -> Dict[str, tuple]:
    s = {}
    boc = c["block_out_channels"]

    def conv(n, o, i, k):
        s[n + ".weight"] = (o, i, k, k); s[n + ".bias"] = (o,)

    def norm(n, ch):
        s[n + ".weight"] = (ch,); s[n + ".bias"] = (ch,)

    def resnet(n, i, o):
        norm(n + ".norm1", i); conv(n + ".conv1", o, i, 3); norm(n + ".norm2", o); conv(n + ".conv2", o, o, 3)
        if i != o:
            conv(n + ".conv_shortcut", o, i, 1)

    lc = c["latent_channels"]
    conv("post_quant_conv", lc, lc, 1)
    top = boc[-1]
    conv("decoder.conv_in", top, lc, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for p in ("to_q", "to_k", "to_v", "to_out.0"):
        s[a + "." + p + ".weight"] = (top, top); s[a + "." + p + ".bias"] = (top,)
    resnet("decoder.mid_block.resnets.1", top, top)
    ch = top
    for i, o in enumerate(reversed(boc)):
        for j in range(c["layers_per_block"] + 1):
            resnet("decoder.up_blocks.%d.resnets.%d" % (i, j), ch, o)
            ch = o
        if i < len(boc) - 1:
            conv("decoder.up_blocks.%d.upsamplers.0.conv" % i, o, o, 3)
    norm("decoder.conv_norm_out", ch)
    conv("decoder.conv_out", c["out_channels"], ch, 3)
    return s


def unet_param_count(c) -> int:
    return sum(int(np.prod(v)) for v in unet_shapes(c).values())


def synth_weights(shapes, seed, dtype=torch.float32) -> W:
    import synth
    wd = {}
    for i, (k, shp) in enumerate(sorted(shapes.items())):
        if k.endswith("weight") and len(shp) == 1:
            wd[k] = synth.normal_like(seed * 100003 + i, shp, 0.1, 1.0, dtype=dtype)
        elif len(shp) == 1:
            wd[k] = synth.normal_like(seed * 100003 + i, shp, 0.02, dtype=dtype)
        else:
            fan_in = int(np.prod(shp[1:]))
            wd[k] = synth.normal_like(seed * 100003 + i, shp, 1.0 / math.sqrt(fan_in), dtype=dtype)
    return wd


# ---- building blocks (NCHW) -----------------------------------------------------------------------------

def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def _lin(wd, n, x):
    return F.linear(x, wd[n + ".weight"], wd.get(n + ".bias"))


def _gn(wd, n, x, groups, eps):
    return F.group_norm(x, groups, wd[n + ".weight"], wd[n + ".bias"], eps)


def _resnet(wd, n, x, temb, groups, eps=1e-5):
    h = F.silu(_gn(wd, n + ".norm1", x, groups, eps))
    h = F.conv2d(h, wd[n + ".conv1.weight"], wd[n + ".conv1.bias"], padding=1)
    if temb is not None:
        h = h + _lin(wd, n + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(_gn(wd, n + ".norm2", h, groups, eps))
    h = F.conv2d(h, wd[n + ".conv2.weight"], wd[n + ".conv2.bias"], padding=1)
    if (n + ".conv_shortcut.weight") in wd:
        x = F.conv2d(x, wd[n + ".conv_shortcut.weight"], wd[n + ".conv_shortcut.bias"])
    return x + h


def _attn(q, k, v, heads):
    B, Lq, C = q.shape
    hd = C // heads
    qh = q.view(B, Lq, heads, hd).transpose(1, 2)
    kh = k.view(B, -1, heads, hd).transpose(1, 2)
    vh = v.view(B, -1, heads, hd).transpose(1, 2)
    s = torch.matmul(qh.float(), kh.float().transpose(-1, -2)) / math.sqrt(hd)
    o = torch.matmul(torch.softmax(s, dim=-1), vh.float()).to(q.dtype)
    return o.transpose(1, 2).reshape(B, Lq, C)


def _transformer(wd, n, x, ctx, heads, layers, groups):
    B, C, Hh, Ww = x.shape
    res = x
    h = _gn(wd, n + ".norm", x, groups, 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(B, Hh * Ww, C)
    h = _lin(wd, n + ".proj_in", h)
    for k in range(layers):
        b = n + ".transformer_blocks.%d" % k
        y = F.layer_norm(h, (C,), wd[b + ".norm1.weight"], wd[b + ".norm1.bias"], 1e-5)
        a = _attn(_lin(wd, b + ".attn1.to_q", y), _lin(wd, b + ".attn1.to_k", y), _lin(wd, b + ".attn1.to_v", y), heads)
        h = h + _lin(wd, b + ".attn1.to_out.0", a)
        y = F.layer_norm(h, (C,), wd[b + ".norm2.weight"], wd[b + ".norm2.bias"], 1e-5)
        a = _attn(_lin(wd, b + ".attn2.to_q", y), _lin(wd, b + ".attn2.to_k", ctx), _lin(wd, b + ".attn2.to_v", ctx), heads)
        h = h + _lin(wd, b + ".attn2.to_out.0", a)
        y = F.layer_norm(h, (C,), wd[b + ".norm3.weight"], wd[b + ".norm3.bias"], 1e-5)
        g = _lin(wd, b + ".ff.net.0.proj", y)
        val, gate = g.chunk(2, dim=-1)
        h = h + _lin(wd, b + ".ff.net.2", val * F.gelu(gate))
    h = _lin(wd, n + ".proj_out", h)
    return h.reshape(B, Hh, Ww, C).permute(0, 3, 1, 2) + res


def unet_forward(wd: W, c, sample, timestep, ctx, text_embeds, time_ids):
    """UNet2DConditionModel.forward (epsilon prediction), SDXL 'text_time' added conditioning."""
    boc = c["block_out_channels"]
    G = c["norm_groups"]
    B = sample.shape[0]
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(B)
    temb = timestep_embedding(t, boc[0]).to(sample.dtype)
    emb = _lin(wd, "time_embedding.linear_2", F.silu(_lin(wd, "time_embedding.linear_1", temb)))
    tid = timestep_embedding(time_ids.flatten(), c["addition_time_embed_dim"]).reshape(B, -1).to(sample.dtype)
    add = torch.cat([text_embeds, tid], dim=-1)
    emb = emb + _lin(wd, "add_embedding.linear_2", F.silu(_lin(wd, "add_embedding.linear_1", add)))
    h = F.conv2d(sample, wd["conv_in.weight"], wd["conv_in.bias"], padding=1)
    skips = [h]
    L = c["layers_per_block"]
    for i in range(len(boc)):
        for j in range(L):
            h = _resnet(wd, "down_blocks.%d.resnets.%d" % (i, j), h, emb, G)
            if c["transformer_layers"][i]:
                h = _transformer(wd, "down_blocks.%d.attentions.%d" % (i, j), h, ctx, c["num_heads"][i],
                                 c["transformer_layers"][i], G)
            skips.append(h)
        if i < len(boc) - 1:
            n = "down_blocks.%d.downsamplers.0.conv" % i
            h = F.conv2d(h, wd[n + ".weight"], wd[n + ".bias"], stride=2, padding=1)
            skips.append(h)
    h = _resnet(wd, "mid_block.resnets.0", h, emb, G)
    h = _transformer(wd, "mid_block.attentions.0", h, ctx, c["num_heads"][-1], c["transformer_layers"][-1], G)
    h = _resnet(wd, "mid_block.resnets.1", h, emb, G)
    for i in range(len(boc)):
        ri = len(boc) - 1 - i
        for j in range(L + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = _resnet(wd, "up_blocks.%d.resnets.%d" % (i, j), h, emb, G)
            if c["transformer_layers"][ri]:
                h = _transformer(wd, "up_blocks.%d.attentions.%d" % (i, j), h, ctx, c["num_heads"][ri],
                                 c["transformer_layers"][ri], G)
        if i < len(boc) - 1:
            n = "up_blocks.%d.upsamplers.0.conv" % i
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, wd[n + ".weight"], wd[n + ".bias"], padding=1)
    h = F.silu(_gn(wd, "conv_norm_out", h, G, 1e-5))
    return F.conv2d(h, wd["conv_out.weight"], wd["conv_out.bias"], padding=1)


def vae_decode(wd: W, c, latents):
    """AutoencoderKL.decode(latents / scaling_factor) -> image in [-1, 1] (NCHW)."""
    G = c["norm_groups"]
    boc = c["block_out_channels"]
    z = latents / c["scaling_factor"]
    z = F.conv2d(z, wd["post_quant_conv.weight"], wd["post_quant_conv.bias"])
    h = F.conv2d(z, wd["decoder.conv_in.weight"], wd["decoder.conv_in.bias"], padding=1)
    h = _resnet(wd, "decoder.mid_block.resnets.0", h, None, G, 1e-6)
    a = "decoder.mid_block.attentions.0"
    B, C, Hh, Ww = h.shape
    y = _gn(wd, a + ".group_norm", h, G, 1e-6).reshape(B, C, Hh * Ww).transpose(1, 2)
    o = _attn(_lin(wd, a + ".to_q", y), _lin(wd, a + ".to_k", y), _lin(wd, a + ".to_v", y), 1)
    o = _lin(wd, a + ".to_out.0", o).transpose(1, 2).reshape(B, C, Hh, Ww)
    h = h + o
    h = _resnet(wd, "decoder.mid_block.resnets.1", h, None, G, 1e-6)
    for i in range(len(boc)):
        for j in range(c["layers_per_block"] + 1):
            h = _resnet(wd, "decoder.up_blocks.%d.resnets.%d" % (i, j), h, None, G, 1e-6)
        if i < len(boc) - 1:
            n = "decoder.up_blocks.%d.upsamplers.0.conv" % i
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, wd[n + ".weight"], wd[n + ".bias"], padding=1)
    h = F.silu(_gn(wd, "decoder.conv_norm_out", h, G, 1e-6))
    return F.conv2d(h, wd["decoder.conv_out.weight"], wd["decoder.conv_out.bias"], padding=1)


# ---- EulerDiscreteScheduler + pipeline ---------------------------------------------------------------

def euler_sigmas(num_inference_steps, num_train=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
    """scaled-linear betas, 'leading' timestep spacing (+offset), linearly interpolated sigmas, final 0.
    Returns (timesteps float32 [n], sigmas float32 [n+1], init_noise_sigma)."""
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=np.float32) ** 2
    ac = np.cumprod(1.0 - betas)
    sig = ((1 - ac) / ac) ** 0.5
    ratio = num_train // num_inference_steps
    ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.float32) + steps_offset
    s = np.interp(ts, np.arange(0, len(sig)), sig)
    s = np.concatenate([s, [0.0]]).astype(np.float32)
    init = float((s.max() ** 2 + 1) ** 0.5)
    return torch.from_numpy(ts), torch.from_numpy(s), init


def sdxl_generate_latents(wd: W, c, ctx_pos, ctx_neg, pooled_pos, pooled_neg, noise, steps=30, guidance=7.5,
                          size=1024):
    """StableDiffusionXLPipeline.__call__ as SDXLAdapter.generate drives it:
    latents = noise * init_noise_sigma; per step: [neg; pos] batch, x / sqrt(sigma^2+1), UNet, CFG, Euler."""
    ts, sig, init = euler_sigmas(steps)
    x = noise * init
    time_ids = torch.tensor([[size, size, 0, 0, size, size]] * 2, dtype=noise.dtype)
    ctx = torch.cat([ctx_neg, ctx_pos], dim=0)
    pooled = torch.cat([pooled_neg, pooled_pos], dim=0)
    for i in range(steps):
        xin = torch.cat([x, x], dim=0) / ((sig[i] ** 2 + 1) ** 0.5)
        eps = unet_forward(wd, c, xin, ts[i], ctx, pooled, time_ids)
        eu, ec = eps.chunk(2)
        e = eu + guidance * (ec - eu)
        x = x + e * (sig[i + 1] - sig[i])
    return x


def postprocess(img):
    """(img / 2 + 0.5).clamp(0, 1) -> uint8 HWC, as the pipeline's image processor."""
    x = (img / 2 + 0.5).clamp(0, 1)
    return (x.permute(0, 2, 3, 1) * 255).round().to(torch.uint8)
