"""TEST INFRASTRUCTURE ONLY — CPU restatement of the SDXL de-tokenizer the reference drives through diffusers
(``StableDiffusionXLPipeline`` built text-encoder-less at ``src/models_ipa/adapter_modules.py:369-375`` and called at
``:455-466``; UNet / VAE / scheduler instantiated at ``src/inference/gen_george.py:60-64``).

**Parity unpinned**: diffusers is absent from the reference tree and from this image (unpinned in
``requirements.txt:6``), and the reference has no test that pins a number at this boundary.  This file restates the
*published* SDXL-base architecture and the EulerDiscrete / pipeline semantics.

Independence from the product: nothing here shares code or structure with
``seedstory/diffusion.py``.  The networks are described as a flat *program* (a list of op records built from the
Appendix-B stage table), parameter shapes are derived from that program through per-op parameter tables, and
``unet_forward`` / ``vae_decode`` are interpreters of the program on NCHW torch tensors; the scheduler is computed in
float64 from the closed form.  Structural pins checked in tests/test_sdxl_oracle.py: the SDXL-base program has the
published 2,567,463,684 UNet parameters (2,566,942,084 + the 521,600 of ``add_embedding.linear_1`` rows for the pooled
text embedding as counted in SURVEY Appendix B), 70 transformer blocks and 140 attention calls; the 30-step schedule
is 958, 925, ..., 1 with sigma_0 = 11.4768.

dtype: every function runs in the dtype of the tensors passed in, one torch op per module (conv / norm / linear),
so with bf16 tensors each op's output is rounded to bf16 exactly where a bf16 diffusers run rounds it — the
"same rounding points" reference for the product's bf16 mode.  Never imported by the product path.
"""
import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

W = Dict[str, torch.Tensor]

# ---- configurations -------------------------------------------------------------------------------------------
# keys as the tests pass them to both sides; values of SDXL-base from SURVEY Appendix B (public SDXL-base config)
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

# ---- parameter tables per op kind: (sub-name, kind, dims) ------------------------------------------------------------
#   "n" = affine norm over `a` channels, "c3" / "c1" = conv a -> b with bias, "l" = linear a -> b with bias, "l0" = no bias


def _res_params(cin, cout, temb):
    p = [("norm1", "n", cin, None), ("conv1", "c3", cin, cout), ("norm2", "n", cout, None), ("conv2", "c3", cout, cout)]
    if temb:
        p.append(("time_emb_proj", "l", temb, cout))
    if cin != cout:
        p.append(("conv_shortcut", "c1", cin, cout))
    return p


def _xf_params(ch, depth, xdim):
    p = [("norm", "n", ch, None), ("proj_in", "l", ch, ch), ("proj_out", "l", ch, ch)]
    for d in range(depth):
        b = "transformer_blocks.%d." % d
        p += [(b + "norm1", "n", ch, None), (b + "norm2", "n", ch, None), (b + "norm3", "n", ch, None),
              (b + "attn1.to_q", "l0", ch, ch), (b + "attn1.to_k", "l0", ch, ch), (b + "attn1.to_v", "l0", ch, ch),
              (b + "attn1.to_out.0", "l", ch, ch),
              (b + "attn2.to_q", "l0", ch, ch), (b + "attn2.to_k", "l0", xdim, ch), (b + "attn2.to_v", "l0", xdim, ch),
              (b + "attn2.to_out.0", "l", ch, ch),
              (b + "ff.net.0.proj", "l", ch, 8 * ch), (b + "ff.net.2", "l", 4 * ch, ch)]
    return p


def _emit(shapes, prefix, params):
    for sub, kind, a, b in params:
        n = prefix + "." + sub if prefix else sub
        if kind == "n":
            shapes[n + ".weight"] = (a,)
            shapes[n + ".bias"] = (a,)
        elif kind in ("c3", "c1"):
            k = 3 if kind == "c3" else 1
            shapes[n + ".weight"] = (b, a, k, k)
            shapes[n + ".bias"] = (b,)
        else:
            shapes[n + ".weight"] = (b, a)
            if kind == "l":
                shapes[n + ".bias"] = (b,)


# ---- the UNet as a program ---------------------------------------------------------------------------------------

def unet_program(c) -> List[dict]:
    """Execution-ordered op list of UNet2DConditionModel for config c (SURVEY Appendix B stage table).
    Ops: conv_in | res(name,cin,cout,skip_in) | xf(name,ch,depth,heads) | down(name,ch) | up(name,ch) | push | out."""
    widths = list(c["block_out_channels"])
    depth = list(c["transformer_layers"])
    heads = list(c["num_heads"])
    per = c["layers_per_block"]
    prog = [dict(op="conv_in", cin=c["in_channels"], cout=widths[0]), dict(op="push", ch=widths[0])]
    stack = [widths[0]]                       # channel counts of the skip connections, in push order
    cur = widths[0]
    for s, w in enumerate(widths):            # encoder stages
        for j in range(per):
            prog.append(dict(op="res", name="down_blocks.%d.resnets.%d" % (s, j), cin=cur, cout=w, skip=0))
            cur = w
            if depth[s]:
                prog.append(dict(op="xf", name="down_blocks.%d.attentions.%d" % (s, j), ch=w, depth=depth[s], heads=heads[s]))
            prog.append(dict(op="push", ch=cur))
            stack.append(cur)
        if s + 1 < len(widths):
            prog.append(dict(op="down", name="down_blocks.%d.downsamplers.0.conv" % s, ch=w))
            prog.append(dict(op="push", ch=cur))
            stack.append(cur)
    prog.append(dict(op="res", name="mid_block.resnets.0", cin=cur, cout=cur, skip=0))
    prog.append(dict(op="xf", name="mid_block.attentions.0", ch=cur, depth=depth[-1], heads=heads[-1]))
    prog.append(dict(op="res", name="mid_block.resnets.1", cin=cur, cout=cur, skip=0))
    for u, s in enumerate(reversed(range(len(widths)))):     # decoder stages mirror the encoder
        w = widths[s]
        for j in range(per + 1):
            sk = stack.pop()
            prog.append(dict(op="res", name="up_blocks.%d.resnets.%d" % (u, j), cin=cur + sk, cout=w, skip=sk))
            cur = w
            if depth[s]:
                prog.append(dict(op="xf", name="up_blocks.%d.attentions.%d" % (u, j), ch=w, depth=depth[s], heads=heads[s]))
        if s > 0:
            prog.append(dict(op="up", name="up_blocks.%d.upsamplers.0.conv" % u, ch=w))
    assert not stack
    prog.append(dict(op="out", cin=cur, cout=c["out_channels"]))
    return prog


def unet_shapes(c) -> Dict[str, tuple]:
    """diffusers parameter name -> shape, derived from the program."""
    temb = 4 * c["block_out_channels"][0]
    s: Dict[str, tuple] = {}
    _emit(s, "", [("time_embedding.linear_1", "l", c["block_out_channels"][0], temb),
                  ("time_embedding.linear_2", "l", temb, temb),
                  ("add_embedding.linear_1", "l", 6 * c["addition_time_embed_dim"] + c["pooled_dim"], temb),
                  ("add_embedding.linear_2", "l", temb, temb)])
    for o in unet_program(c):
        if o["op"] == "conv_in":
            _emit(s, "", [("conv_in", "c3", o["cin"], o["cout"])])
        elif o["op"] == "res":
            _emit(s, o["name"], _res_params(o["cin"], o["cout"], temb))
        elif o["op"] == "xf":
            _emit(s, o["name"], _xf_params(o["ch"], o["depth"], c["cross_attention_dim"]))
        elif o["op"] in ("down", "up"):
            _emit(s, "", [(o["name"], "c3", o["ch"], o["ch"])])
        elif o["op"] == "out":
            _emit(s, "", [("conv_norm_out", "n", o["cin"], None), ("conv_out", "c3", o["cin"], o["cout"])])
    return s


def unet_stats(c) -> dict:
    prog = unet_program(c)
    blocks = sum(o["depth"] for o in prog if o["op"] == "xf")
    return dict(params=sum(int(np.prod(v)) for v in unet_shapes(c).values()), transformer_blocks=blocks,
                attention_calls=2 * blocks, resnets=sum(o["op"] == "res" for o in prog))


def unet_param_count(c) -> int:
    return unet_stats(c)["params"]


def vae_program(c) -> List[dict]:
    widths = list(c["block_out_channels"])
    top = widths[-1]
    prog = [dict(op="pq", ch=c["latent_channels"]), dict(op="conv_in", cin=c["latent_channels"], cout=top),
            dict(op="res", name="decoder.mid_block.resnets.0", cin=top, cout=top),
            dict(op="attn", name="decoder.mid_block.attentions.0", ch=top),
            dict(op="res", name="decoder.mid_block.resnets.1", cin=top, cout=top)]
    cur = top
    for u, w in enumerate(reversed(widths)):
        for j in range(c["layers_per_block"] + 1):
            prog.append(dict(op="res", name="decoder.up_blocks.%d.resnets.%d" % (u, j), cin=cur, cout=w))
            cur = w
        if u + 1 < len(widths):
            prog.append(dict(op="up", name="decoder.up_blocks.%d.upsamplers.0.conv" % u, ch=w))
    prog.append(dict(op="out", cin=cur, cout=c["out_channels"]))
    return prog


def vae_decoder_shapes(c) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}
    for o in vae_program(c):
        if o["op"] == "pq":
            _emit(s, "", [("post_quant_conv", "c1", o["ch"], o["ch"])])
        elif o["op"] == "conv_in":
            _emit(s, "", [("decoder.conv_in", "c3", o["cin"], o["cout"])])
        elif o["op"] == "res":
            _emit(s, o["name"], _res_params(o["cin"], o["cout"], 0))
        elif o["op"] == "attn":
            ch = o["ch"]
            _emit(s, o["name"], [("group_norm", "n", ch, None), ("to_q", "l", ch, ch), ("to_k", "l", ch, ch),
                                 ("to_v", "l", ch, ch), ("to_out.0", "l", ch, ch)])
        elif o["op"] == "up":
            _emit(s, "", [(o["name"], "c3", o["ch"], o["ch"])])
        elif o["op"] == "out":
            _emit(s, "", [("decoder.conv_norm_out", "n", o["cin"], None), ("decoder.conv_out", "c3", o["cin"], o["cout"])])
    return s


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


# ---- module semantics (NCHW, one torch op per module) ----------------------------------------------------------------

def sinusoid(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin] halves, fp32."""
    k = torch.arange(dim // 2, dtype=torch.float32)
    ang = t.float().reshape(-1, 1) * torch.exp(-math.log(10000.0) * k / (dim // 2)).reshape(1, -1)
    return torch.cat([ang.cos(), ang.sin()], dim=1)


timestep_embedding = sinusoid


def linear(wd, name, x):
    return F.linear(x, wd[name + ".weight"], wd.get(name + ".bias"))


def group_norm(wd, name, x, groups, eps):
    return F.group_norm(x, groups, wd[name + ".weight"], wd[name + ".bias"], eps)


def resnet_block(wd, name, x, temb, groups, eps=1e-5):
    """diffusers ResnetBlock2D (time_embedding_norm='default', output_scale_factor 1)."""
    h = F.conv2d(F.silu(group_norm(wd, name + ".norm1", x, groups, eps)), wd[name + ".conv1.weight"],
                 wd[name + ".conv1.bias"], padding=1)
    if temb is not None:
        h = h + linear(wd, name + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.conv2d(F.silu(group_norm(wd, name + ".norm2", h, groups, eps)), wd[name + ".conv2.weight"],
                 wd[name + ".conv2.bias"], padding=1)
    if (name + ".conv_shortcut.weight") in wd:
        x = F.conv2d(x, wd[name + ".conv_shortcut.weight"], wd[name + ".conv_shortcut.bias"])
    return x + h


def mha_nomask(q, k, v, heads):
    """softmax(q k^T / sqrt(d)) v per head with fp32 softmax; q [B,Lq,C], k/v [B,Lk,C] -> [B,Lq,C] in q.dtype."""
    B, Lq, C = q.shape
    d = C // heads
    qh, kh, vh = (t.reshape(B, -1, heads, d).permute(0, 2, 1, 3).float() for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(2, 3) * (d ** -0.5), dim=-1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B, Lq, C).to(q.dtype)


def transformer_2d(wd, name, x, ctx, heads, depth, groups):
    """diffusers Transformer2DModel (use_linear_projection=True) of BasicTransformerBlocks (GEGLU feed-forward)."""
    B, C, Hh, Ww = x.shape
    h = group_norm(wd, name + ".norm", x, groups, 1e-6).permute(0, 2, 3, 1).reshape(B, Hh * Ww, C)
    h = linear(wd, name + ".proj_in", h)
    for d in range(depth):
        b = "%s.transformer_blocks.%d" % (name, d)
        y = F.layer_norm(h, (C,), wd[b + ".norm1.weight"], wd[b + ".norm1.bias"], 1e-5)
        h = h + linear(wd, b + ".attn1.to_out.0", mha_nomask(linear(wd, b + ".attn1.to_q", y), linear(wd, b + ".attn1.to_k", y),
                                                             linear(wd, b + ".attn1.to_v", y), heads))
        y = F.layer_norm(h, (C,), wd[b + ".norm2.weight"], wd[b + ".norm2.bias"], 1e-5)
        h = h + linear(wd, b + ".attn2.to_out.0", mha_nomask(linear(wd, b + ".attn2.to_q", y), linear(wd, b + ".attn2.to_k", ctx),
                                                             linear(wd, b + ".attn2.to_v", ctx), heads))
        y = F.layer_norm(h, (C,), wd[b + ".norm3.weight"], wd[b + ".norm3.bias"], 1e-5)
        val, gate = linear(wd, b + ".ff.net.0.proj", y).chunk(2, dim=-1)
        h = h + linear(wd, b + ".ff.net.2", val * F.gelu(gate))
    h = linear(wd, name + ".proj_out", h)
    return h.reshape(B, Hh, Ww, C).permute(0, 3, 1, 2) + x


def unet_forward(wd: W, c, sample, timestep, ctx, text_embeds, time_ids):
    """UNet2DConditionModel.forward (epsilon prediction), SDXL 'text_time' added conditioning: interpreter of
    unet_program(c)."""
    G = c["norm_groups"]
    B = sample.shape[0]
    w0 = c["block_out_channels"][0]
    t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(B)
    emb = linear(wd, "time_embedding.linear_2", F.silu(linear(wd, "time_embedding.linear_1", sinusoid(t, w0).to(sample.dtype))))
    tid = sinusoid(time_ids.flatten(), c["addition_time_embed_dim"]).reshape(B, -1).to(sample.dtype)
    aug = linear(wd, "add_embedding.linear_2", F.silu(linear(wd, "add_embedding.linear_1", torch.cat([text_embeds, tid], dim=-1))))
    emb = emb + aug
    h = None
    stack = []
    for o in unet_program(c):
        k = o["op"]
        if k == "conv_in":
            h = F.conv2d(sample, wd["conv_in.weight"], wd["conv_in.bias"], padding=1)
        elif k == "push":
            stack.append(h)
        elif k == "res":
            if o["skip"]:
                h = torch.cat([h, stack.pop()], dim=1)
            h = resnet_block(wd, o["name"], h, emb, G)
        elif k == "xf":
            h = transformer_2d(wd, o["name"], h, ctx, o["heads"], o["depth"], G)
        elif k == "down":
            h = F.conv2d(h, wd[o["name"] + ".weight"], wd[o["name"] + ".bias"], stride=2, padding=1)
        elif k == "up":
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), wd[o["name"] + ".weight"],
                         wd[o["name"] + ".bias"], padding=1)
        elif k == "out":
            h = F.conv2d(F.silu(group_norm(wd, "conv_norm_out", h, G, 1e-5)), wd["conv_out.weight"], wd["conv_out.bias"],
                         padding=1)
    return h


def vae_mid_attention(wd, name, h, groups):
    B, C, Hh, Ww = h.shape
    y = group_norm(wd, name + ".group_norm", h, groups, 1e-6).reshape(B, C, Hh * Ww).transpose(1, 2)
    o = mha_nomask(linear(wd, name + ".to_q", y), linear(wd, name + ".to_k", y), linear(wd, name + ".to_v", y), 1)
    return h + linear(wd, name + ".to_out.0", o).transpose(1, 2).reshape(B, C, Hh, Ww)


def vae_decode(wd: W, c, latents):
    """AutoencoderKL.decode(latents / scaling_factor) -> image in [-1, 1] (NCHW): interpreter of vae_program(c)."""
    G = c["norm_groups"]
    h = latents / c["scaling_factor"]
    for o in vae_program(c):
        k = o["op"]
        if k == "pq":
            h = F.conv2d(h, wd["post_quant_conv.weight"], wd["post_quant_conv.bias"])
        elif k == "conv_in":
            h = F.conv2d(h, wd["decoder.conv_in.weight"], wd["decoder.conv_in.bias"], padding=1)
        elif k == "res":
            h = resnet_block(wd, o["name"], h, None, G, 1e-6)
        elif k == "attn":
            h = vae_mid_attention(wd, o["name"], h, G)
        elif k == "up":
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), wd[o["name"] + ".weight"],
                         wd[o["name"] + ".bias"], padding=1)
        elif k == "out":
            h = F.conv2d(F.silu(group_norm(wd, "decoder.conv_norm_out", h, G, 1e-6)), wd["decoder.conv_out.weight"],
                         wd["decoder.conv_out.bias"], padding=1)
    return h


# ---- EulerDiscreteScheduler + pipeline ---------------------------------------------------------------------------

def euler_schedule(n_steps: int, n_train: int = 1000, beta_lo: float = 0.00085, beta_hi: float = 0.012, offset: int = 1):
    """SDXL-base scheduler config: scaled-linear betas, 'leading' spacing + steps_offset, epsilon
    prediction, no churn.  Closed form in float64: beta_i = (sqrt(lo) + i (sqrt(hi) - sqrt(lo)) / (T-1))^2,
    abar_t = prod_{i<=t} (1 - beta_i), sigma_t = sqrt(1/abar_t - 1); the inference timesteps t_k = (n-1-k) * (T // n)
    + offset are integers, so no interpolation is involved.  Returns (timesteps [n] ints, sigmas [n+1] with a final 0,
    init_noise_sigma = sqrt(sigma_0^2 + 1))."""
    root_lo, root_hi = math.sqrt(beta_lo), math.sqrt(beta_hi)
    abar, log_abar = [], 0.0
    for i in range(n_train):
        beta = (root_lo + (root_hi - root_lo) * i / (n_train - 1)) ** 2
        log_abar += math.log1p(-beta)
        abar.append(math.exp(log_abar))
    stride = n_train // n_steps
    ts = [(n_steps - 1 - k) * stride + offset for k in range(n_steps)]
    sig = [math.sqrt(1.0 / abar[t] - 1.0) for t in ts] + [0.0]
    return ts, sig, math.sqrt(sig[0] ** 2 + 1.0)


def euler_sigmas(num_inference_steps):
    ts, sig, init = euler_schedule(num_inference_steps)
    return torch.tensor(ts, dtype=torch.float32), torch.tensor(sig, dtype=torch.float32), init


def sdxl_generate_latents(wd: W, c, ctx_pos, ctx_neg, pooled_pos, pooled_neg, noise, steps=30, guidance=7.5, size=1024):
    """StableDiffusionXLPipeline.__call__ as SDXLAdapter.generate drives it: latents = noise *
    init_noise_sigma; per step: [neg; pos] batch scaled by 1/sqrt(sigma^2+1), UNet, classifier-free guidance,
    Euler step x += eps * (sigma_next - sigma)."""
    ts, sig, init = euler_schedule(steps)
    x = noise * init
    ids = torch.tensor([[size, size, 0, 0, size, size]] * 2, dtype=noise.dtype)
    ctx = torch.cat([ctx_neg, ctx_pos], dim=0)
    pooled = torch.cat([pooled_neg, pooled_pos], dim=0)
    for k in range(steps):
        xin = torch.cat([x, x], dim=0) / math.sqrt(sig[k] ** 2 + 1.0)
        e_neg, e_pos = unet_forward(wd, c, xin, float(ts[k]), ctx, pooled, ids).chunk(2)
        x = x + (e_neg + guidance * (e_pos - e_neg)) * (sig[k + 1] - sig[k])
    return x


def postprocess(img):
    """(img / 2 + 0.5).clamp(0, 1) -> uint8 HWC, as the pipeline's image processor."""
    return ((img / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1) * 255).round().to(torch.uint8)


def adapter_image_embeds(xlv2_wd: W, xlv2_cfg: dict, feat_pos: torch.Tensor, feat_neg: torch.Tensor):
    """SDXLAdapter.get_image_embeds(image_embeds=..., return_negative=True) — src/models_ipa/adapter_modules.py:387-428:
    the regressed feature and the feature of an all-zeros image (:406-414, supplied by the caller: it is the ViT of a
    zero tensor) are concatenated [pos; neg], sent through ResamplerXLV2 TOGETHER (F.normalize over dim=1 acts per
    sample, resampler.py:269) and split back with chunk(2) (:420-422).
    Returns (ctx_pos, ctx_neg, pooled_pos, pooled_neg)."""
    import seedstory_oracle as O
    both = torch.cat([feat_pos, feat_neg], dim=0)
    ctx, pooled = O.resampler_xlv2_forward(xlv2_wd, both, depth=xlv2_cfg["depth"], heads=xlv2_cfg["heads"],
                                           dim_head=xlv2_cfg["dim_head"])
    ctx_pos, ctx_neg = ctx.chunk(2)
    pooled_pos, pooled_neg = pooled.chunk(2)
    return ctx_pos, ctx_neg, pooled_pos, pooled_neg
