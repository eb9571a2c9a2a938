"""TEST INFRASTRUCTURE ONLY — a second, class-based statement of the SDXL de-tokenizer networks, used to cross-check
``sdxl_oracle.py`` while no diffusers-generated fixture exists ("sdxl_oracle blocks vs
torch.nn.GroupNorm / Conv2d / MultiheadAttention-built modules written from the published config by name").

What the reference calls (diffusers is a pip dependency, ``/root/reference/requirements.txt:6``, absent here):
``UNet2DConditionModel`` / ``AutoencoderKL`` / ``EulerDiscreteScheduler`` at ``src/inference/gen_george.py:10,60-64``,
driven through ``StableDiffusionXLPipeline`` at ``src/models_ipa/adapter_modules.py:369-375,455-466``.

How this differs from ``sdxl_oracle.py`` (so a misreading there does not repeat here by construction):
  * the network is a tree of ``torch.nn`` modules named like the published checkpoints' keys and assembled from the
    published ``config.json`` keys of stabilityai/stable-diffusion-xl-base-1.0 (``down_block_types``,
    ``transformer_layers_per_block``, ``attention_head_dim`` …, restated in ``SDXL_UNET_CONFIG`` / ``SDXL_VAE_CONFIG``
    below), not a flat op program built from SURVEY's stage table;
  * the oracle's tensors enter through ``load_state_dict(strict=True)``: a wrong name, a wrong shape, a missing or
    surplus bias is a load error;
  * attention is ``torch.nn.MultiheadAttention`` (packed in-projection assembled from to_q / to_k / to_v, the output
    projection is the module's ``out_proj``) for self-attention and ``F.scaled_dot_product_attention`` for the
    cross-attention (different key width), never the oracle's explicit softmax;
  * normalisations are ``nn.GroupNorm`` / ``nn.LayerNorm`` module instances, the feed-forward is an explicit GEGLU
    module, the sinusoid is built from ``torch.outer`` in float64.
It is still written by the same hands, so the status of the de-tokenizer oracle stays **parity unpinned** until
``oracle/make_golden_sdxl_diffusers.py`` has been run on a box that has diffusers (tests/test_sdxl_pin.py).
Never imported by the product path.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# Published config.json values (stabilityai/stable-diffusion-xl-base-1.0, unet/ and vae/), by their diffusers key names.
SDXL_UNET_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=[320, 640, 1280], layers_per_block=2,
    down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
    up_block_types=["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"],
    transformer_layers_per_block=[1, 2, 10], attention_head_dim=[5, 10, 20],   # (the key holds the number of heads)
    cross_attention_dim=2048, addition_embed_type="text_time", addition_time_embed_dim=256,
    projection_class_embeddings_input_dim=2816, use_linear_projection=True, norm_num_groups=32, norm_eps=1e-5,
    flip_sin_to_cos=True, freq_shift=0, act_fn="silu", downsample_padding=1)
SDXL_VAE_CONFIG = dict(
    latent_channels=4, out_channels=3, block_out_channels=[128, 256, 512, 512], layers_per_block=2,
    up_block_types=["UpDecoderBlock2D"] * 4, norm_num_groups=32, scaling_factor=0.13025, force_upcast=True)


def config_from_oracle(c) -> dict:
    """The oracle's compact dict (sdxl_oracle.SDXL_BASE_UNET / TINY_UNET) -> published-key config."""
    n = len(c["block_out_channels"])
    depth = list(c["transformer_layers"])
    return dict(SDXL_UNET_CONFIG, in_channels=c["in_channels"], out_channels=c["out_channels"],
                block_out_channels=list(c["block_out_channels"]), layers_per_block=c["layers_per_block"],
                down_block_types=["CrossAttnDownBlock2D" if d else "DownBlock2D" for d in depth],
                up_block_types=["CrossAttnUpBlock2D" if d else "UpBlock2D" for d in reversed(depth)],
                transformer_layers_per_block=[max(d, 1) for d in depth], attention_head_dim=list(c["num_heads"])[:n],
                cross_attention_dim=c["cross_attention_dim"], addition_time_embed_dim=c["addition_time_embed_dim"],
                projection_class_embeddings_input_dim=6 * c["addition_time_embed_dim"] + c["pooled_dim"],
                norm_num_groups=c["norm_groups"])


def vae_config_from_oracle(c) -> dict:
    return dict(SDXL_VAE_CONFIG, latent_channels=c["latent_channels"], out_channels=c["out_channels"],
                block_out_channels=list(c["block_out_channels"]), layers_per_block=c["layers_per_block"],
                up_block_types=["UpDecoderBlock2D"] * len(c["block_out_channels"]), norm_num_groups=c["norm_groups"],
                scaling_factor=c["scaling_factor"])


# ---- leaves -----------------------------------------------------------------------------------------------------

class Timesteps(nn.Module):
    """Sinusoidal projection.  Published semantics: exponent_i = -ln(10000)·i / (half - freq_shift), the embedding is
    [sin | cos] and ``flip_sin_to_cos`` swaps the halves."""

    def __init__(self, num_channels, flip_sin_to_cos, freq_shift):
        super().__init__()
        self.num_channels, self.flip, self.shift = num_channels, flip_sin_to_cos, freq_shift

    def forward(self, t):
        half = self.num_channels // 2
        freqs = torch.exp(torch.arange(half, dtype=torch.float64) * (-math.log(10000.0) / (half - self.shift)))
        arg = torch.outer(t.to(torch.float64).flatten(), freqs)
        sin, cos = arg.sin(), arg.cos()
        out = torch.cat([cos, sin], dim=-1) if self.flip else torch.cat([sin, cos], dim=-1)
        return out.to(torch.float32)


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_channels, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        if temb_channels:
            self.time_emb_proj = nn.Linear(temb_channels, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if temb is not None:
            h = h + self.time_emb_proj(F.silu(temb)).unsqueeze(-1).unsqueeze(-1)
        h = self.conv2(F.silu(self.norm2(h)))
        skip = self.conv_shortcut(x) if hasattr(self, "conv_shortcut") else x
        return skip + h


class Downsample2D(nn.Module):
    def __init__(self, ch, padding):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=padding)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Attention(nn.Module):
    """Parameters under the published names (to_q / to_k / to_v / to_out.0); the arithmetic is torch's own:
    ``nn.MultiheadAttention`` when query and key widths agree, ``F.scaled_dot_product_attention`` otherwise."""

    def __init__(self, query_dim, heads, cross_dim=None, qkv_bias=False):
        super().__init__()
        self.heads = heads
        kd = cross_dim or query_dim
        self.to_q = nn.Linear(query_dim, query_dim, bias=qkv_bias)
        self.to_k = nn.Linear(kd, query_dim, bias=qkv_bias)
        self.to_v = nn.Linear(kd, query_dim, bias=qkv_bias)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim), nn.Identity()])   # [linear, dropout]

    def forward(self, x, context=None):
        C = self.to_q.in_features
        if context is None:
            mha = nn.MultiheadAttention(C, self.heads, bias=True, batch_first=True, dtype=x.dtype)
            zeros = torch.zeros(C, dtype=x.dtype)
            with torch.no_grad():
                mha.in_proj_weight.copy_(torch.cat([self.to_q.weight, self.to_k.weight, self.to_v.weight]))
                mha.in_proj_bias.copy_(torch.cat([zeros if l.bias is None else l.bias for l in (self.to_q, self.to_k, self.to_v)]))
                mha.out_proj.weight.copy_(self.to_out[0].weight)
                mha.out_proj.bias.copy_(self.to_out[0].bias)
            return mha.eval()(x, x, x, need_weights=False)[0]
        B, L, _ = x.shape
        d = C // self.heads

        def split(t):
            return t.reshape(B, -1, self.heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(split(self.to_q(x)), split(self.to_k(context)), split(self.to_v(context)))
        return self.to_out[0](o.transpose(1, 2).reshape(B, L, C))


class GEGLU(nn.Module):
    """proj to 2·inner; the FIRST half is the value, the SECOND half goes through GELU (erf form) as the gate."""

    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, 2 * inner)

    def forward(self, x):
        y = self.proj(x)
        inner = y.shape[-1] // 2
        return y[..., :inner] * F.gelu(y[..., inner:], approximate="none")


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Identity(), nn.Linear(4 * dim, dim)])   # [GEGLU, dropout, linear]

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, cross_dim=cross_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, context):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), context)
        return x + self.ff(self.norm3(x))


class Transformer2DModel(nn.Module):
    def __init__(self, ch, heads, depth, cross_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.proj_in = nn.Linear(ch, ch)                 # use_linear_projection
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, heads, cross_dim) for _ in range(depth)])
        self.proj_out = nn.Linear(ch, ch)

    def forward(self, x, context):
        B, C, H, Wd = x.shape
        h = self.norm(x).flatten(2).transpose(1, 2)      # [B, H·W, C]: the linear projection acts on tokens
        h = self.proj_in(h)
        for blk in self.transformer_blocks:
            h = blk(h, context)
        h = self.proj_out(h)
        return h.transpose(1, 2).reshape(B, C, H, Wd) + x


# ---- UNet blocks ----------------------------------------------------------------------------------------------------

class DownBlock(nn.Module):
    def __init__(self, kind, cin, cout, temb, n_layers, add_down, heads, depth, cfg):
        super().__init__()
        g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, g, eps) for i in range(n_layers)])
        if kind == "CrossAttnDownBlock2D":
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, depth, cfg["cross_attention_dim"], g)
                                             for _ in range(n_layers)])
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout, cfg["downsample_padding"])])

    def forward(self, x, temb, context):
        outs = []
        for i, res in enumerate(self.resnets):
            x = res(x, temb)
            if hasattr(self, "attentions"):
                x = self.attentions[i](x, context)
            outs.append(x)
        if hasattr(self, "downsamplers"):
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, ch, temb, heads, depth, cfg):
        super().__init__()
        g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb, g, eps), ResnetBlock2D(ch, ch, temb, g, eps)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, heads, depth, cfg["cross_attention_dim"], g)])

    def forward(self, x, temb, context):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, context)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, kind, cin, cout, cprev, temb, n_layers, add_up, heads, depth, cfg):
        """``cin`` = width of the encoder block at the mirrored position (its first skip comes from the block below
        it), ``cprev`` = width arriving from the decoder block before this one."""
        super().__init__()
        g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        self.resnets = nn.ModuleList()
        for i in range(n_layers):
            skip_ch = cin if i == n_layers - 1 else cout
            self.resnets.append(ResnetBlock2D((cprev if i == 0 else cout) + skip_ch, cout, temb, g, eps))
        if kind == "CrossAttnUpBlock2D":
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, depth, cfg["cross_attention_dim"], g)
                                             for _ in range(n_layers)])
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])

    def forward(self, x, skips, temb, context):
        for i, res in enumerate(self.resnets):
            x = res(torch.cat([x, skips.pop()], dim=1), temb)     # decoder tensor first, skip second
            if hasattr(self, "attentions"):
                x = self.attentions[i](x, context)
        if hasattr(self, "upsamplers"):
            x = self.upsamplers[0](x)
        return x


class UNet2DConditionModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        widths = cfg["block_out_channels"]
        temb = 4 * widths[0]
        self.conv_in = nn.Conv2d(cfg["in_channels"], widths[0], 3, padding=1)
        self.time_proj = Timesteps(widths[0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
        self.time_embedding = TimestepEmbedding(widths[0], temb)
        self.add_time_proj = Timesteps(cfg["addition_time_embed_dim"], cfg["flip_sin_to_cos"], cfg["freq_shift"])
        self.add_embedding = TimestepEmbedding(cfg["projection_class_embeddings_input_dim"], temb)
        heads, depth, per = cfg["attention_head_dim"], cfg["transformer_layers_per_block"], cfg["layers_per_block"]
        self.down_blocks = nn.ModuleList()
        cout = widths[0]
        for i, kind in enumerate(cfg["down_block_types"]):
            cin, cout = cout, widths[i]
            self.down_blocks.append(DownBlock(kind, cin, cout, temb, per, i + 1 < len(widths), heads[i], depth[i], cfg))
        self.mid_block = MidBlock(widths[-1], temb, heads[-1], depth[-1], cfg)
        self.up_blocks = nn.ModuleList()
        rw, rh, rd = widths[::-1], heads[::-1], depth[::-1]
        cout = rw[0]
        for i, kind in enumerate(cfg["up_block_types"]):
            cprev, cout = cout, rw[i]
            cin = rw[min(i + 1, len(widths) - 1)]
            self.up_blocks.append(UpBlock(kind, cin, cout, cprev, temb, per + 1, i + 1 < len(widths), rh[i], rd[i], cfg))
        self.conv_norm_out = nn.GroupNorm(cfg["norm_num_groups"], widths[0], eps=cfg["norm_eps"])
        self.conv_out = nn.Conv2d(widths[0], cfg["out_channels"], 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, text_embeds, time_ids):
        B = sample.shape[0]
        t = torch.as_tensor(timestep, dtype=torch.float32).flatten().expand(B)
        emb = self.time_embedding(self.time_proj(t).to(sample.dtype))
        tid = self.add_time_proj(time_ids.flatten()).reshape(B, -1).to(sample.dtype)
        emb = emb + self.add_embedding(torch.cat([text_embeds, tid], dim=-1))
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states)
            skips += outs
        x = self.mid_block(x, emb, encoder_hidden_states)
        for blk in self.up_blocks:
            x = blk(x, skips, emb, encoder_hidden_states)
        assert not skips
        return self.conv_out(F.silu(self.conv_norm_out(x)))


# ---- VAE decoder ----------------------------------------------------------------------------------------------------

class VaeMidBlock(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, 0, groups, 1e-6), ResnetBlock2D(ch, ch, 0, groups, 1e-6)])
        att = Attention(ch, 1, qkv_bias=True)            # one head of width ch, residual connection, biased q/k/v
        att.group_norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.attentions = nn.ModuleList([att])

    def forward(self, x):
        x = self.resnets[0](x)
        a = self.attentions[0]
        B, C, H, Wd = x.shape
        x = x + a(a.group_norm(x).flatten(2).transpose(1, 2)).transpose(1, 2).reshape(B, C, H, Wd)
        return self.resnets[1](x)


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n_layers, add_up, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, 0, groups, 1e-6) for i in range(n_layers)])
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.upsamplers[0](x) if hasattr(self, "upsamplers") else x


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        widths, g = cfg["block_out_channels"], cfg["norm_num_groups"]
        self.conv_in = nn.Conv2d(cfg["latent_channels"], widths[-1], 3, padding=1)
        self.mid_block = VaeMidBlock(widths[-1], g)
        self.up_blocks = nn.ModuleList()
        rw = widths[::-1]
        cout = rw[0]
        for i in range(len(rw)):
            cin, cout = cout, rw[i]
            self.up_blocks.append(UpDecoderBlock2D(cin, cout, cfg["layers_per_block"] + 1, i + 1 < len(rw), g))
        self.conv_norm_out = nn.GroupNorm(g, widths[0], eps=1e-6)
        self.conv_out = nn.Conv2d(widths[0], cfg["out_channels"], 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKLDecoder(nn.Module):
    """decode(): post_quant_conv then the decoder; the pipeline divides the latents by ``scaling_factor`` first."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.post_quant_conv = nn.Conv2d(cfg["latent_channels"], cfg["latent_channels"], 1)
        self.decoder = Decoder(cfg)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


# ---- EulerDiscreteScheduler, array form ---------------------------------------------------------------------------

def euler_tables(n_steps, n_train=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
    """Published scheduler_config.json of SDXL-base: scaled_linear betas, timestep_spacing 'leading', steps_offset 1,
    prediction_type epsilon, interpolation_type linear, no Karras sigmas.  Array formulation (cumprod / interp), as
    opposed to the oracle's scalar closed form."""
    import numpy as np
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n_train, dtype=np.float64) ** 2
    ac = np.cumprod(1.0 - betas)
    sig_all = np.sqrt((1.0 - ac) / ac)
    ts = (np.arange(0, n_steps) * (n_train // n_steps)).round()[::-1].astype(np.float64) + steps_offset
    sig = np.interp(ts, np.arange(n_train), sig_all)
    sig = np.concatenate([sig, [0.0]])
    return ts, sig, float((sig.max() ** 2 + 1.0) ** 0.5)     # 'leading' spacing: init sigma = sqrt(max^2 + 1)
