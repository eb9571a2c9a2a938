"""MI355X-native SDXL de-tokenizer: stand-ins for the four diffusers classes the reference
instantiates (``src/inference/gen_george.py:10,60-64``; pipeline built at
``src/models_ipa/adapter_modules.py:369-375`` and called at ``:455-466``):

    EulerDiscreteScheduler.from_pretrained(path, subfolder="scheduler")
    AutoencoderKL.from_pretrained(path, subfolder="vae")
    UNet2DConditionModel.from_pretrained(path, subfolder="unet")
    StableDiffusionXLPipeline(vae=, unet=, scheduler=, tokenizer=None, text_encoder=None, ...)

diffusers itself is absent from this image.  The modules keep the diffusers parameter names (so
``diffusion_pytorch_model.safetensors`` and the de-tokenizer checkpoint's ``unet.*`` keys load with
``load_state_dict``) and run entirely on the HIP kernels of libseedstory_hip.so: activations are NHWC,
3x3 convolutions are implicit GEMMs on the matrix cores (``ss_conv3x3``: fused bias / time-embedding /
residual / nearest-2x upsample), GroupNorm+SiLU, fused-QKV flash attention, GEGLU, and the Euler/CFG
update are hand-written kernels.  No torch compute on the device, no CPU path.
"""
import json
import math
import os

import numpy as np
import torch
from torch import nn

from . import _lib, ops

SDXL_BASE_UNET = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
                      transformer_layers=(0, 2, 10), num_heads=(5, 10, 20), cross_attention_dim=2048,
                      addition_time_embed_dim=256, pooled_dim=1280, norm_groups=32)
SDXL_BASE_VAE = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                     norm_groups=32, scaling_factor=0.13025)


# ---- parameter tree with diffusers key names ----------------------------------------------------------

def _unet_shapes(c):
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
            for pn in ("to_q", "to_k", "to_v"):
                lin(b + ".attn1." + pn, ch, ch, bias=False)
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


def _vae_shapes(c):
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
    for pn in ("to_q", "to_k", "to_v", "to_out.0"):
        s[a + "." + pn + ".weight"] = (top, top); s[a + "." + pn + ".bias"] = (top,)
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


class _Node(nn.Module):
    """Bare module used to grow a parameter tree whose state_dict keys equal the diffusers names."""


def _grow(root, shapes):
    for key, shp in shapes.items():
        parts = key.split(".")
        m = root
        for part in parts[:-1]:
            if part not in m._modules:
                m.add_module(part, _Node())
            m = m._modules[part]
        m.register_parameter(parts[-1], nn.Parameter(torch.empty(*shp), requires_grad=False))


def _init_synthetic(module, seed):
    for i, (name, prm) in enumerate(module.named_parameters()):
        if prm.dim() == 1:
            prm.data.fill_(1.0 if name.endswith("weight") else 0.0)
        else:
            fan_in = int(np.prod(prm.shape[1:]))
            if prm.is_cuda:
                prm.data.normal_(0.0, 1.0 / math.sqrt(fan_in))
            else:
                g = torch.Generator().manual_seed(seed * 7919 + i)
                prm.data.copy_(torch.randn(prm.shape, generator=g) / math.sqrt(fan_in))
    return module


def _sig(module):
    p0 = next(module.parameters())
    return (p0.data_ptr(), p0._version, p0.dtype, str(p0.device), sum(p._version for p in module.parameters()))


def _conv_w(w, cpad=None):
    """[Co, Ci, 3, 3] -> [Co, 9 * Ci(pad)] with k = (ky*3 + kx) * Ci + ci  (host-side re-layout, once)."""
    co, ci = w.shape[0], w.shape[1]
    cp = ci if cpad is None else cpad
    t = torch.zeros(co, 3, 3, cp, dtype=w.dtype, device=w.device)
    t[..., :ci] = w.permute(0, 2, 3, 1)
    return t.reshape(co, 9 * cp).contiguous()


def timestep_embedding(t, dim):
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0), fp32 on the host
    (a 1x320 vector per denoising step: scalar pre-processing, not device work)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t.float().reshape(-1)[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class _Sample:
    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


# ---- UNet ------------------------------------------------------------------------------------------------

class UNet2DConditionModel(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        self.cfg = dict(SDXL_BASE_UNET if config is None else config)
        _grow(self, _unet_shapes(self.cfg))
        self._prep = None
        self._prep_sig = None

    @property
    def config(self):
        return self.cfg

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **kw):
        folder = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(folder, "config.json")) as f:
            jc = json.load(f)
        tl = jc.get("transformer_layers_per_block", 1)
        boc = tuple(jc["block_out_channels"])
        types = jc.get("down_block_types", ())
        tl = tuple(tl) if isinstance(tl, (list, tuple)) else (tl,) * len(boc)
        tl = tuple(t if "CrossAttn" in types[i] else 0 for i, t in enumerate(tl)) if types else tl
        heads = jc.get("attention_head_dim", 8)
        heads = tuple(heads) if isinstance(heads, (list, tuple)) else (heads,) * len(boc)
        cfg = dict(in_channels=jc.get("in_channels", 4), out_channels=jc.get("out_channels", 4), block_out_channels=boc,
                   layers_per_block=jc.get("layers_per_block", 2), transformer_layers=tl, num_heads=heads,
                   cross_attention_dim=jc.get("cross_attention_dim", 2048),
                   addition_time_embed_dim=jc.get("addition_time_embed_dim", 256),
                   pooled_dim=jc.get("projection_class_embeddings_input_dim", 2816) - 6 * jc.get("addition_time_embed_dim", 256),
                   norm_groups=jc.get("norm_num_groups", 32))
        m = cls(cfg)
        from safetensors.torch import load_file
        sd = {}
        for fn in sorted(os.listdir(folder)):
            if fn.endswith(".safetensors") and "fp16" not in fn or fn == "diffusion_pytorch_model.fp16.safetensors" and not sd:
                sd.update(load_file(os.path.join(folder, fn)))
        from . import ckpt as _ckpt
        _ckpt.load_checked(m, sd, "unet:")
        return m.to(torch_dtype) if torch_dtype is not None else m

    def init_synthetic(self, seed=0):
        _init_synthetic(self, seed)
        self._prep = None
        return self

    # -- weight preparation (once per dtype/device/version) -------------------------------------------------
    def _prepare(self):
        sig = _sig(self)
        if self._prep is not None and self._prep_sig == sig:
            return self._prep
        sd = {k: v.data for k, v in self.named_parameters()}
        P = {}
        for k, v in sd.items():
            if v.dim() == 4 and v.shape[-1] == 3:
                cpad = 8 if v.shape[1] < 8 else None
                P[k] = _conv_w(v, cpad)
            elif v.dim() == 4:
                P[k] = v.reshape(v.shape[0], v.shape[1]).contiguous()
            else:
                P[k] = v
        # fused projections: self-attention q|k|v, cross-attention k|v
        for k in list(sd.keys()):
            if k.endswith("attn1.to_q.weight"):
                b = k[:-len("to_q.weight")]
                P[b + "qkv"] = torch.cat([sd[b + "to_q.weight"], sd[b + "to_k.weight"], sd[b + "to_v.weight"]], 0).contiguous()
            if k.endswith("ff.net.0.proj.weight"):   # GEGLU fused into the GEMM epilogue: interleave (value_i, gate_i) rows
                b = k[:-len("weight")]
                w, bb = sd[k], sd[b + "bias"]
                d = w.shape[0] // 2
                P[b + "pairs.weight"] = torch.stack([w[:d], w[d:]], dim=1).reshape(2 * d, w.shape[1]).contiguous()
                P[b + "pairs.bias"] = torch.stack([bb[:d], bb[d:]], dim=1).reshape(2 * d).contiguous()
            if k.endswith("attn2.to_k.weight"):
                b = k[:-len("to_k.weight")]
                P[b + "kv"] = torch.cat([sd[b + "to_k.weight"], sd[b + "to_v.weight"]], 0).contiguous()
        # every ResBlock's time_emb_proj stacked into one [sum(Cout), temb] matrix: one GEMM per forward
        # instead of 17-23 latency-bound M=2 launches; each block reads its slice through rowvec_stride
        names = [k[:-len(".time_emb_proj.weight")] for k in sd if k.endswith(".time_emb_proj.weight")]
        if names:
            P["temb_all.weight"] = torch.cat([sd[n + ".time_emb_proj.weight"] for n in names], 0).contiguous()
            P["temb_all.bias"] = torch.cat([sd[n + ".time_emb_proj.bias"] for n in names], 0).contiguous()
            off = 0
            P["temb_all.offsets"] = {}
            for n in names:
                P["temb_all.offsets"][n] = off
                off += sd[n + ".time_emb_proj.weight"].shape[0]
        if getattr(self, "_lnfold", False) and sd[next(iter(sd))].dtype in (torch.bfloat16, torch.float16):
            self._prepare_lnfold(P)
        if getattr(self, "_fp8", False):
            if sd[next(iter(sd))].dtype != torch.bfloat16:
                raise _lib.SSError("the fp8 UNet path produces bf16 activations: move the module to bf16 first")
            self._prepare_fp8(P)
        self._prep, self._prep_sig = P, sig
        self._prep_gen = getattr(self, "_prep_gen", 0) + 1      # identity of the prepared weights (graph cache key)
        return P

    # -- forward ----------------------------------------------------------------------------------------------
    def _resnet(self, P, n, x, B, H, W, temb_act, groups, eps=1e-5):
        h = ops.groupnorm(x, P[n + ".norm1.weight"], P[n + ".norm1.bias"], B, groups, eps, silu=True)
        tp, tld = None, 0
        if temb_act is not None:                                        # temb_act = stacked projections [B, sum(Cout)]
            off = P["temb_all.offsets"][n]
            tld = temb_act.shape[1]
            tp = temb_act[:, off:]                                      # view: row stride tld, first Cout columns used
        h, _, _ = ops.conv3x3(h, P[n + ".conv1.weight"], B, H, W, bias=P[n + ".conv1.bias"], rowvec=tp, rowvec_stride=tld)
        h = ops.groupnorm(h, P[n + ".norm2.weight"], P[n + ".norm2.bias"], B, groups, eps, silu=True)
        sc = x
        if (n + ".conv_shortcut.weight") in P:
            sc = ops.gemm(x, P[n + ".conv_shortcut.weight"], bias=P[n + ".conv_shortcut.bias"])
        h, _, _ = ops.conv3x3(h, P[n + ".conv2.weight"], B, H, W, bias=P[n + ".conv2.bias"], residual=sc)
        return h

    # -- fp8 (OCP e4m3) linear layers of the transformer blocks (BASELINE configs[4]; SURVEY §8 ★ row) -----------
    def enable_fp8(self, on=True):
        """Run proj_in / q|k|v / to_out / to_q / ff1 (GEGLU) / ff2 / proj_out of every transformer block through the fp8
        MFMA GEMM (row-wise dynamic activation scales, per-output-channel weight scales, fp32 accumulate, bf16 out).
        Convolutions, attention and the 64-token context K/V projection stay bf16."""
        if bool(on) != getattr(self, "_fp8", False):
            self._fp8 = bool(on)
            self._prep = None          # weights are re-prepared (and a captured forward is invalidated via _prep_gen)
        return self

    def enable_lnfold(self, on=True):
        """Fold norm1 / norm2 / norm3 of every transformer block into the GEMM that follows (q|k|v, to_q, GEGLU ff1):
        the GEMM streams the raw rows and its epilogue applies the per-row statistics.  The statistics are per-strip
        (sum, sum of squares) partials written by the epilogue of the GEMM that PRODUCED the rows (`ss_gemm_rowpart`: no
        atomics) and folded in a fixed order by the consumer's epilogue (`ss_gemm_lnfold_part`: no finalize launch); where
        the producer's tile is not eligible, a statistics pass (`ss_rowstats` + `ss_gemm_lnfold`).  16-bit path only (fp8
        fuses the LayerNorm into its quantiser instead; fp32 runs the plain LayerNorm).  OFF by default: on MI355X the
        forward does not get faster either way (round 3, atomic form: 66.4 ms on and off, profiles/round3_lnfold_rowstat_ab.txt;
        round 4, strip form: still +-0) — the LayerNorm launches are traded for a slower folded epilogue on
        the q|k|v / ff1 GEMMs.  (The round-3 `ss_gemm_rowstat` + `ss_rowstat_finalize` atomic form is legacy API: kept in
        the C ABI with its test, no caller in the model.)"""
        if bool(on) != getattr(self, "_lnfold", False):
            self._lnfold = bool(on)
            self._prep = None
        return self

    def _prepare_lnfold(self, P):
        """Wg = gamma ∘ W (one rounding to the model dtype), c = row sums of Wg (fp32), d = b + W · beta."""
        for k in list(P.keys()):
            if not k.endswith(".norm1.weight") or ".transformer_blocks." not in k:
                continue
            b = k[:-len(".norm1.weight")]
            for norm, wkey, bkey in ((".norm1", ".attn1.qkv", None), (".norm2", ".attn2.to_q.weight", None),
                                     (".norm3", ".ff.net.0.proj.pairs.weight", ".ff.net.0.proj.pairs.bias")):
                W = P[b + wkey]
                if W.shape[1] % 64 or W.shape[0] % 16:
                    continue
                gamma, beta = P[b + norm + ".weight"].float(), P[b + norm + ".bias"].float()
                wg = (W.float() * gamma[None, :]).to(W.dtype).contiguous()
                c = wg.float().sum(dim=1).contiguous()
                d = W.float() @ beta
                if bkey is not None:
                    d = d + P[b + bkey].float()
                P[b + wkey + ".lnf"] = (wg, c, d.to(W.dtype).contiguous())

    def _prepare_fp8(self, P):
        for k in list(P.keys()):
            v = P[k]
            if not (isinstance(v, torch.Tensor) and v.dim() == 2 and ".attentions." in k):
                continue
            if k.endswith(("attn1.qkv", "pairs.weight", "to_out.0.weight", "attn2.to_q.weight", "ff.net.2.weight",
                           "proj_in.weight", "proj_out.weight")) and v.shape[1] % 128 == 0:
                P[k + ".fp8"] = ops.quantize_rows_fp8(v.contiguous())

    def _lin(self, P, name, x, *, ln=None, bias=None, residual=None, geglu=False, rowstat=None):
        """One linear layer of a transformer block: fp8 when enabled and prepared for this weight, else bf16.
        ``rowstat``: the forward's row-statistics carrier (see `_rs_new`): this GEMM's epilogue also writes the per-strip
        (sum, sum of squares) of its OUTPUT rows for the LayerNorm that follows (see `_lin_ln`)."""
        f8 = P.get(name + ".fp8") if getattr(self, "_fp8", False) else None
        if f8 is None:
            if ln is not None:
                x = ops.layernorm(x, ln[0], ln[1], ln[2])
            if geglu:
                return ops.gemm_geglu(x, P[name], bias)
            part = None
            if rowstat is not None:
                M, K = x.shape
                N = P[name].shape[0]
                strips = ops.rowpart_strips(M, N, K, x.dtype)
                part = rowstat["buf"][:M * strips * 2].view(M, strips, 2)
                rowstat["cur"] = part
            return ops.gemm(x, P[name], bias=bias, residual=residual, rowpart=part)
        assert rowstat is None
        x8, sx = ops.quantize_rows_fp8(x, ln=ln)
        return ops.gemm_fp8(x8, sx, f8[0], f8[1], bias=bias, residual=residual, geglu=geglu)

    def _lnfold_on(self):
        return getattr(self, "_lnfold", False) and not getattr(self, "_fp8", False)

    def _lin_ln(self, P, name, x, ln, bias=None, geglu=False, rowstat=None):
        """LayerNorm + linear: folded (ss_gemm_lnfold) when enabled and prepared for this weight.  The row statistics come
        from the per-strip partials the GEMM that produced x wrote in its epilogue (``rowstat``; ss_gemm_rowpart ->
        ss_gemm_lnfold_part: no pass over x, no finalize launch, no atomics), else from a statistics pass over x
        (ss_rowstats)."""
        lnf = P.get(name + ".lnf") if self._lnfold_on() else None
        if lnf is None:
            assert rowstat is None
            return self._lin(P, name, x, ln=ln, bias=bias, geglu=geglu)
        if rowstat is not None:
            return ops.gemm_lnfold_part(x, lnf[0], rowstat["cur"], x.shape[1], ln[2], lnf[1], bias_d=lnf[2], geglu=geglu)
        rstd, shift = ops.rowstats(x, ln[2])
        return ops.gemm_lnfold(x, lnf[0], rstd, shift, lnf[1], bias_d=lnf[2], geglu=geglu)

    def _rs_new(self, rows, width, device):
        """Carrier of producer-side row statistics for ``rows`` tokens of ``width`` channels: one fp32 buffer of per-strip
        (sum, sum of squares) pairs, fully rewritten by every producer (nothing to zero), sized for the narrowest strip (64)."""
        bufs = self.__dict__.setdefault("_rs_bufs", {})
        key = (rows, width, str(device))
        b = bufs.get(key)
        if b is None:
            b = bufs[key] = torch.empty(rows * ((width + 63) // 64) * 2, dtype=torch.float32, device=device)
            self._kv_gen = getattr(self, "_kv_gen", 0) + 1      # a captured forward does not know this buffer
        return {"buf": b, "cur": None}

    def _transformer(self, P, n, x, B, HW, ctx2d, Lctx, heads, layers, groups):
        h = ops.groupnorm(x, P[n + ".norm.weight"], P[n + ".norm.bias"], B, groups, 1e-6, silu=False)
        # producer-carried LayerNorm statistics: every GEMM that writes h (proj_in, attn1/attn2 to_out, ff.net.2)
        # accumulates the row sums the NEXT norm needs while it stores h; requires the folded form of all three norms
        rs = None
        if self._lnfold_on() and h.shape[0] > 128 and all(
                (n + ".transformer_blocks.%d%s.lnf" % (k, w)) in P for k in range(layers)
                for w in (".attn1.qkv", ".attn2.to_q.weight", ".ff.net.0.proj.pairs.weight")):
            width = P[n + ".proj_in.weight"].shape[0]
            if all(ops.rowpart_strips(h.shape[0], width, kk, h.dtype) > 0 for kk in (h.shape[1], width, 4 * width)):
                rs = self._rs_new(h.shape[0], width, h.device)
        h = self._lin(P, n + ".proj_in.weight", h, bias=P[n + ".proj_in.bias"], rowstat=rs)
        for k in range(layers):
            b = n + ".transformer_blocks.%d" % k
            qkv = self._lin_ln(P, b + ".attn1.qkv", h, (P[b + ".norm1.weight"], P[b + ".norm1.bias"], 1e-5), rowstat=rs)
            a = ops.attention_qkv_packed(qkv, B, HW, heads)
            h = self._lin(P, b + ".attn1.to_out.0.weight", a, bias=P[b + ".attn1.to_out.0.bias"], residual=h, rowstat=rs)
            q = self._lin_ln(P, b + ".attn2.to_q.weight", h, (P[b + ".norm2.weight"], P[b + ".norm2.bias"], 1e-5), rowstat=rs)
            kv = self._ctx_kv.get(b)
            if kv is None:   # K/V of the 64 context tokens do not depend on the denoising step: once per image,
                # written into a per-block buffer that keeps its address (a captured forward reads it on replay)
                w = P[b + ".attn2.kv"]
                buf = self._ctx_kv_buf.get(b) if hasattr(self, "_ctx_kv_buf") else None
                if buf is None or buf.shape != (ctx2d.shape[0], w.shape[0]) or buf.dtype != ctx2d.dtype:
                    if not hasattr(self, "_ctx_kv_buf"):
                        self._ctx_kv_buf = {}
                    buf = self._ctx_kv_buf[b] = torch.empty(ctx2d.shape[0], w.shape[0], dtype=ctx2d.dtype, device=ctx2d.device)
                    self._kv_gen = getattr(self, "_kv_gen", 0) + 1      # a captured forward holding the old address is stale
                kv = self._ctx_kv[b] = ops.gemm(ctx2d, w, out=buf)
            a = ops.attention_q_kvpacked(q, kv, B, HW, Lctx, heads)
            h = self._lin(P, b + ".attn2.to_out.0.weight", a, bias=P[b + ".attn2.to_out.0.bias"], residual=h, rowstat=rs)
            u = self._lin_ln(P, b + ".ff.net.0.proj.pairs.weight", h, (P[b + ".norm3.weight"], P[b + ".norm3.bias"], 1e-5),
                             bias=P[b + ".ff.net.0.proj.pairs.bias"], geglu=True, rowstat=rs)
            h = self._lin(P, b + ".ff.net.2.weight", u, bias=P[b + ".ff.net.2.bias"], residual=h,
                          rowstat=rs if k + 1 < layers else None)      # the last block's output feeds proj_out, not a norm
        return self._lin(P, n + ".proj_out.weight", h, bias=P[n + ".proj_out.bias"], residual=x)

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None, return_dict=True, **kw):
        """sample [B,4,h,w] NCHW (diffusers convention) -> eps [B,4,h,w]."""
        c = self.cfg
        P = self._prepare()
        dt, dev = sample.dtype, sample.device
        B, _, H, W = sample.shape
        boc, G, L = c["block_out_channels"], c["norm_groups"], c["layers_per_block"]
        # time + added ("text_time") conditioning: sinusoids on the host, MLPs on the device
        temb = kw.get("temb_in")          # [B, 320] device tensor (graph replay: no host->device copy inside the capture)
        if temb is None:
            t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(B)
            temb = timestep_embedding(t, boc[0]).to(device=dev, dtype=dt)
        emb = ops.gemm(ops.silu(ops.gemm(temb, P["time_embedding.linear_1.weight"], bias=P["time_embedding.linear_1.bias"])),
                       P["time_embedding.linear_2.weight"], bias=P["time_embedding.linear_2.bias"])
        tids = added_cond_kwargs["time_ids"]
        tkey = (tuple(tids.flatten().tolist()), dt, str(dev)) if tids.device.type == "cpu" else None
        tid = self._tid_cache.get(tkey) if tkey is not None and hasattr(self, "_tid_cache") else None
        if tid is None:                   # constant of (height, width): embedded on the host once, kept on the device
            tid = timestep_embedding(tids.flatten().cpu(), c["addition_time_embed_dim"]).reshape(B, -1).to(device=dev, dtype=dt)
            if tkey is not None:
                self._tid_cache = {tkey: tid}
        add = torch.cat([added_cond_kwargs["text_embeds"].to(dt), tid], dim=-1).contiguous()
        emb = ops.gemm(ops.silu(ops.gemm(add, P["add_embedding.linear_1.weight"], bias=P["add_embedding.linear_1.bias"])),
                       P["add_embedding.linear_2.weight"], bias=P["add_embedding.linear_2.bias"], residual=emb)
        temb_act = ops.silu(emb)                                            # F.silu(temb) feeds every ResBlock
        temb_act = ops.gemm(temb_act, P["temb_all.weight"], bias=P["temb_all.bias"])   # all time_emb_proj at once
        ctx = encoder_hidden_states.to(dt).contiguous()
        Lctx = ctx.shape[1]
        ctx2d = ctx.view(B * Lctx, -1)
        ckey = (encoder_hidden_states.data_ptr(), encoder_hidden_states._version, tuple(ctx.shape), dt, id(P))
        if getattr(self, "_ctx_key", None) != ckey:       # new conditioning (new image): drop the cached cross-attn K/V
            self._ctx_key, self._ctx_kv = ckey, {}

        x = ops.nchw_to_nhwc(sample.contiguous(), 8)                       # 4 -> 8 zero-padded channels
        h, _, _ = ops.conv3x3(x, P["conv_in.weight"], B, H, W, bias=P["conv_in.bias"])
        skips = [(h, H, W)]
        for i in range(len(boc)):
            for j in range(L):
                h = self._resnet(P, "down_blocks.%d.resnets.%d" % (i, j), h, B, H, W, temb_act, G)
                if c["transformer_layers"][i]:
                    h = self._transformer(P, "down_blocks.%d.attentions.%d" % (i, j), h, B, H * W, ctx2d, Lctx,
                                          c["num_heads"][i], c["transformer_layers"][i], G)
                skips.append((h, H, W))
            if i < len(boc) - 1:
                n = "down_blocks.%d.downsamplers.0.conv" % i
                h, H, W = ops.conv3x3(h, P[n + ".weight"], B, H, W, stride=2, bias=P[n + ".bias"])
                skips.append((h, H, W))
        h = self._resnet(P, "mid_block.resnets.0", h, B, H, W, temb_act, G)
        h = self._transformer(P, "mid_block.attentions.0", h, B, H * W, ctx2d, Lctx, c["num_heads"][-1],
                              c["transformer_layers"][-1], G)
        h = self._resnet(P, "mid_block.resnets.1", h, B, H, W, temb_act, G)
        for i in range(len(boc)):
            ri = len(boc) - 1 - i
            for j in range(L + 1):
                sk, _, _ = skips.pop()
                h = ops.concat_channels(h, sk)                               # torch.cat([h, skip], dim=1)
                h = self._resnet(P, "up_blocks.%d.resnets.%d" % (i, j), h, B, H, W, temb_act, G)
                if c["transformer_layers"][ri]:
                    h = self._transformer(P, "up_blocks.%d.attentions.%d" % (i, j), h, B, H * W, ctx2d, Lctx,
                                          c["num_heads"][ri], c["transformer_layers"][ri], G)
            if i < len(boc) - 1:
                n = "up_blocks.%d.upsamplers.0.conv" % i
                h, H, W = ops.conv3x3(h, P[n + ".weight"], B, H, W, upsample=True, bias=P[n + ".bias"])
        h = ops.groupnorm(h, P["conv_norm_out.weight"], P["conv_norm_out.bias"], B, G, 1e-5, silu=True)
        o, _, _ = ops.conv3x3(h, P["conv_out.weight"], B, H, W, bias=P["conv_out.bias"])
        out = ops.nhwc_to_nchw(o, B, c["out_channels"], H, W)
        return _Sample(out) if return_dict else (out,)


# ---- VAE decoder ---------------------------------------------------------------------------------------------

class _VaeConfig:
    def __init__(self, scaling_factor, force_upcast=False):
        self.scaling_factor = scaling_factor
        self.force_upcast = force_upcast


class AutoencoderKL(nn.Module):
    """Decoder half only (the story path never encodes).

    Decode precision (diffusers: the SDXL VAE config sets ``force_upcast`` and the pipeline decodes an fp16 VAE in
    fp32, because its activations overflow fp16 at 1024 px — gen_george.py:62 + StableDiffusionXLPipeline):
      * bf16 module -> bf16 decode (fp32's exponent range: no overflow; tests/test_fulldim_gpu.py::
        test_vae_decode_full_size_pixel_parity measures the uint8 deviation from the fp32 reference);
      * fp16 module with ``force_upcast`` (the reference scripts' default dtype, gen_george.py:19) -> decoded in
        **fp32**, which is what diffusers does (``needs_upcasting = vae.dtype == float16 and config.force_upcast``);
        ``set_tuning("vae_bf16", 1)`` opts into the 7x faster bf16 decode for that case (fp32's exponent range, so no
        overflow, but narrower arithmetic than the reference: uint8 mean deviation 0.57, max 5-6) — never fp16;
      * ``set_tuning("vae_fp32", 1)`` -> decoded in fp32 whatever the module dtype (exact-fp32 MFMA path)."""

    def __init__(self, config=None):
        super().__init__()
        self.cfg = dict(SDXL_BASE_VAE if config is None else config)
        _grow(self, _vae_shapes(self.cfg))
        self.config = _VaeConfig(self.cfg["scaling_factor"], bool(self.cfg.get("force_upcast", True)))
        self._prep = None
        self._prep_sig = None

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **kw):
        folder = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(folder, "config.json")) as f:
            jc = json.load(f)
        cfg = dict(latent_channels=jc.get("latent_channels", 4), out_channels=jc.get("out_channels", 3),
                   block_out_channels=tuple(jc["block_out_channels"]), layers_per_block=jc.get("layers_per_block", 2),
                   norm_groups=jc.get("norm_num_groups", 32), scaling_factor=jc.get("scaling_factor", 0.13025),
                   force_upcast=jc.get("force_upcast", True))
        m = cls(cfg)
        from safetensors.torch import load_file
        sd = {}
        for fn in sorted(os.listdir(folder)):
            if fn.endswith(".safetensors") and not sd:
                sd.update(load_file(os.path.join(folder, fn)))
        # older checkpoints name the mid-block attention projections query/key/value/proj_attn
        ren = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}
        sd = {_rename(k, ren): v for k, v in sd.items()}
        from . import ckpt as _ckpt
        _ckpt.load_checked(m, sd, "vae:", expect_unexpected=("encoder.", "quant_conv."))      # decode-only module
        return m.to(torch_dtype) if torch_dtype is not None else m

    def init_synthetic(self, seed=0):
        _init_synthetic(self, seed)
        self._prep = None
        return self

    def decode_dtype(self):
        """Arithmetic dtype of the decoder for this module's parameter dtype (see the class docstring)."""
        p0 = next(self.parameters()).dtype
        if _lib.get_tuning("vae_fp32", 0):
            return torch.float32
        if p0 == torch.float16 and self.config.force_upcast:
            return torch.bfloat16 if _lib.get_tuning("vae_bf16", 0) else torch.float32
        return p0

    def _prepare(self, run_dtype=None):
        run_dtype = self.decode_dtype() if run_dtype is None else run_dtype
        sig = _sig(self) + (run_dtype,)
        if self._prep is not None and self._prep_sig == sig:
            return self._prep
        P = {}
        for k, v in self.named_parameters():
            v = v.data.to(run_dtype)
            if v.dim() == 4 and v.shape[-1] == 3:
                P[k] = _conv_w(v, 8 if v.shape[1] < 8 else None)
            elif v.dim() == 4:
                P[k] = v.reshape(v.shape[0], v.shape[1]).contiguous()
            else:
                P[k] = v
        # post_quant_conv (1x1, 4 -> 4) consumes the 8-channel padded latent layout
        w = P["post_quant_conv.weight"]
        wp = torch.zeros(8, 8, dtype=w.dtype, device=w.device)
        wp[:w.shape[0], :w.shape[1]] = w
        bp = torch.zeros(8, dtype=w.dtype, device=w.device)
        bp[:w.shape[0]] = P["post_quant_conv.bias"]
        P["post_quant_conv.weight"], P["post_quant_conv.bias"] = wp, bp
        self._prep, self._prep_sig = P, sig
        return P

    def _resnet(self, P, n, x, B, H, W, G):
        h = ops.groupnorm(x, P[n + ".norm1.weight"], P[n + ".norm1.bias"], B, G, 1e-6, silu=True)
        h, _, _ = ops.conv3x3(h, P[n + ".conv1.weight"], B, H, W, bias=P[n + ".conv1.bias"])
        h = ops.groupnorm(h, P[n + ".norm2.weight"], P[n + ".norm2.bias"], B, G, 1e-6, silu=True)
        sc = x
        if (n + ".conv_shortcut.weight") in P:
            sc = ops.gemm(x, P[n + ".conv_shortcut.weight"], bias=P[n + ".conv_shortcut.bias"])
        h, _, _ = ops.conv3x3(h, P[n + ".conv2.weight"], B, H, W, bias=P[n + ".conv2.bias"], residual=sc)
        return h

    @torch.no_grad()
    def decode_nhwc(self, latents_scaled, prescale=1.0):
        """latents [B,4,h,w] (times `prescale`, i.e. pass 1/scaling_factor for raw latents) -> NHWC image
        tensor [B*H*W, 8] in [-1,1] (channels 3..7 are zero padding)."""
        c = self.cfg
        run_dtype = self.decode_dtype()
        P = self._prepare(run_dtype)
        G, boc = c["norm_groups"], c["block_out_channels"]
        B, _, H, W = latents_scaled.shape
        z = ops.nchw_to_nhwc(latents_scaled.to(run_dtype).contiguous(), 8)
        pq = P["post_quant_conv.weight"]
        if prescale != 1.0:
            key = "post_quant_conv.weight@%r" % prescale
            if key not in P:
                P[key] = (pq.float() * prescale).to(pq.dtype)
            pq = P[key]
        z = ops.gemm(z, pq, bias=P["post_quant_conv.bias"])
        h, _, _ = ops.conv3x3(z, P["decoder.conv_in.weight"], B, H, W, bias=P["decoder.conv_in.bias"])
        h = self._resnet(P, "decoder.mid_block.resnets.0", h, B, H, W, G)
        a = "decoder.mid_block.attentions.0"
        C = h.shape[1]
        y = ops.groupnorm(h, P[a + ".group_norm.weight"], P[a + ".group_norm.bias"], B, G, 1e-6, silu=False)
        q = ops.gemm(y, P[a + ".to_q.weight"], bias=P[a + ".to_q.bias"])
        k = ops.gemm(y, P[a + ".to_k.weight"], bias=P[a + ".to_k.bias"])
        v = ops.gemm(y, P[a + ".to_v.weight"], bias=P[a + ".to_v.bias"])
        outs = []
        T = H * W
        for b in range(B):   # single head of dim C (512): materialised scores like the reference's eager path
            qb, kb, vb = q[b * T:(b + 1) * T], k[b * T:(b + 1) * T], v[b * T:(b + 1) * T]
            if C <= 128:
                outs.append(ops.attention(qb.unsqueeze(0).contiguous(), kb.unsqueeze(0).contiguous(),
                                          vb.unsqueeze(0).contiguous(), 1)[0])
            else:
                s = ops.gemm(qb.contiguous(), kb.contiguous())               # [T, T]
                ops.softmax_rows_(s, 1.0 / math.sqrt(C))
                outs.append(ops.gemm(s, ops.transpose(vb.contiguous())))     # P @ V
        o = torch.cat(outs, dim=0) if B > 1 else outs[0]
        h = ops.gemm(o.contiguous(), P[a + ".to_out.0.weight"], bias=P[a + ".to_out.0.bias"], residual=h)
        h = self._resnet(P, "decoder.mid_block.resnets.1", h, B, H, W, G)
        for i in range(len(boc)):
            for j in range(c["layers_per_block"] + 1):
                h = self._resnet(P, "decoder.up_blocks.%d.resnets.%d" % (i, j), h, B, H, W, G)
            if i < len(boc) - 1:
                n = "decoder.up_blocks.%d.upsamplers.0.conv" % i
                h, H, W = ops.conv3x3(h, P[n + ".weight"], B, H, W, upsample=True, bias=P[n + ".bias"])
        h = ops.groupnorm(h, P["decoder.conv_norm_out.weight"], P["decoder.conv_norm_out.bias"], B, G, 1e-6, silu=True)
        # conv_out to 3 channels, written into an 8-channel padded NHWC tensor
        w = P["decoder.conv_out.weight"]
        wp = torch.zeros(8, w.shape[1], dtype=w.dtype, device=w.device)
        wp[:w.shape[0]] = w
        bp = torch.zeros(8, dtype=w.dtype, device=w.device)
        bp[:w.shape[0]] = P["decoder.conv_out.bias"]
        img, _, _ = ops.conv3x3(h, wp, B, H, W, bias=bp)
        return img, H, W

    @torch.no_grad()
    def decode(self, latents, return_dict=True):
        """diffusers signature: latents are ALREADY divided by scaling_factor by the caller."""
        img, H, W = self.decode_nhwc(latents)
        B = latents.shape[0]
        out = ops.nhwc_to_nchw(img, B, self.cfg["out_channels"], H, W).to(latents.dtype)
        return _Sample(out) if return_dict else (out,)


def _rename(k, table):
    for a, b in table.items():
        k = k.replace(a, b)
    return k


# ---- scheduler + pipeline ----------------------------------------------------------------------------------------

class EulerDiscreteScheduler:
    """EulerDiscreteScheduler of SDXL-base: scaled-linear betas, 'leading' timestep spacing with
    steps_offset, linearly interpolated sigmas, epsilon prediction, no churn."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 timestep_spacing="leading", steps_offset=1, prediction_type="epsilon", **kw):
        assert beta_schedule == "scaled_linear" and prediction_type == "epsilon"
        self.num_train_timesteps, self.steps_offset, self.timestep_spacing = num_train_timesteps, steps_offset, timestep_spacing
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float32) ** 2
        ac = np.cumprod(1.0 - betas)
        self._sigmas_all = ((1 - ac) / ac) ** 0.5
        self.timesteps = None
        self.sigmas = None
        self.init_noise_sigma = None

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        folder = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(folder, "scheduler_config.json")) as f:
            jc = json.load(f)
        return cls(**{k: v for k, v in jc.items() if not k.startswith("_")})

    def set_timesteps(self, n):
        if self.timestep_spacing == "leading":
            ratio = self.num_train_timesteps // n
            ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.float32) + self.steps_offset
        elif self.timestep_spacing == "linspace":
            ts = np.linspace(0, self.num_train_timesteps - 1, n, dtype=np.float32)[::-1].copy()
        else:
            ratio = self.num_train_timesteps / n
            ts = (np.arange(self.num_train_timesteps, 0, -ratio)).round().astype(np.float32) - 1
        s = np.interp(ts, np.arange(0, len(self._sigmas_all)), self._sigmas_all)
        self.sigmas = np.concatenate([s, [0.0]]).astype(np.float32)
        self.timesteps = ts
        mx = float(self.sigmas.max())
        self.init_noise_sigma = mx if self.timestep_spacing in ("linspace", "trailing") else float((mx ** 2 + 1) ** 0.5)


class _PipeOut:
    def __init__(self, images):
        self.images = images


class StableDiffusionXLPipeline:
    """The slice of diffusers' StableDiffusionXLPipeline that SDXLAdapter.generate uses (prompt embeds
    supplied, no text encoders; adapter_modules.py:369-375,455-466)."""

    def __init__(self, vae, unet, scheduler, tokenizer=None, tokenizer_2=None, text_encoder=None, text_encoder_2=None,
                 **kw):
        self.vae, self.unet, self.scheduler = vae, unet, scheduler

    def _stable(self, name, value):
        """Conditioning lives in buffers that keep their address across calls (same shape/dtype): the captured UNet
        forward reads them on replay; a new image is an in-place copy (which also bumps the tensor version the
        UNet keys its cross-attention K/V cache on)."""
        bufs = self.__dict__.setdefault("_bufs", {})
        cur = bufs.get(name)
        if cur is None or cur.shape != value.shape or cur.dtype != value.dtype or cur.device != value.device:
            cur = bufs[name] = value.contiguous().clone()
            # captured forwards read the OLD buffer: they can never be replayed again (and the caching allocator may hand
            # its address to an unrelated tensor) -> drop them, their private pools with them
            self._buf_gen = getattr(self, "_buf_gen", 0) + 1
            self.__dict__.get("_graphs", {}).clear()
        else:
            cur.copy_(value)
        return cur

    def _unet_graph(self, xin, ctx, cond, ts, n_steps):
        """hipGraph of ONE UNet forward (all ~1700 launches), captured once per (shape, conditioning buffers) and
        replayed for steps 1..n-1 of every render: removes the per-launch gaps (~5 % of a forward at batch 8, ~10 %
        at batch 2).  Returns None (eager fallback) if capture is not possible."""
        graphs = self.__dict__.setdefault("_graphs", {})
        # identity = shapes + GENERATION counters of every buffer the captured launches read (conditioning buffers,
        # per-block cross-attention K/V buffers, prepared weights) — never raw addresses or id(): those are reused
        key = (tuple(xin.shape), xin.dtype, tuple(ctx.shape), tuple(cond["time_ids"].flatten().tolist()),
               getattr(self, "_buf_gen", 0), getattr(self.unet, "_kv_gen", 0), getattr(self.unet, "_prep_gen", 0),
               _lib.tuning_generation())     # a knob set after the capture (arithmetic mode, kernel version) invalidates it
        for k_old in [k for k in graphs if k[4:] != key[4:]]:      # entries of an older generation are dead
            del graphs[k_old]
        ent = graphs.get(key)
        table = timestep_embedding(torch.as_tensor(ts, dtype=torch.float32), self.unet.cfg["block_out_channels"][0])
        table = table.to(device=xin.device, dtype=xin.dtype)[:, None, :].expand(-1, xin.shape[0], -1).contiguous()
        if ent is not None:
            ent["temb_table"] = table
            return ent
        try:
            sx = torch.empty_like(xin)
            st = torch.empty(xin.shape[0], table.shape[-1], dtype=xin.dtype, device=xin.device)
            sx.copy_(xin)
            st.copy_(table[0])
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                eps = self.unet(sx, None, ctx, added_cond_kwargs=cond, return_dict=False, temb_in=st)[0]
            ent = graphs[key] = {"g": g, "x": sx, "temb": st, "eps": eps, "temb_table": table}
            return ent
        except Exception as ex:      # keep rendering eagerly
            import sys
            print("seedstory: UNet graph capture unavailable (%r); running eagerly" % (ex,), file=sys.stderr)
            graphs[key] = None
            torch.cuda.synchronize()
            return None

    @torch.no_grad()
    def __call__(self, prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds,
                 guidance_scale=7.5, num_inference_steps=30, generator=None, height=1024, width=1024, latents=None,
                 output_type="pil", **kw):
        dev, dt = prompt_embeds.device, prompt_embeds.dtype
        lh, lw = height // 8, width // 8
        B = prompt_embeds.shape[0]           # images rendered together (UNet batch 2B: [B uncond; B cond])
        self.unet._ctx_key = None            # new conditioning: recompute the cached cross-attention K/V
        self.scheduler.set_timesteps(num_inference_steps)
        sig, ts = self.scheduler.sigmas, self.scheduler.timesteps
        if latents is None:
            latents = torch.randn((B, 4, lh, lw), generator=generator, device=dev, dtype=dt)
        assert latents.shape[0] == B
        x = (latents.to(device=dev, dtype=dt) * self.scheduler.init_noise_sigma).contiguous()
        time_ids = torch.tensor([[height, width, 0, 0, height, width]] * (2 * B), dtype=torch.float32)
        ctx = self._stable("ctx", torch.cat([negative_prompt_embeds, prompt_embeds], dim=0))
        pooled = self._stable("pooled", torch.cat([negative_pooled_prompt_embeds, pooled_prompt_embeds], dim=0))
        cond = {"text_embeds": pooled, "time_ids": time_ids}
        use_graph = _lib.get_tuning("unet_graph", 1) != 0 and num_inference_steps > 2
        graph = None
        for i in range(num_inference_steps):
            xin = ops.euler_scale_dup(x, float(sig[i])).view(2 * B, 4, lh, lw)   # [uncond; cond] batch, x/sqrt(s^2+1)
            if i == 0 or not use_graph:
                # step 0 always runs eagerly: it refreshes the cross-attention K/V of this conditioning (and lets
                # the GEMM autotuner see any new shape) before the captured forward replays
                eps = self.unet(xin, float(ts[i]), ctx, added_cond_kwargs=cond, return_dict=False)[0]
            else:
                if graph is None:
                    graph = self._unet_graph(xin, ctx, cond, ts, num_inference_steps)
                    if graph is None:
                        use_graph = False
                        eps = self.unet(xin, float(ts[i]), ctx, added_cond_kwargs=cond, return_dict=False)[0]
                if graph is not None:
                    graph["x"].copy_(xin)
                    graph["temb"].copy_(graph["temb_table"][i])
                    graph["g"].replay()
                    eps = graph["eps"]
            ops.euler_cfg_step_(x, eps.contiguous(), guidance_scale, float(sig[i]), float(sig[i + 1]))
        if output_type == "latent":
            return _PipeOut(x)
        # latents / scaling_factor is folded into the 1x1 post_quant_conv weights (linear, exact in fp32)
        imgs = []
        for b in range(B):                   # one image at a time: the 1024^2 decoder activations are 0.5 GB each
            img, H, W = self.vae.decode_nhwc(x[b:b + 1].contiguous(), prescale=1.0 / self.vae.config.scaling_factor)
            imgs.append(ops.image_to_u8(img, H * W).view(H, W, 3))
        if output_type == "pt":
            return _PipeOut(imgs[0] if B == 1 else torch.stack(imgs))
        from PIL import Image
        return _PipeOut([Image.fromarray(u8.cpu().numpy()) for u8 in imgs])
