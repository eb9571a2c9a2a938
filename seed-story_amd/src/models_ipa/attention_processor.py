"""diffusers attention-processor plug point, landing on the HIP library.

The reference's UNet attention boundary is the diffusers processor protocol
(``/root/reference/src/models_ipa/attention_processor.py:7-79`` ``AttnProcessor``, ``:189-280`` ``AttnProcessor2_0``;
installed with ``unet.set_attn_processor({...})`` at ``adapter_modules.py:38-62``):

    proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None) -> hidden_states

where ``attn`` is a diffusers ``Attention`` module (``to_q / to_k / to_v`` Linear, ``to_out = [Linear, Dropout]``, ``heads``,
``group_norm``, ``spatial_norm``, ``norm_cross``, ``residual_connection``, ``rescale_output_factor``).

``seedstory.diffusion.UNet2DConditionModel`` does not go through this protocol (its transformer blocks run a fused q|k|v
projection straight into the flash kernel); this class is for a user who KEEPS the real diffusers modules and wants their
attention arithmetic on the MI355X kernels: every product of the reference processor — the three projections, softmax(q k^T
/ sqrt(d)) v, the output projection — is one call through the C ABI (``ss_gemm`` / ``ss_attention``); the module-level glue
(spatial / group norm, 4-D reshape, residual, rescale) follows the reference line by line.

    from src.models_ipa.attention_processor import AttnProcessor
    unet.set_attn_processor({name: AttnProcessor() for name in unet.attn_processors})

Not supported (raises, never falls back to torch): an ``attention_mask`` (the SDXL path passes none; ``ss_attention`` has the
bottom-right causal mask only) and the IP-Adapter variants (``IPAttnProcessor*``: only ``IPAdapterSD`` installs them,
outside SURVEY §8).
"""
import torch
from torch import nn

from seedstory import ops
from seedstory._lib import SSError


class AttnProcessor(nn.Module):
    """Drop-in for the reference's ``AttnProcessor`` / ``AttnProcessor2_0`` (same constructor, same call protocol)."""

    def __init__(self, hidden_size=None, cross_attention_dim=None):
        super().__init__()

    @staticmethod
    def _linear(lin, x2d, residual=None):
        w = lin.weight
        b = getattr(lin, "bias", None)
        return ops.gemm(x2d, w if w.is_contiguous() else w.contiguous(), bias=b, residual=residual)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, *args, **kwargs):
        if attention_mask is not None:
            raise SSError("AttnProcessor: attention_mask is not supported by ss_attention (the SDXL path passes none)")
        if not hidden_states.is_cuda:
            raise SSError("AttnProcessor: tensors must live on the GPU (there is no CPU fallback)")
        residual = hidden_states                                             # attention_processor.py:27
        if getattr(attn, "spatial_norm", None) is not None:                  # :29-30
            hidden_states = attn.spatial_norm(hidden_states, temb)
        input_ndim = hidden_states.ndim
        if input_ndim == 4:                                                  # :34-36
            batch_size, channel, height, width = hidden_states.shape
            hidden_states = hidden_states.view(batch_size, channel, height * width).transpose(1, 2)
        batch_size, q_len, width_q = hidden_states.shape
        if getattr(attn, "group_norm", None) is not None:                    # :45-46
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        hidden_states = hidden_states.contiguous()
        x2d = hidden_states.reshape(batch_size * q_len, width_q)
        query = self._linear(attn.to_q, x2d)                                 # :48
        if encoder_hidden_states is None:                                    # :50-53
            enc2d, kv_len = x2d, q_len
        else:
            if getattr(attn, "norm_cross", False):
                encoder_hidden_states = attn.norm_encoder_hidden_states(encoder_hidden_states)
            encoder_hidden_states = encoder_hidden_states.to(hidden_states.dtype).contiguous()
            kv_len = encoder_hidden_states.shape[1]
            enc2d = encoder_hidden_states.reshape(batch_size * kv_len, encoder_hidden_states.shape[2])
        key = self._linear(attn.to_k, enc2d)                                 # :55-56
        value = self._linear(attn.to_v, enc2d)
        inner = query.shape[1]
        heads = int(attn.heads)
        # head_to_batch_dim -> softmax(q k^T * scale) v -> batch_to_head_dim (:58-64; 2_0: :246-258) in one launch: heads stay
        # packed along the feature axis, which is exactly ss_attention's [B, L, heads * d] layout
        scale = getattr(attn, "scale", None)
        out = ops.attention(query.view(batch_size, q_len, inner), key.view(batch_size, kv_len, inner),
                            value.view(batch_size, kv_len, inner), heads, scale=None if scale is None else float(scale))
        lin_out = attn.to_out[0]
        fuse_res = bool(getattr(attn, "residual_connection", False)) and input_ndim != 4 and \
            float(getattr(attn, "rescale_output_factor", 1.0)) == 1.0 and residual.shape[-1] == lin_out.weight.shape[0]
        res2d = residual.contiguous().reshape(batch_size * q_len, -1) if fuse_res else None
        hidden_states = self._linear(lin_out, out.view(batch_size * q_len, inner), residual=res2d)     # :67 (+ :76-77 fused)
        hidden_states = attn.to_out[1](hidden_states)                        # dropout (:69; identity at inference)
        hidden_states = hidden_states.view(batch_size, q_len, -1)
        if input_ndim == 4:                                                  # :71-72
            hidden_states = hidden_states.transpose(-1, -2).reshape(batch_size, channel, height, width)
        if getattr(attn, "residual_connection", False) and not fuse_res:     # :74-75
            hidden_states = hidden_states + residual
        rof = float(getattr(attn, "rescale_output_factor", 1.0))
        if rof != 1.0:                                                       # :77
            hidden_states = hidden_states / rof
        return hidden_states


# the reference selects the 2_0 classes when torch has scaled_dot_product_attention (adapter_modules.py:15-18): same arithmetic
AttnProcessor2_0 = AttnProcessor


def install(unet):
    """``unet.set_attn_processor`` with one ``AttnProcessor`` per attention layer of a diffusers UNet."""
    unet.set_attn_processor({name: AttnProcessor() for name in unet.attn_processors})
    return unet
