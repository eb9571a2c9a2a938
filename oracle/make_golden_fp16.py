"""fp16 rows from the REAL reference modules (the dtype the reference scripts actually run: gen_george.py:19-20,57,62-67).

    python oracle/make_golden_fp16.py            (build container only: reads /root/reference)

Same seeded tiny configurations as ``oracle/make_golden.py`` (its functions are reused with ``dtype=torch.float16``), run on CPU
behind ``oracle/ref_shims.py``: LLaMA forward (prefill / continuation / decode), ``ContinuousLVLM.generate`` semantics, both
Resamplers, the ViT with attention pool, ResamplerXLV2 and one ViT-G-width block.  The restatement (``oracle/seedstory_oracle.py``)
is asserted against every fp16 row on the way (1e-2: one-ulp flips from matmul blocking).  Written to a SEPARATE fixture,
``tests/golden/hotpath_tiny_fp16.safetensors``, so that ``hotpath_tiny.safetensors`` keeps reproducing bit-identically.
``tests/test_fp16_gpu.py`` compares the HIP path in fp16 with these rows and gates its distance to the fp32 rows by 1.5 x the
reference's own fp16-vs-fp32 distance.
"""
import json
import os
import sys
from functools import partial

import torch
from safetensors.torch import load_file, save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
import ref_shims  # noqa: E402
import seedstory_oracle as O  # noqa: E402
import synth  # noqa: E402

H = torch.float16


def main():
    torch.set_num_threads(8)
    llama_mod, qwen_mod, gen_mod, ipa_mod = ref_shims.import_reference()
    g32 = load_file(os.path.join(MG.GOLD, "hotpath_tiny.safetensors"))
    out = {}
    print("llama fp16"); MG.golden_llama(llama_mod, H, "llama_f16", out)
    print("generate fp16"); MG.golden_generate(llama_mod, gen_mod, qwen_mod, out, H, "gen_f16")
    # resamplers / ViT / ResamplerXLV2: the fp32 fixture's modules and inputs, cast to fp16
    for tag, cfg, n_kv, seed in (("res_in", MG.RES_IN, 64, 21), ("res_out", MG.RES_OUT, 16, 22)):
        wd = synth.resampler_weights(seed, "", cfg["grid"], cfg["embed"])
        m = qwen_mod.Resampler(grid_size=cfg["grid"], embed_dim=cfg["embed"], num_heads=cfg["heads"], kv_dim=cfg["embed"]).eval()
        m.load_state_dict(wd, strict=False)
        x = synth.normal_like(seed + 100, (3, n_kv, cfg["embed"]), 1.0)
        assert torch.equal(x, g32[tag + ".x"])
        with torch.no_grad():
            y = m.to(H)(x.to(H))
        mine = O.resampler_forward({k: v.to(H) for k, v in wd.items()}, "", x.to(H), cfg["heads"])
        MG.check(tag + " fp16", mine, y, 1e-2)
        out[tag + "_f16.y"] = y
        print("  %s: fp16 vs fp32 %.2e" % (tag, MG.rel(y, g32[tag + ".y"])))
    c = MG.VIT
    wd = synth.vit_weights(31, c["width"], c["layers"], c["heads"], c["mlp_width"], c["patch"], c["out_dim"], c["n_queries"])
    m = qwen_mod.VisionTransformerWithAttnPool(image_size=c["image"], patch_size=c["patch"], width=c["width"], layers=c["layers"],
                                               heads=c["heads"], mlp_ratio=c["mlp_width"] / c["width"], n_queries=c["n_queries"],
                                               output_dim=c["out_dim"]).eval()
    m.load_state_dict(wd, strict=False)
    x = g32["vit.x"]
    with torch.no_grad():
        y = m.to(H)(x.to(H))
    out["vit_f16.y"] = y
    print("  vit: fp16 vs fp32 %.2e" % MG.rel(y, g32["vit.y"]))
    c = MG.XLV2
    wd = synth.resampler_xlv2_weights(41, **c)
    m = ipa_mod.ResamplerXLV2(**c).eval()
    m.load_state_dict(wd, strict=False)
    x = g32["xlv2.x"]
    with torch.no_grad():
        ctx, pooled = m.to(H)(x.to(H))
    out["xlv2_f16.ctx"], out["xlv2_f16.pooled"] = ctx, pooled
    print("  xlv2: fp16 vs fp32 ctx %.2e pooled %.2e" % (MG.rel(ctx, g32["xlv2.ctx"]), MG.rel(pooled, g32["xlv2.pooled"])))
    # one ViT-G-width block (1664 / 16 heads / MLP 8192, 1024 tokens): every 32nd row
    c = MG.VITBLK
    wd = synth.vit_block_weights(61, c["width"], c["mlp_width"], dtype=H)
    blk = qwen_mod.VisualAttentionBlock(c["width"], c["heads"], c["mlp_width"] / c["width"], norm_layer=partial(torch.nn.LayerNorm, eps=1e-6)).eval()
    blk.load_state_dict(wd, strict=False)
    blk = blk.to(H)
    x = synth.normal_like(161, (c["tokens"], 1, c["width"]), 1.0, dtype=H)
    with torch.no_grad():
        ref = blk(x)
    mine = O.vit_block_forward(wd, "", x.transpose(0, 1), c["heads"]).transpose(0, 1)
    MG.check("vitblk_f16", mine, ref, 1e-2)
    out["vitblk_f16.y_rows"] = ref[::c["row_stride"], 0]
    print("  vit block: fp16 rows vs fp32 rows %.2e" % MG.rel(out["vitblk_f16.y_rows"], g32["vitblk_f32.y_rows"]))
    out = {k: v.contiguous() for k, v in out.items()}
    path = os.path.join(MG.GOLD, "hotpath_tiny_fp16.safetensors")
    save_file(out, path, metadata={"generator": "oracle/make_golden_fp16.py", "torch": torch.__version__})
    print("wrote %d tensors, %.1f KiB" % (len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
