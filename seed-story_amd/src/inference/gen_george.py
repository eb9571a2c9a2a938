"""MI355X-native story-generation driver — the flow of the reference's ``src/inference/gen_george.py``
(load everything from the hydra-style YAMLs, then per story: ViT encode -> ``agent.generate`` ->
``adapter.generate`` -> save JPEG, 8-image sliding window, 25 steps) on the drop-in modules of this repo.

    PYTHONPATH=seed-story_amd python -m src.inference.gen_george --val data/json/val.jsonl        # real checkpoints
    PYTHONPATH=seed-story_amd python -m src.inference.gen_george --synthetic --tiny --steps 4     # random weights

Differences from the reference script:
* structural only: one ``main()`` instead of module-level code; the image transform runs on the device
  (``transform.to(device, dtype)``: Pillow-exact resize + normalise as HIP kernels, bit-identical tensor).
* **intentional deviation (default mode)**: the context is managed at token-id level (``seedstory.story.StoryContext``) —
  the generated caption ids in front of ``<img>`` are appended verbatim and eviction cuts at the first ``</img>`` id.
  The reference appends the decoded, regex-scrubbed TEXT (everything generated minus ``<...>`` tokens, ``</s>`` included),
  re-tokenises the whole prompt every step, and on eviction also drops ``len('[INST]')`` = 6 more characters of the next
  caption (:196,231-243); after the first step its token stream can therefore differ from the id-level one at
  tokenisation boundaries and by those six characters.
* ``--parity`` reproduces the reference's string surgery byte for byte (``seedstory.story.PromptStory``) for A/B checks
  against the reference on real checkpoints; it re-prefills everything each step like the reference does.
"""
import argparse
import json
import os
import re
import zlib

import torch

from seedstory import instantiate as I
from seedstory.story import PromptStory, StoryContext

BOI_TOKEN = '<img>'
EOI_TOKEN = '</img>'
IMG_TOKEN = '<img_{:05d}>'
CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "configs")


SUBTITLE_BAR = 80          # pixel height of the caption bar under the picture
SUBTITLE_LINE = 14         # line pitch of PIL's default bitmap font as the reference spaces it


def add_subtitle(original_image, text):
    """Caption bar under an image, as the reference writes its ``NN.jpg`` / ``000start_image.jpg`` files
    (gen_george.py:114-149): an RGB canvas 80 px taller than the picture, black, picture pasted at the top; the text is
    cut in the MIDDLE OF ITS CHARACTERS (``len(text) // 2``, not at a word boundary) into two lines drawn in white with
    PIL's default font at x = 10, the first at ``height + (80 - 14) // 2``, the second 14 px below.  Host-side PIL work
    (the step after the path: SURVEY section 8f row 3 keeps it on the host)."""
    from PIL import Image, ImageDraw
    w, h = original_image.width, original_image.height
    canvas = Image.new("RGB", (w, h + SUBTITLE_BAR), "black")
    canvas.paste(original_image, (0, 0))
    pen = ImageDraw.Draw(canvas)
    half = len(text) // 2
    y0 = h + (SUBTITLE_BAR - SUBTITLE_LINE) // 2
    for k, line in enumerate((text[:half], text[half:])):
        pen.text((10, y0 + k * SUBTITLE_LINE), line, fill="white")
    return canvas


class SyntheticTokenizer:
    """Stand-in used with --synthetic: ids 3..vocab-67 are 'text' (one id per whitespace-separated word: ``w<id>`` maps
    to ``id``, any other word to a CRC of its bytes), the last 66 ids are ``<img>``, ``<img_00000>``.., ``</img>``."""
    bos_token_id, eos_token_id = 1, 2

    def __init__(self, vocab=32066):
        self.vocab = vocab
        self.img = list(range(vocab - 66, vocab))
        names = [BOI_TOKEN] + [IMG_TOKEN.format(i) for i in range(64)] + [EOI_TOKEN]
        self.added = dict(zip(names, self.img))
        self.names = {i: n for n, i in self.added.items()}

    def encode(self, s, add_special_tokens=False):
        ids = [self.bos_token_id] if add_special_tokens else []
        for part in re.split(r'(</?img(?:_\d{5})?>)', s):
            if part in self.added:
                ids.append(self.added[part])
                continue
            for w in part.split():
                m = re.fullmatch(r'w(\d+)', w)
                ids.append(int(m.group(1)) if m and 3 <= int(m.group(1)) < self.vocab - 66
                           else 3 + zlib.crc32(w.encode()) % (self.vocab - 70))
        return ids

    def decode(self, ids, skip_special_tokens=False):
        return " ".join(self.names[int(i)] if int(i) in self.names else "w%d" % int(i) for i in ids)


def build(args, device, dtype):
    if not args.synthetic:
        tokenizer = I.instantiate(I.load(os.path.join(CFG, "tokenizer/clm_llama_tokenizer.yaml")))
        transform = I.instantiate(I.load(os.path.join(CFG, "processer/qwen_448_transform.yaml")))
        vit = I.instantiate(I.load(os.path.join(CFG, "visual_tokenizer/qwen_vitg_448.yaml"))).eval().to(device, dtype=dtype)
        llm = I.instantiate(I.load(os.path.join(CFG, "clm_models/llama2chat7b_lora.yaml")), torch_dtype=args.dtype)
        agent = I.instantiate(I.load(os.path.join(CFG, "clm_models/agent_7b_sft.yaml")), llm=llm).eval().to(device, dtype=dtype)
        from seedstory.diffusion import AutoencoderKL, EulerDiscreteScheduler, UNet2DConditionModel
        sched = EulerDiscreteScheduler.from_pretrained(args.sdxl, subfolder="scheduler")
        vae = AutoencoderKL.from_pretrained(args.sdxl, subfolder="vae").to(device, dtype=dtype)
        unet = UNet2DConditionModel.from_pretrained(args.sdxl, subfolder="unet").to(device, dtype=dtype)
        adapter = I.instantiate(I.load(os.path.join(CFG, "detokenizer/detokenizer_sdxl_qwen_vit_adapted.yaml")), unet=unet)
        adapter = adapter.to(device, dtype=dtype).eval()
        disc = I.instantiate(I.load(os.path.join(CFG, "discrete_model/discrete_identity.yaml"))).to(device).eval()
        adapter.init_pipe(vae=vae, scheduler=sched, visual_encoder=vit, image_transform=transform, discrete_model=disc,
                          dtype=dtype, device=device)
        return tokenizer, transform, vit, agent, adapter
    # ---- synthetic weights (no checkpoints on this box) -------------------------------------------------
    from seedstory.diffusion import AutoencoderKL, EulerDiscreteScheduler, UNet2DConditionModel
    from src.models.discrete_models import DiscreteModleIdentity
    from src.models.qwen_visual import Resampler, VisionTransformerWithAttnPool
    from src.models_clm.modeling_llama_xformer import LlamaConfig, LlamaForCausalLM
    from src.models_clm.models import ContinuousLVLM
    from src.models_ipa.adapter_modules import SDXLAdapter
    from src.models_ipa.resampler import ResamplerXLV2
    from src.processer.transforms import get_transform
    t = args.tiny
    H, heads, layers, inter, vocab = (256, 2, 2, 512, 1066) if t else (4096, 32, 32, 11008, 32066)
    tokenizer = SyntheticTokenizer(vocab)
    cfg = LlamaConfig(hidden_size=H, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                      vocab_size=vocab)
    llm = LlamaForCausalLM(cfg).to(device, dtype=dtype).init_synthetic(0)
    llm.use_kv_cache_head = False
    rin = Resampler(grid_size=8, embed_dim=H, num_heads=heads, kv_dim=H).to(device, dtype=dtype).init_synthetic(1)
    rout = Resampler(grid_size=16, embed_dim=H, num_heads=heads, kv_dim=H).to(device, dtype=dtype).init_synthetic(2)
    agent = ContinuousLVLM(llm, rin, rout).eval()
    if t:
        vit = VisionTransformerWithAttnPool(image_size=224, patch_size=14, width=208, layers=2, heads=2, mlp_ratio=2.0,
                                            n_queries=256, output_dim=H)
    else:
        vit = VisionTransformerWithAttnPool(image_size=448, patch_size=14, width=1664, layers=48, heads=16,
                                            mlp_ratio=4.9231, output_dim=H)
    vit = vit.to(device, dtype=dtype).init_synthetic(3)
    import sys
    unet_cfg = vae_cfg = None
    if t:
        unet_cfg = dict(in_channels=4, out_channels=4, block_out_channels=(64, 128, 256), layers_per_block=2,
                        transformer_layers=(0, 1, 2), num_heads=(1, 2, 4), cross_attention_dim=128,
                        addition_time_embed_dim=32, pooled_dim=80, norm_groups=32)
        vae_cfg = dict(latent_channels=4, out_channels=3, block_out_channels=(32, 64, 64, 64), layers_per_block=2,
                       norm_groups=32, scaling_factor=0.13025)
    unet = UNet2DConditionModel(unet_cfg).to(device, dtype=dtype).init_synthetic(4)
    vae = AutoencoderKL(vae_cfg).to(device, dtype=dtype).init_synthetic(5)
    rs = (ResamplerXLV2(dim=128, depth=2, dim_head=32, heads=4, num_queries=8, embedding_dim=H, output1_dim=48,
                        output2_dim=80, ff_mult=4) if t else
          ResamplerXLV2(dim=1024, depth=4, dim_head=64, heads=16, num_queries=64, embedding_dim=H, output1_dim=768,
                        output2_dim=1280, ff_mult=4)).to(device, dtype=dtype).init_synthetic(6)
    adapter = SDXLAdapter.from_pretrained(unet=unet, resampler=rs).eval()
    size = 224 if t else 448
    adapter.init_pipe(vae=vae, scheduler=EulerDiscreteScheduler(), visual_encoder=vit,
                      image_transform=get_transform('clip', image_size=size, keep_ratio=False),
                      discrete_model=DiscreteModleIdentity(), dtype=dtype, device=device)
    return tokenizer, adapter.image_transform, vit, agent, adapter


def run_story(args, j, question, image, tokenizer, transform, vit, agent, adapter, device, dtype):
    save_folder = os.path.join(args.out, "val_%d" % j)
    os.makedirs(save_folder, exist_ok=True)
    add_subtitle(image, question).save(os.path.join(save_folder, "000start_image.jpg"))     # gen_george.py:161-163
    boi = tokenizer.encode(BOI_TOKEN, add_special_tokens=False)[0]
    eoi = tokenizer.encode(EOI_TOKEN, add_special_tokens=False)[0]
    img_all = tokenizer.encode(BOI_TOKEN + ''.join(IMG_TOKEN.format(i) for i in range(64)) + EOI_TOKEN, add_special_tokens=False)
    image_tensor = transform(image).unsqueeze(0).to(device, dtype=dtype)
    if args.parity:      # the reference's string-level prompt surgery, byte for byte (re-tokenise + '[INST]' skip)
        ctx = PromptStory(tokenizer, window=args.window)
        with torch.no_grad():
            ctx.start(question, vit(image_tensor))
    else:                # id-level context: generated ids kept verbatim
        ctx = StoryContext(tokenizer.bos_token_id, boi, eoi, img_all[1:-1], window=args.window)
        with torch.no_grad():
            ctx.start(tokenizer.encode(question, add_special_tokens=False), vit(image_tensor))   # gen_george.py:168-188
    llama = agent.llm.base_model.model if hasattr(agent.llm, "base_model") else agent.llm
    llama.use_kv_cache_head = False                                                         # :165
    size = args.image_size
    forced = None
    for step in range(1, args.steps + 1):
        if args.synthetic:   # random weights never emit <img>: force a caption + <img> (SURVEY section 8d schedule)
            g = torch.Generator().manual_seed(1000 * j + step)
            forced = torch.randint(3, 1000, (args.caption_tokens,), generator=g).tolist() + [boi]
        ids_mask, emb_mask = ctx.masks(device)
        out = agent.generate(tokenizer=tokenizer, input_ids=ctx.input_ids(device), image_embeds=ctx.image_embeds,
                             embeds_cmp_mask=emb_mask, ids_cmp_mask=ids_mask, max_new_tokens=500, num_img_gen_tokens=64,
                             forced_tokens=forced)
        text = re.sub(r'\s*<[^>]*>\s*', ' ', out['text']).strip()
        with open(os.path.join(save_folder, "text.txt"), "a+") as f:
            f.write(text + "\n")
        with open(os.path.join(save_folder, "token.txt"), "a+") as f:
            f.write("context token: {}\n".format((1, len(ctx.ids))))
        if not out['has_img_output']:
            break
        images = adapter.generate(image_embeds=out['img_gen_feat'], num_inference_steps=args.diffusion_steps,
                                  height=size, width=size, input_image_size=transform.size)
        images[0].save(os.path.join(save_folder, 'ori_{:02d}.jpg'.format(step)))                 # :217-218
        add_subtitle(images[0], text).save(os.path.join(save_folder, '{:02d}.jpg'.format(step)))   # :212-222
        ctx.advance(out)                                                                    # :224, :231, :235-239
    return save_folder


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--val", default="data/json/val.jsonl")
    ap.add_argument("--image-root", default="data/image/george_full")
    ap.add_argument("--sdxl", default="pretrained/stable-diffusion-xl-base-1.0")
    ap.add_argument("--out", default="output")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--steps", type=int, default=24)             # story_len 25 (:205)
    ap.add_argument("--window", type=int, default=8)             # window_size (:206)
    ap.add_argument("--diffusion-steps", type=int, default=50)   # :210
    ap.add_argument("--image-size", type=int, default=1024)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--caption-tokens", type=int, default=48)
    ap.add_argument("--stories", type=int, default=1)
    ap.add_argument("--parity", action="store_true",
                    help="string-level prompt bookkeeping exactly as the reference driver (decode -> scrub -> re-tokenise, "
                         "'[INST]'-length skip on eviction) instead of the id-level context")
    args = ap.parse_args()
    device = "cuda:0"
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    tokenizer, transform, vit, agent, adapter = build(args, device, dtype)
    if hasattr(transform, "to"):
        transform.to(device, dtype)          # resize + CLIP-normalise as HIP kernels (PIL-exact), output already in HBM
    from PIL import Image
    if args.synthetic:
        data = [{"images": [None], "captions": ["a synthetic story question number %d" % i]} for i in range(args.stories)]
    else:
        data = [json.loads(l) for l in open(args.val)]
    for j, d in enumerate(data):
        if args.synthetic:
            image = Image.new("RGB", (320, 240), (40 * j % 255, 120, 200))
        else:
            image = Image.open(os.path.join(args.image_root, d['images'][0])).convert('RGB')
        folder = run_story(args, j, d['captions'][0], image, tokenizer, transform, vit, agent, adapter, device, dtype)
        print("story", j, "->", folder)


if __name__ == "__main__":
    main()
