"""MI355X-native "story visualisation" driver — the flow of the reference's
``src/inference/vis_george_sink.py``: captions are GIVEN (teacher-forced context), the model only produces the
image tokens, and the context is bounded by the multimodal attention sink on the KV cache.

The reference script computes the sliced cache (:266-295) but then calls ``generate(past_key_values=None)``
(:316) with ``use_kv_cache_head=False`` (:173), i.e. as released it re-prefills everything and the sink is
dead code.  Here ``--cache-mode img_head_tail`` (default, the reference's ``cache_mode``
constant, :42) runs the *intended* algorithm: the KV slab is truncated to the prompt (:243), re-packed by
``StoryContext.evict_sink`` (= ``ss_llama_kv_gather``) when more than ``window`` images are live, and the next
``generate`` continues from the cached prefix through the reference's own ``past_key_values`` /
``kv_cache_head`` / ``use_kv_cache_head`` protocol (modeling_llama_xformer.py:676-678, 804-826).
``--cache-mode none`` reproduces the as-released behaviour (full re-prefill, sliding window).

    PYTHONPATH=seed-story_amd python -m src.inference.vis_george_sink --synthetic --tiny --steps 6 --window 2
"""
import argparse
import json
import os

import torch

from seedstory.story import StoryContext
from src.inference.gen_george import BOI_TOKEN, EOI_TOKEN, IMG_TOKEN, build


def run_story(args, j, start_text, captions, image, tokenizer, transform, vit, agent, adapter, device, dtype):
    save_folder = os.path.join(args.out, "val_%d" % j)
    os.makedirs(save_folder, exist_ok=True)
    boi = tokenizer.encode(BOI_TOKEN, add_special_tokens=False)[0]
    eoi = tokenizer.encode(EOI_TOKEN, add_special_tokens=False)[0]
    img_all = tokenizer.encode(BOI_TOKEN + ''.join(IMG_TOKEN.format(i) for i in range(64)) + EOI_TOKEN, add_special_tokens=False)
    enc = lambda s: tokenizer.encode(s, add_special_tokens=False)  # noqa: E731
    llama = agent.llm.base_model.model if hasattr(agent.llm, "base_model") else agent.llm
    llama.kv_cache_head = None                                   # :171-173
    llama.past_key_values = None
    sink = args.cache_mode == "img_head_tail"
    llama.use_kv_cache_head = sink
    ctx = StoryContext(tokenizer.bos_token_id, boi, eoi, img_all[1:-1], window=args.window)
    with torch.no_grad():
        ctx.start(enc(start_text), vit(transform(image).unsqueeze(0).to(device, dtype=dtype)))
    ctx.ids = ctx.ids + enc(captions[0])                          # prompt = instruction(start + image_tokens) + text (:181)
    past = None
    cached = 0                                                    # window tokens whose KV is already in the slab
    forced = [boi] if args.synthetic else None                    # random weights never open an image on their own
    for step in range(1, min(args.steps, len(captions)) + 1):
        ids_mask, emb_mask = ctx.masks(device)
        if sink and past is not None:
            llama.kv_cache_head = cached
        out = agent.generate(tokenizer=tokenizer, input_ids=ctx.input_ids(device), image_embeds=ctx.image_embeds,
                             embeds_cmp_mask=emb_mask, ids_cmp_mask=ids_mask, max_new_tokens=500, num_img_gen_tokens=64,
                             past_key_values=past, forced_tokens=forced)
        with open(os.path.join(save_folder, "token.txt"), "a+") as f:
            f.write("context token: {} cached: {} sink: {}\n".format((1, len(ctx.ids)), cached, ctx.sink_len))
        if not out['has_img_output']:
            break
        images = adapter.generate(image_embeds=out['img_gen_feat'], num_inference_steps=args.diffusion_steps,
                                  height=args.image_size, width=args.image_size, input_image_size=transform.size)
        images[0].save(os.path.join(save_folder, 'ori_{:02d}.jpg'.format(step)))
        if step >= len(captions):
            break
        eng = llama.engine_for_generation(tuple(img_all))
        prompt_len = len(ctx.ids)
        if sink:
            # keep only the prompt's KV (drop the generated image tokens: their inputs were the placeholder
            # embeddings, the next prompt carries the regressed features instead) — vis_george_sink.py:243-244
            kv_len = ctx.sink_len + prompt_len
            eng.set_lengths(kv_len, prompt_len)
            cached = prompt_len
        ctx.append_step([], out['img_gen_feat'])                  # prompt += image_tokens (:246), features appended (:232)
        ctx.ids = ctx.ids + enc(captions[step])                   # ... + next caption
        if ctx.over_window():
            if sink:
                before = len(ctx.ids)
                kv_len = ctx.evict_sink(eng, kv_len)              # :254-295 on the slab
                cached -= before - len(ctx.ids)                   # kv_cache_head -= eoi + 1  (:293)
            else:
                ctx.evict_recompute()
        past = eng.past_key_values() if sink else None
    return save_folder


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--val", default="data/json/val.jsonl")
    ap.add_argument("--image-root", default="data/image/george_full")
    ap.add_argument("--sdxl", default="pretrained/stable-diffusion-xl-base-1.0")
    ap.add_argument("--out", default="output_vis")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--window", type=int, default=8)
    ap.add_argument("--diffusion-steps", type=int, default=50)
    ap.add_argument("--image-size", type=int, default=1024)
    ap.add_argument("--cache-mode", default="img_head_tail", choices=["img_head_tail", "none"])
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--caption-tokens", type=int, default=48)
    ap.add_argument("--stories", type=int, default=1)
    args = ap.parse_args()
    device = "cuda:0"
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    tokenizer, transform, vit, agent, adapter = build(args, device, dtype)
    if hasattr(transform, "to"):
        transform.to(device, dtype)          # resize + CLIP-normalise as HIP kernels (PIL-exact), output already in HBM
    from PIL import Image
    if args.synthetic:
        words = lambda k, n: " ".join("w%d_%d" % (k, i) for i in range(n))  # noqa: E731
        data = [{"images": [None], "captions": [words(0, 6)] + [words(s + 1, args.caption_tokens) for s in range(args.steps + 1)]}
                for _ in range(args.stories)]
    else:
        data = [json.loads(l) for l in open(args.val)]
    for j, d in enumerate(data):
        image = (Image.new("RGB", (320, 240), (90, 40 * j % 255, 160)) if args.synthetic
                 else Image.open(os.path.join(args.image_root, d['images'][0])).convert('RGB'))
        folder = run_story(args, j, d['captions'][0], d['captions'][1:], image, tokenizer, transform, vit, agent, adapter,
                           device, dtype)
        print("story", j, "->", folder)


if __name__ == "__main__":
    main()
