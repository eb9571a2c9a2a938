"""Multi-GPU partitioning of the story path — one process per GPU, ``torch.distributed`` over
RCCL (backend "nccl" on ROCm; "gloo" in the CPU tests).  No all-reduce anywhere on the data path.

* **story replicas** (``stories_for_rank``; ``bench.py --gpus N``): the outer loop over stories
  (gen_george.py:152) has no cross-iteration state -> stories are dealt to ranks, zero communication.  This is the
  throughput mode: every rank runs the whole pipeline for its own stories.

* **slot ring** (``run_slot_ring``; ``bench.py --gpus N --partition slots``, BASELINE configs[3]): ONE story stream
  (the S lock-step stories of a node) whose image slots are sharded over the GPUs.  MLLM step t+1 needs only the
  regressed feature of step t (gen_george.py:224), never the rendered pixels, so the stream is advanced by a
  *rotating owner*: round r (= story step r of every resident story) belongs to rank ``r mod N``.  The owner

    1. runs the MLLM half of the round on its mirror of the context (KV-cached continuation: 65 new rows + 115
       decode tokens per story),
    2. **broadcasts what the mirrors need to own a later round**, packed into ONE flat buffer (one collective per
       round): the generated token ids, the new image feature(s) [S, 256, 4096] and the KV-cache rows it appended
       (L x 2 x [heads, rows, hd], 0.5 MiB per row: ~57 MiB per story and round) — the RCCL broadcast of the MLLM KV
       cache over xGMI that north_star names; xGMI is point-to-point, so this is N-1 link-bound copies of a few ms.
       When the round's context update evicts an image (every round once the 8-image window is full), the next owner
       re-prefills the window from ids + features anyway, so NO KV rows are shipped for that round,
    3. renders its S images (30 UNet steps + VAE, ~2 s) on a second stream / host thread while the next owner is
       already computing round r+1.

  In steady state a node finishes one round per max(t_MLLM, t_render / N); with N >= t_render / t_MLLM the MLLM
  chain is the limiter, which is why replicas remain the throughput mode and the ring is the
  latency mode of a single stream.  The schedule below is engine-agnostic (a ``backend`` object supplies the MLLM
  round, the payload and the render), which is how the world-size-2 gloo test drives it on CPU.
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import torch
import torch.distributed as dist


def stories_for_rank(n_stories, rank, world):
    return list(range(rank, n_stories, world))


def slot_owner(slot, world, first=0):
    """Rank that owns (computes and renders) image slot / round `slot` of the stream."""
    return (first + slot) % world if world > 1 else 0


def slots_for_rank(n_slots, rank, world, first=0):
    return [s for s in range(n_slots) if slot_owner(s, world, first) == rank]


def broadcast_feature(feat, src=0, group=None):
    """One-to-all of the regressed image feature (in place on non-src ranks)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(feat, src=src, group=group)
    return feat


def send_feature(feat, src, dst, group=None):
    """Point-to-point hand-off of one slot's feature (xGMI is point-to-point: one link, 2 MiB)."""
    if not dist.is_initialized() or src == dst:
        return feat
    rank = dist.get_rank(group)
    if rank == src:
        dist.send(feat, dst=dst, group=group)
    elif rank == dst:
        dist.recv(feat, src=src, group=group)
    return feat


def broadcast_kv(engine_k, engine_v, length, src=0, group=None, start=0):
    """Broadcast KV-slab rows [start, length) of [L, H, cap, hd] planes (in place on the mirrors)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1 and length > start:
        k = engine_k[:, :, start:length].contiguous()
        v = engine_v[:, :, start:length].contiguous()
        dist.broadcast(k, src=src, group=group)
        dist.broadcast(v, src=src, group=group)
        if dist.get_rank(group) != src:
            engine_k[:, :, start:length].copy_(k)
            engine_v[:, :, start:length].copy_(v)
    return engine_k, engine_v


def max_over_ranks(seconds, device=None, group=None):
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


# ---------------------------------------------------------------------------------------------------------------------
# slot ring
# ---------------------------------------------------------------------------------------------------------------------
META_LEN = 64          # int64 header broadcast ahead of every round's payload


class SlotRingBackend:
    """What ``run_slot_ring`` needs from an engine.  All methods are called on the communication thread except
    ``render`` (render thread).

    mllm_round(r) -> (meta, tensors): owner only.  Runs the MLLM half of round r on the local mirror and returns the
        payload that turns every other rank's mirror into a valid owner of a later round: ``meta`` (< META_LEN ints
        describing the tensors) and the tensors themselves (contiguous, on the communication device).
    alloc(meta) -> tensors: mirrors only; empty receive buffers matching ``meta``.
    apply(r, meta, tensors): mirrors only; install the payload (KV rows into the slab, context bookkeeping).
    render(r, meta, tensors): owner only, on the render thread; the de-tokenizer half of round r.
    """

    def mllm_round(self, r):
        raise NotImplementedError

    def alloc(self, meta):
        raise NotImplementedError

    def apply(self, r, meta, tensors):
        raise NotImplementedError

    def render(self, r, meta, tensors):
        raise NotImplementedError


def _bcast(t, src, group=None):
    """Broadcast in place.  RCCL moves device tensors directly; under gloo (the CPU tests and the single-device flow test
    of the N > 1 path) device tensors are staged through host memory — gloo's CUDA transport is not part of ROCm builds."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


def _flat_layout(tensors):
    """Byte offsets of the tensors inside ONE flat payload buffer (16-byte aligned pieces)."""
    offs, off = [], 0
    for t in tensors:
        offs.append(off)
        off += (t.numel() * t.element_size() + 15) // 16 * 16
    return offs, off


def pack_payload(tensors, device):
    """All payload tensors of a round in ONE contiguous byte buffer: one collective per round instead of one per
    tensor (2 + 2 S of them), and xGMI sees a single large message."""
    offs, total = _flat_layout(tensors)
    flat = torch.empty(max(total, 16), dtype=torch.uint8, device=device)
    for t, o in zip(tensors, offs):
        n = t.numel() * t.element_size()
        if n:      # straight into its place in the flat buffer (t may be a strided view of the KV slab: no staging copy)
            flat[o:o + n].view(t.dtype).view(t.shape).copy_(t)
    return flat


def unpack_payload(flat, like):
    """Views into ``flat`` with the shapes / dtypes of ``like`` (the mirrors' ``alloc`` result)."""
    offs, total = _flat_layout(like)
    assert flat.numel() >= total
    out = []
    for t, o in zip(like, offs):
        n = t.numel() * t.element_size()
        out.append(flat[o:o + n].view(t.dtype).view(t.shape) if n else t)
    return out


def run_slot_ring(backend, n_rounds, rank, world, group=None, meta_device="cpu", first_round=0):
    """Advance the stream by ``n_rounds`` rounds (round r is owned by rank (first_round + r) mod world).  Returns
    the list of rounds this rank rendered.  Collective order is identical on every rank: per round ONE header
    broadcast followed by ONE payload broadcast (all tensors packed into a flat byte buffer), both from the round's
    owner.  The collectives are issued whenever a process group exists — also at world size 1, so that a single-GPU
    run exercises the RCCL initialisation and broadcast path."""
    rendered = []
    pool = ThreadPoolExecutor(max_workers=1)
    pending = []
    err = []
    # collectives run on the group this ring was GIVEN: a default group wider than `world` must not be touched
    comm = world > 1
    if not comm and dist.is_available() and dist.is_initialized():
        comm = dist.get_world_size(group) == world
    if comm:
        assert dist.get_world_size(group) == world, (dist.get_world_size(group), world)
    pdev = getattr(backend, "device", None)        # where payload buffers live when `alloc` only describes them (meta tensors)

    def render_job(r, meta, tensors):
        try:
            backend.render(r, meta, tensors)
        except BaseException as ex:      # surfaced by the communication thread below
            err.append(ex)

    for i in range(n_rounds):
        r = first_round + i
        owner = slot_owner(r, world)
        header = torch.zeros(META_LEN, dtype=torch.int64, device=meta_device)
        if rank == owner:
            meta, tensors = backend.mllm_round(r)
            assert len(meta) < META_LEN
            header[0] = len(meta)
            header[1:1 + len(meta)] = torch.tensor(list(meta), dtype=torch.int64)
        if comm:
            _bcast(header, owner, group)
        if rank != owner:
            meta = header[1:1 + int(header[0])].tolist()
            tensors = backend.alloc(meta)
        if comm and tensors:
            dev_ = tensors[0].device if tensors[0].device.type != "meta" else pdev
            if rank == owner:       # (stream-ordered behind the MLLM half; RCCL waits on the stream, gloo stages through .cpu())
                flat = pack_payload(tensors, dev_)
            else:
                flat = torch.empty(max(_flat_layout(tensors)[1], 16), dtype=torch.uint8, device=dev_)
            _bcast(flat, owner, group)
            if rank != owner:
                tensors = unpack_payload(flat, tensors)
        if rank == owner:
            rendered.append(r)
            pending.append(pool.submit(render_job, r, meta, tensors))
        else:
            backend.apply(r, meta, tensors)
        if err:
            raise err[0]
    for f in pending:
        f.result()
    pool.shutdown()
    if err:
        raise err[0]
    return rendered


class StoryRingBackend(SlotRingBackend):
    """The slot ring over the real engines (used by ``bench.py --partition slots``): S lock-step synthetic stories,
    mirrored on every rank.  ``bm`` is the bench module (its Story class / mllm_part / advance_context are the
    single-GPU schedule's pieces, reused unchanged)."""

    def __init__(self, bm, eng, rin, rout, vit, adapter, spg, device, dtype, diffusion_steps, seed0=4242):
        self.bm, self.eng, self.rin, self.rout, self.vit, self.adapter = bm, eng, rin, rout, vit, adapter
        self.spg, self.device, self.dtype, self.steps = spg, device, dtype, diffusion_steps
        self.story_no = seed0
        self.sts = None
        self.render_stream = torch.cuda.Stream(device=device) if adapter is not None else None
        self.lock = threading.Lock()
        self._ready = {}            # round -> event recorded behind the round's MLLM half (the render stream waits on it)

    def _stories(self):
        bm = self.bm
        if self.sts is None or self.sts[0].step >= bm.STORY_LEN:
            self.sts = []
            for _ in range(self.spg):
                self.story_no += 1
                self.sts.append(bm.Story(self.story_no, self.device))
        return self.sts

    def mllm_round(self, r):
        bm = self.bm
        sts = self._stories()
        first = sts[0].step == 0
        S = [len(st.ids) for st in sts]
        full = [first or st.evicted_last for st in sts]          # this round re-prefills the whole window
        bm.mllm_part(sts, self.eng, self.rin, self.rout, self.vit, True)
        forced = [st.last_forced for st in sts]                  # the round's forced ids, recorded by mllm_part
        # payload: forced / generated ids, the image features appended this round, the KV rows a later owner needs
        ids = torch.tensor(forced, dtype=torch.int32, device=self.device)                       # [S, 115]
        n_new = 2 if first else 1                                 # step 0 also appends the first image's ViT feature
        new_embeds = torch.stack([st.image_embeds[-n_new:] for st in sts]).contiguous()        # eviction drops from the FRONT
        tensors = [ids, new_embeds]
        meta = [self.spg, ids.shape[1], n_new, int(first)]
        for b, st in enumerate(sts):
            lo = 0 if full[b] else S[b] - 65
            hi = S[b] + 49                                       # rows the next continuation keeps (S_next - 65)
            if st.evicted_last:
                # this round's context update evicted an image: whoever owns the NEXT round re-prefills the whole window
                # from ids + features (positions shift), so no cached row of this round is ever read again — ship none
                # (from story step WINDOW on this is every round: 0 bytes of KV instead of 913 rows x 0.5 MiB per story)
                lo = hi = 0
            self.eng.select(b)
            tensors += [self.eng.k_cache[:, :, lo:hi], self.eng.v_cache[:, :, lo:hi]]      # views: packed without a staging copy
            meta += [lo, hi]
        if self.render_stream is not None:
            # the render of this round runs on ANOTHER stream (and thread): it must not read the regressed features before the
            # MLLM half that produces them has run (under RCCL nothing between here and the render synchronises the host)
            ev = torch.cuda.Event()
            ev.record()
            self._ready[r] = ev
        return meta, tensors

    def alloc(self, meta):
        spg, T, n_new = meta[0], meta[1], meta[2]
        e = self.eng
        # DESCRIPTORS only (meta tensors: shape + dtype): the receive side allocates ONE flat buffer and views into it
        out = [torch.empty(spg, T, dtype=torch.int32, device="meta"),
               torch.empty(spg, n_new, 256, e.hidden, dtype=self.dtype, device="meta")]
        for b in range(spg):
            lo, hi = meta[4 + 2 * b], meta[5 + 2 * b]
            shp = (e.n_layers, e.n_heads, hi - lo, e.hd)
            out += [torch.empty(shp, dtype=self.dtype, device="meta"), torch.empty(shp, dtype=self.dtype, device="meta")]
        return out

    def apply(self, r, meta, tensors):
        bm = self.bm
        first = bool(meta[3])
        sts = self._stories()
        ids, new_embeds = tensors[0].tolist(), tensors[1]
        for b, st in enumerate(sts):
            lo, hi = meta[4 + 2 * b], meta[5 + 2 * b]
            self.eng.select(b)
            if hi > lo:
                self.eng.k_cache[:, :, lo:hi].copy_(tensors[2 + 2 * b])
                self.eng.v_cache[:, :, lo:hi].copy_(tensors[3 + 2 * b])
            # the same context update mllm_part performed on the owner (bench.advance_context)
            st.forced()                                                       # keep the story's RNG stream in step
            add = new_embeds[b]
            if first:
                st.image_embeds = add[:1]                                     # ViT feature of the story's first image
                add = add[1:]
            bm.advance_context(st, ids[b], add)

    def render(self, r, meta, tensors):
        if self.adapter is None:
            return
        feat = tensors[1][:, -1]                                              # [S, 256, 4096]: this round's regressed feature
        torch.cuda.set_device(self.device)
        ev = self._ready.pop(r, None)
        with self.lock, torch.cuda.stream(self.render_stream):
            if ev is not None:
                self.render_stream.wait_event(ev)
            self.adapter.generate(image_embeds=feat.contiguous(), num_inference_steps=self.steps, output_type="pt")
        self.render_stream.synchronize()


def bench_slot_partition(args, rank, world, device, dtype, bm):
    """``bench.py --gpus N --partition slots``: time `--steps` rounds of the slot ring (after untimed warm-up
    rounds), barrier + max over ranks, one JSON line from rank 0."""
    import json
    import os
    import time
    eng, _ = bm.build_engine(device, dtype, args.stories_per_gpu)
    rin, rout, vit = bm.build_frontend(device, dtype)
    adapter = None if args.mllm_only else bm.build_detokenizer(device, dtype, vit)
    be = StoryRingBackend(bm, eng, rin, rout, vit, adapter, args.stories_per_gpu, device, dtype, args.diffusion_steps)
    gloo = bool(os.environ.get("SS_BENCH_SINGLE_DEVICE"))
    meta_dev = "cpu" if gloo else device
    if os.environ.get("SS_BENCH_WATCHDOG_S"):       # flow tests: dump every thread's stack and exit instead of hanging
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SS_BENCH_WATCHDOG_S"]), exit=True)
    # every rank owns one MLLM round and renders once before the clock starts (tile-table entries, graph capture)
    run_slot_ring(be, max(world, args.warmup), rank, world, meta_device=meta_dev)
    be.sts = None
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    mine = run_slot_ring(be, args.steps, rank, world, meta_device=meta_dev)
    torch.cuda.synchronize()
    dist.barrier()
    my_s = time.perf_counter() - t0
    dt_s = max_over_ranks(my_s, "cpu" if gloo else device)
    mine_t = torch.tensor([my_s, float(len(mine))], dtype=torch.float64, device="cpu" if gloo else device)
    every = [torch.zeros_like(mine_t) for _ in range(world)]
    dist.all_gather(every, mine_t)
    per_rank = [{"rank": i, "seconds": round(float(v[0]), 4), "rounds_rendered": int(v[1]),
                 "story_steps_rendered": int(v[1]) * args.stories_per_gpu} for i, v in enumerate(every)]
    roof = cpu = None
    if rank == 0:       # the same roofline section as the replica partition (rank 0's engines); CPU baseline at N = 1 only
        roof = None if args.no_roofline else bm.measure_roofline(eng, adapter, args.stories_per_gpu, 1, device, dtype, args)
        if world == 1 and not args.no_cpu_baseline:
            cpu = bm.cpu_baseline(with_sdxl=not args.mllm_only, diffusion_steps=args.diffusion_steps)
    dist.barrier()
    if hasattr(bm, "flush_c_stdio"):
        bm.flush_c_stdio()          # RCCL's version banner (C stdio) goes out before the JSON line, not behind it at exit
    if rank == 0:
        spg = args.stories_per_gpu
        out = {"metric": "story-steps/sec (text + 1024x1024 image)", "value": round(args.steps * spg / dt_s, 4),
               "unit": "story-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt_s / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "BASELINE configs[3]: ONE stream of %d lock-step stories (story length %d) per node; round r "
                                      "owned by rank r mod %d: MLLM half (KV-cached continuation) + RCCL broadcast of ids, "
                                      "image feature and the KV rows it appended, render on the owner under the next rounds"
                                      % (spg, bm.STORY_LEN, world),
                          "partition": "slots", "stories_per_gpu": spg, "diffusion_steps": args.diffusion_steps,
                          "parallelism": "slot ring x%d (rotating owner, KV-cache broadcast over xGMI)" % world},
               "rounds_rendered_by_rank0": mine, "backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(),
               "per_rank": per_rank,
               "collectives_per_round": "1 header + 1 flat payload broadcast (ids, image feature, appended KV rows; no KV rows "
                                        "for a round whose context update evicted an image)",
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))
    dist.destroy_process_group()
