"""CPU tests of the host logic: C-ABI export surface, config instantiation, id-level story
bookkeeping, and the N>1 partitioning under a world_size-2 gloo group."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "seed-story_amd")


def test_cabi_exports_every_declared_symbol():
    """The shared library loads (no GPU needed) and exports exactly what include/seedstory_hip.h declares."""
    from seedstory import _lib
    lib = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "seedstory_hip.h")).read()
    declared = set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ss_abi_version() == 1


def test_no_cpu_fallback_loud_failure():
    """Without a GPU every compute entry point must fail loudly, never fall back."""
    from seedstory import _lib, ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.SSError):
        ops.rmsnorm(torch.zeros(2, 8), torch.ones(8), 1e-5)


def test_context_and_rccl_entry_points_fail_loudly_without_gpu():
    """ss_create validates the device; without a GPU it reports an error instead of handing out a handle."""
    from seedstory import _lib, comm
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.SSError):
        comm.Context(0)
    with pytest.raises(_lib.SSError):      # NULL communicator / buffer are argument errors, not crashes
        _lib.check(_lib.lib().ss_rccl_bcast(None, None, 0, _lib.SS_BF16, 0, None), "ss_rccl_bcast")


def test_product_never_imports_oracle():
    bad = []
    for dp, _, fns in os.walk(PKG):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                if re.search(r"^\s*(from|import)\s+(seedstory_oracle|synth|ref_shims|oracle)\b", txt, re.M):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_instantiate_reference_style_configs():
    from seedstory import instantiate as I
    cfg = I.load(os.path.join(PKG, "configs", "processer", "qwen_448_transform.yaml"))
    t = I.instantiate(cfg)
    from PIL import Image
    x = t(Image.new("RGB", (300, 200), (10, 200, 30)))
    assert x.shape == (3, 448, 448)
    d = I.instantiate(I.load(os.path.join(PKG, "configs", "discrete_model", "discrete_identity.yaml")))
    assert d.encode_image_embeds(torch.ones(2)).sum() == 2
    agent_cfg = I.load(os.path.join(PKG, "configs", "clm_models", "agent_7b_sft.yaml"))
    agent_cfg["input_resampler"]["embed_dim"] = 256
    agent_cfg["input_resampler"]["num_heads"] = 2
    agent_cfg["input_resampler"]["kv_dim"] = 256
    r = I.instantiate(agent_cfg["input_resampler"])
    assert r.num_queries == 64 and r.query.shape == (64, 256)
    # the reference's own YAMLs resolve too (same _target_ paths) when the tree is available
    ref = "/root/reference/configs/clm_models/agent_7b_sft.yaml"
    if os.path.exists(ref):
        rc = I.load(ref)
        assert rc["_target_"] == agent_cfg["_target_"]
        assert I.locate(rc["output_resampler"]["_target_"]).__name__ == "Resampler"


def test_peft_wrapper_state_dict_layout():
    from src.models_clm.modeling_llama_xformer import LlamaConfig, LlamaForCausalLM
    from src.models_clm.peft_models import get_peft_model_with_resize_embedding
    cfg = LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=1, vocab_size=50)
    m = LlamaForCausalLM(cfg)
    pm = get_peft_model_with_resize_embedding(m, peft_config={"r": 16, "lora_alpha": 32, "target_modules": ["q_proj", "down_proj"],
                                                              "modules_to_save": ["input_layernorm", "norm"]},
                                              vocab_size=66, torch_dtype="bf16")
    keys = set(pm.state_dict().keys())
    assert "base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight" in keys
    assert "base_model.model.model.layers.0.mlp.down_proj.lora_B.default.weight" in keys
    assert "base_model.model.model.layers.0.input_layernorm.modules_to_save.default.weight" in keys
    assert "base_model.model.model.norm.original_module.weight" in keys
    assert pm.base_model.model.model.embed_tokens.weight.shape[0] == 66
    pm.base_model.model.use_kv_cache_head = False          # the attribute path the drivers use
    assert pm.use_kv_cache_head is False and pm.past_key_values is None
    flat = pm.base_model.model.collect_flat_state()
    assert "model.layers.0.input_layernorm.weight" in flat and "model.norm.weight" in flat
    assert "model.layers.0.self_attn.q_proj.lora_A.default.weight" in flat


def test_sink_eviction_index_spec():
    """Clean spec of vis_george_sink.py:266-295 — index arithmetic only."""
    import seedstory_oracle as O
    keep, sink = O.sink_evict_indices(n_kv=300, boi=20, eoi=85, sink_len=0, first=True)
    assert keep[:4] == [0, 1, 2, 3]
    assert keep[4:16] == list(range(16, 28)) and keep[16:28] == list(range(77, 89))
    assert sink == 28 and keep[28:] == list(range(86, 300))
    keep2, sink2 = O.sink_evict_indices(n_kv=len(keep), boi=40, eoi=105, sink_len=sink, first=False)
    assert keep2[:28] == list(range(28)) and sink2 == 28 + 24


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SS_PKG"])
from seedstory import parallel as P
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
mine = P.slots_for_rank(10, rank, world)
allslots = [None] * world
dist.all_gather_object(allslots, mine)
assert sorted(sum(allslots, [])) == list(range(10)), allslots
assert P.stories_for_rank(5, rank, world) == list(range(rank, 5, world))
feat = torch.full((1, 256, 64), float(rank + 1))
P.broadcast_feature(feat, src=0)
assert float(feat.mean()) == 1.0
f2 = torch.full((4,), float(rank))
P.send_feature(f2, src=1, dst=0)
if rank == 0:
    assert float(f2[0]) == 1.0
k = torch.full((2, 2, 8, 4), float(rank)); v = k.clone() + 10
P.broadcast_kv(k, v, 5, src=0)
assert float(k[:, :, :5].sum()) == 0.0 and float(v[:, :, :5].mean()) == 10.0
if rank == 1:
    assert float(k[:, :, 5:].mean()) == 1.0
assert P.max_over_ranks(1.0 + rank) == float(world)
dist.destroy_process_group()
print("ok", rank)
'''


def test_partitioning_world_size_2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, SS_PKG=PKG, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)], env=env,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def test_bench_story_schedule_matches_baseline_shapes():
    """bench.py's synthetic story (SURVEY §8d schedule): prompt of step i has 115 + 114 i ids, the forced decode
    schedule is 48 caption ids + <img> + 64 image tokens + </img> + EOS = 115 iterations."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    st = bench.Story(7, "cpu")
    assert len(st.ids) == 115 and st.ids[0] == bench.BOS and st.ids[-66:] == bench.IMG_IDS
    f = st.forced()
    assert len(f) == bench.T_GEN == 115 and f[-1] == bench.EOS and f[48:48 + 66] == bench.IMG_IDS
    assert all(3 <= t < 32000 for t in f[:48])
    # context growth of one step (mllm_part's bookkeeping): + caption + the 66 image ids
    grown = st.ids + f[:bench.CAPTION] + bench.IMG_IDS
    assert len(grown) == 115 + 114
    assert tuple(st.image.shape) == (1, 3, 448, 448)


def test_xcd_tile_order_is_a_permutation():
    """Python restatement of ss::xcd_tile (ss_gemm.hip): every (block id) maps to a distinct output tile for any
    grid, group size and ragged last group — the property the GEMM kernels rely on."""
    def xcd_tile(idx, MT, NT, GM):
        total = MT * NT
        xcd, j = idx & 7, idx >> 3
        base, rem = total >> 3, total & 7
        L = xcd * base + min(xcd, rem) + j
        per_group = GM * NT
        gidx = L // per_group
        r = L - gidx * per_group
        m0 = gidx * GM
        gm = min(GM, MT - m0)
        nt = r // gm
        return m0 + r - nt * gm, nt

    for MT, NT in [(1, 1), (3, 1), (1, 7), (5, 5), (64, 8), (32, 5), (13, 11), (256, 4), (7, 80)]:
        for GM in (4, 8, 16):
            seen = {xcd_tile(i, MT, NT, GM) for i in range(MT * NT)}
            assert len(seen) == MT * NT
            assert all(0 <= m < MT and 0 <= n < NT for m, n in seen)


def test_pipeline_stable_conditioning_buffers():
    """The captured UNet forward reads its conditioning from buffers whose ADDRESS must survive across renders: same
    shape -> same storage, refreshed in place (version bump = new-conditioning signal for the K/V cache); new shape ->
    new storage."""
    import torch
    from seedstory.diffusion import StableDiffusionXLPipeline
    pipe = StableDiffusionXLPipeline(vae=None, unet=None, scheduler=None)
    a = torch.arange(12, dtype=torch.float32).reshape(2, 6)
    b1 = pipe._stable("ctx", a)
    ptr, ver = b1.data_ptr(), b1._version
    b2 = pipe._stable("ctx", a + 1)
    assert b2.data_ptr() == ptr and b2._version > ver and torch.equal(b2, a + 1)
    assert b1 is b2 and b1.data_ptr() != a.data_ptr()
    b3 = pipe._stable("ctx", torch.zeros(3, 6))
    assert b3.data_ptr() != ptr and b3.shape == (3, 6)
    other = pipe._stable("pooled", a)
    assert other.data_ptr() not in (ptr, b3.data_ptr())


_RING_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SS_PKG"]); sys.path.insert(0, os.environ["SS_ROOT"])
from seedstory import parallel as P
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)

# ---- (1) the scheduling code with a stub engine: a KV slab grown 5 rows per round FROM THE MIRROR's previous rows ----
class Stub(P.SlotRingBackend):
    def __init__(self):
        self.k = torch.zeros(2, 2, 64, 4); self.ctx = []; self.rendered = []; self.applied = []
    def mllm_round(self, r):
        lo, hi = 5 * r, 5 * r + 5
        prev = float(self.k[:, :, :lo].sum())                 # depends on every earlier round being mirrored correctly
        self.k[:, :, lo:hi] = prev * 0.5 + r + 1 + torch.arange(5).view(1, 1, 5, 1)
        self.ctx.append(r * 7 + 1)
        return [lo, hi, r * 7 + 1], [self.k[:, :, lo:hi].contiguous()]
    def alloc(self, meta):
        return [torch.empty(2, 2, meta[1] - meta[0], 4)]
    def apply(self, r, meta, tensors):
        self.k[:, :, meta[0]:meta[1]] = tensors[0]; self.ctx.append(meta[2]); self.applied.append(r)
    def render(self, r, meta, tensors):
        import time; time.sleep(0.05)                          # slower than an MLLM round: later rounds must not wait for it
        self.rendered.append((r, float(tensors[0].sum())))

n = 7
be = Stub()
mine = P.run_slot_ring(be, n, rank, world)
ref = Stub()                                                    # single-process reference: every round computed locally
for r in range(n):
    ref.mllm_round(r)
assert mine == [r for r in range(n) if r % world == rank], mine
assert torch.equal(be.k, ref.k) and be.ctx == ref.ctx
assert [r for r, _ in be.rendered] == mine and sorted(be.applied + mine) == list(range(n))

# ---- (2) StoryRingBackend (payload / mirror logic of the real bench schedule) over a fake engine ----------------------
import bench as bm
bm.STORY_LEN, bm.WINDOW = 6, 3                                   # evictions at steps 3, 4, 5; new stories after step 5
HID = 8
class FakeEng:
    n_layers, n_heads, hd, hidden = 2, 2, 4, HID
    def __init__(self, n_seq):
        self.K = [torch.zeros(2, 2, 1024, 4) for _ in range(n_seq)]; self.V = [torch.zeros(2, 2, 1024, 4) for _ in range(n_seq)]
        self.cur = 0
    def select(self, b): self.cur = b; return self
    @property
    def k_cache(self): return self.K[self.cur]
    @property
    def v_cache(self): return self.V[self.cur]
def row_val(seed, ids, row, gen):                               # value of KV row `row`: depends on the token sitting there
    tok = ids[row] if row < len(ids) else gen[row - len(ids)]
    return float((seed * 31 + row * 7 + tok * 13) % 1009)
def fake_mllm_part(sts, eng, rin, rout, vit, kv_reuse):
    feats = []
    for b, st in enumerate(sts):
        eng.select(b)
        if st.step == 0:
            st.image_embeds = torch.full((1, 256, HID), float(st.seed % 97))
        S = len(st.ids)
        keep = S - 65 if (st.step > 0 and not st.evicted_last) else 0
        for row in range(keep):                                 # the mirror must already hold the rows being reused
            assert float(eng.k_cache[0, 0, row, 0]) == row_val(st.seed, st.ids, row, []), (rank, b, st.step, row)
        st._S, st._keep = S, keep
    forced = [st.forced() for st in sts]
    for st, f in zip(sts, forced):
        st.last_forced = f
    for b, st in enumerate(sts):
        eng.select(b)
        for row in range(st._keep, st._S + 114):
            val = row_val(st.seed, st.ids, row, forced[b])
            eng.k_cache[:, :, row] = val; eng.v_cache[:, :, row] = val + 0.5
        feats.append(torch.full((256, HID), float((st.seed + st.step) % 89)))
    feat = torch.stack(feats)
    for b, st in enumerate(sts):
        bm.advance_context(st, forced[b], feat[b:b + 1])
    return feat
bm.mllm_part = fake_mllm_part
_orig_init = bm.Story.__init__
def _init(self, seed, device):
    _orig_init(self, seed, device); self.seed = seed
bm.Story.__init__ = _init
def run(world_, rank_, rounds):
    be = P.StoryRingBackend(bm, FakeEng(2), None, None, None, None, 2, "cpu", torch.float32, 30)
    P.run_slot_ring(be, rounds, rank_, world_)
    return be
rounds = 9                                                       # crosses the story boundary (6) and several evictions
be = run(world, rank, rounds)
bm2_ref = P.StoryRingBackend(bm, FakeEng(2), None, None, None, None, 2, "cpu", torch.float32, 30)
for r in range(rounds):
    bm2_ref.mllm_round(r)                                        # single-process reference of the same stream
for b in range(2):
    a, c = be.sts[b], bm2_ref.sts[b]
    assert a.ids == c.ids and a.step == c.step and a.evicted_last == c.evicted_last and torch.equal(a.image_embeds, c.image_embeds)
    S = len(a.ids)
    keep = S - 65 if not a.evicted_last else 0                  # what the NEXT round would reuse must be mirrored
    assert torch.equal(be.eng.K[b][:, :, :keep], bm2_ref.eng.K[b][:, :, :keep]) and torch.equal(be.eng.V[b][:, :, :keep], bm2_ref.eng.V[b][:, :, :keep])
dist.destroy_process_group()
print("ok", rank)
'''


def test_ring_payload_pack_unpack_roundtrip():
    """One flat byte buffer per round (one collective instead of 2 + 2 S): mixed dtypes, empty KV pieces, odd sizes."""
    from seedstory import parallel as P
    ts = [torch.arange(7, dtype=torch.int32).view(1, 7), torch.randn(2, 3, 5).to(torch.bfloat16), torch.empty(2, 2, 0, 4),
          torch.randn(2, 2, 3, 4), torch.arange(3, dtype=torch.int64)]
    flat = P.pack_payload(ts, "cpu")
    like = [torch.empty_like(t) for t in ts]
    back = P.unpack_payload(flat.clone(), like)
    assert all(torch.equal(a, b) and a.dtype == b.dtype and a.shape == b.shape for a, b in zip(ts, back))
    offs, total = P._flat_layout(ts)
    assert all(o % 16 == 0 for o in offs) and total == flat.numel()


def test_slot_ring_world_size_2_gloo(tmp_path):
    """The multi-GPU slot ring (north_star: image slots sharded over the GPUs, MLLM KV cache broadcast): (1) the
    scheduling code of run_slot_ring with a stub engine, (2) StoryRingBackend's payload / mirror bookkeeping (KV row
    ranges, eviction re-prefill, story boundaries) over a fake engine whose MLLM round ASSERTS that the rows it
    reuses were mirrored — two gloo ranks vs a single-process reference of the same stream."""
    script = tmp_path / "ring.py"
    script.write_text(_RING_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SS_PKG=PKG, SS_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("ok") == 2


def test_pipeline_graph_cache_dies_with_its_buffers():
    """a captured UNet forward must never be matched again once a buffer it reads was reallocated (batch
    1 -> 4 -> 1): reallocation bumps a generation counter and drops every cached graph."""
    import torch
    from seedstory.diffusion import StableDiffusionXLPipeline
    pipe = StableDiffusionXLPipeline(vae=None, unet=None, scheduler=None)
    pipe._stable("ctx", torch.zeros(2, 6))
    g0 = pipe._buf_gen
    pipe._graphs = {"some-key": object()}
    pipe._stable("ctx", torch.ones(2, 6))                 # same shape: in place, graphs stay
    assert pipe._buf_gen == g0 and len(pipe._graphs) == 1
    pipe._stable("ctx", torch.zeros(8, 6))                # new batch: new storage
    assert pipe._buf_gen == g0 + 1 and pipe._graphs == {}
    pipe._graphs = {"k": 1}
    pipe._stable("ctx", torch.zeros(2, 6))                # back to the first shape: again a NEW buffer, never the old one
    assert pipe._buf_gen == g0 + 2 and pipe._graphs == {}


def test_vae_decode_dtype_policy():
    """The fp16 SDXL VAE with force_upcast (the reference scripts' dtype, gen_george.py:19,62) decodes in fp32 — what
    diffusers does — and in bf16 only behind the `vae_bf16` opt-in, never in fp16; bf16 stays bf16 (diffusers does not
    up-cast a bf16 VAE); vae_fp32 selects fp32 arithmetic for every module dtype."""
    import torch
    from seedstory import _lib
    from seedstory.diffusion import AutoencoderKL
    cfg = dict(latent_channels=4, out_channels=3, block_out_channels=(32, 64, 64, 64), layers_per_block=2,
               norm_groups=32, scaling_factor=0.13025)
    v = AutoencoderKL(cfg)
    assert v.config.force_upcast is True
    assert v.to(torch.float16).decode_dtype() == torch.float32
    _lib.set_tuning("vae_bf16", 1)
    try:
        assert v.to(torch.float16).decode_dtype() == torch.bfloat16
        assert v.to(torch.float32).decode_dtype() == torch.float32          # the opt-in only concerns the fp16 case
    finally:
        _lib.set_tuning("vae_bf16", 0)
    assert v.to(torch.bfloat16).decode_dtype() == torch.bfloat16
    assert v.to(torch.float32).decode_dtype() == torch.float32
    v2 = AutoencoderKL(dict(cfg, force_upcast=False)).to(torch.float16)
    assert v2.decode_dtype() == torch.float16
    _lib.set_tuning("vae_fp32", 1)
    try:
        assert v.to(torch.float16).decode_dtype() == torch.float32
    finally:
        _lib.set_tuning("vae_fp32", 0)


def test_prepare_inputs_for_generation_matches_reference_fixture():
    """``LlamaForCausalLM.prepare_inputs_for_generation`` (reference :796-852) on every branch — kv_cache_head slicing vs
    last-token slicing, embeds on the first step only, positions from ``cumsum(mask) - 1`` — against outputs of the REAL
    reference function (tests/golden/prepare_inputs.json, generated by oracle/make_golden_prepare_inputs.py)."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from make_golden_prepare_inputs import build_args, encode
    from src.models_clm.modeling_llama_xformer import LlamaConfig, LlamaForCausalLM
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "prepare_inputs.json")))
    m = LlamaForCausalLM(LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=1, vocab_size=50))
    m.eval()
    assert len(fx["cases"]) == 88
    for rec in fx["cases"]:
        c = rec["case"]
        m.use_kv_cache_head, m.kv_cache_head = c["use_head"], c["head"]
        ids, past, mask, emb = build_args(c)
        out = m.prepare_inputs_for_generation(ids, past_key_values=past, attention_mask=mask, inputs_embeds=emb, use_cache=True)
        assert encode(out) == rec["out"], c


def test_tile_table_is_data_round_trip_on_cpu(tmp_path):
    """The GEMM tile table is host-side data (ss_tune_import / export / lookup / clear never touch the device): the shipped
    table loads, keys bucket M to 128, an exported table re-imports identically, and clearing empties it."""
    import json
    from seedstory import _lib, tune
    _lib.lib()
    rows = tune.export_table()
    shipped = json.load(open(os.path.join(PKG, "seedstory", "tune_gfx950.json")))
    assert len(rows) == len(shipped["entries"]) > 100 and shipped["arch"] == "gfx950"
    assert sorted(map(tuple, rows)) == sorted(map(tuple, shipped["entries"]))
    # the UNet's ff1 GEGLU GEMM at batch 8 has an entry; M = 8192 - 100 falls into the same 128-row bucket, M = 4096 does not
    hit = tune.lookup(8192, 10240, 1280, _lib.SS_BF16)
    assert hit is not None and hit == tune.lookup(8192 - 100, 10240, 1280, _lib.SS_BF16)
    assert tune.lookup(12345, 777, 1280, _lib.SS_BF16) is None
    path = str(tmp_path / "table.json")
    n = tune.save_table(path, note="round trip")
    assert n == len(rows)
    _lib.check(_lib.lib().ss_tune_clear(), "ss_tune_clear")
    assert tune.export_table() == [] and tune.lookup(8192, 10240, 1280, _lib.SS_BF16) is None
    assert tune.load_table(path) == n
    assert sorted(map(tuple, tune.export_table())) == sorted(map(tuple, rows))
    assert tune.lookup(8192, 10240, 1280, _lib.SS_BF16) == hit


class _FakeDecodeEngine:
    """Host-side stand-in for LlamaEngine's decode primitives (generate / generate_batch / prefill / select / views) with a
    deterministic toy "model": the next token is a hash of everything fed so far, the image-token processor rule on top
    (generation.py:19-31), hidden row of a fed token = [position, token].  Drives the REAL bookkeeping of
    ``generate_img_block`` / ``generate_batch_img_block`` (seedstory/llama.py) on CPU."""

    def __init__(self, img_ids, n_seq=1, eos=2, max_rows=64):
        import types
        from seedstory.llama import LlamaEngine
        self.img_ids, self.eos_id, self.n_seq, self.max_rows = list(img_ids), eos, n_seq, max_rows
        self.device = torch.device("cpu")
        self.embed = torch.arange(4096, dtype=torch.float32).unsqueeze(1)      # "embedding" of token t is [t]
        self.lm_head = torch.zeros(8, 2)
        self.fed = [[] for _ in range(n_seq)]
        self.cur, self.stop2 = 0, -1
        self._gen = [[] for _ in range(n_seq)]
        self._hid = [torch.zeros(0, 2) for _ in range(n_seq)]
        for name in ("_img_block_plan", "_img_block", "generate_img_block", "generate_batch_img_block", "img_block_enabled"):
            fn = LlamaEngine.__dict__[name]
            setattr(self, name, fn.__func__ if isinstance(fn, staticmethod) else types.MethodType(fn, self))

    # ---- the toy model -------------------------------------------------------------------------------------
    def _next(self, b, last):
        ids = self.img_ids
        if last in ids[:-1]:
            return ids[ids.index(last) + 1]
        h = 17
        for t in self.fed[b]:
            h = (h * 31 + t) % 1000003
        t = 3 + h % 200
        return ids[0] if h % 23 == 0 else t                                      # now and then the model opens an image

    def _feed(self, b, tok):
        self.fed[b].append(tok)
        return torch.tensor([[float(len(self.fed[b]) - 1), float(tok)]])

    # ---- engine surface used by the mixin methods ------------------------------------------------------------
    def set_stop_id(self, t):
        self.stop2 = int(t)

    def select(self, b):
        self.cur = b
        return self

    @property
    def gen_ids(self):
        return torch.tensor(self._gen[self.cur], dtype=torch.int32)

    @property
    def hidden_rows(self):
        return self._hid[self.cur]

    def prefill(self, embeds, pos_ids=None, want_hidden=False):
        rows = [self._feed(self.cur, int(v)) for v in embeds[:, 0].tolist()]
        assert len(rows) <= self.max_rows
        return torch.cat(rows)

    def prefill_batch(self, embeds, want_hidden=False):
        """Stacked continuation of several slots: per slot the same rows ``prefill`` would feed."""
        assert sum(0 if e is None else e.shape[0] for e in embeds) <= self.max_rows * self.n_seq
        out = []
        for b, e in enumerate(embeds):
            out.append(None if e is None else torch.cat([self._feed(b, int(v)) for v in e[:, 0].tolist()]))
        return out

    def _loop(self, b, n_steps, last, forced):
        gen, hid = [], []
        for i in range(n_steps):
            tok = forced[i] if i < len(forced) else self._next(b, last)
            gen.append(tok)
            stop = tok == self.eos_id or tok == self.stop2 or i + 1 >= n_steps
            if stop:
                break
            hid.append(self._feed(b, tok))
            last = tok
        self._gen[b], self._hid[b] = gen, (torch.cat(hid) if hid else torch.zeros(0, 2))
        return len(gen)

    def generate(self, n_steps, last_prompt_id, forced=None):
        return self._loop(self.cur, n_steps, last_prompt_id, list(forced or []))

    def generate_batch(self, n_steps, lasts, forced=None, active=None):
        forced = forced or [[] for _ in range(self.n_seq)]
        return [self._loop(b, n_steps, lasts[b], list(forced[b] or [])) if (active is None or active[b]) else 0
                for b in range(self.n_seq)]


def test_img_block_decode_bookkeeping_on_a_toy_model(monkeypatch):
    """Block decode of the forced image-token run == the token-by-token loop, on every path of the host bookkeeping:
    several images per call, budgets that end before / inside / right after a block, forced prefixes that cover the block,
    EOS, prefill chunking, and lock-step slots that diverge (common-limit feed)."""
    from seedstory import _lib, ops
    _lib.lib()
    monkeypatch.setattr(ops, "gather_rows", lambda table, ids: table[ids.long()])
    monkeypatch.setattr(ops, "gemm", lambda a, w, **kw: a @ w.t())
    img = list(range(3000, 3066))
    cases = 0
    for seed_prompt in ([5, 6, 7], [9], [11, 12, 13, 14, 15, 16]):
        for n_steps in (1, 2, 7, 40, 66, 67, 68, 131, 200, 400):
            for forced in ([], [50, 51, img[0]], [50] + img + [2], [img[0]], [50, 51, img[0]] + img[1:10], [60, 2]):
                ref = _FakeDecodeEngine(img)
                ref.prefill(ref.embed[torch.tensor(seed_prompt)])
                n = ref.generate(n_steps, seed_prompt[-1], forced)
                want_ids, want_hid, want_fed = ref.gen_ids.tolist(), ref.hidden_rows, list(ref.fed[0])
                eng = _FakeDecodeEngine(img)
                eng.prefill(eng.embed[torch.tensor(seed_prompt)])
                ids, hid = eng.generate_img_block(n_steps, seed_prompt[-1], forced)
                assert ids == want_ids and n == len(ids), (seed_prompt, n_steps, forced)
                assert torch.equal(hid, want_hid) and eng.fed[0] == want_fed and eng.stop2 == -1
                cases += 1
    assert cases == 180
    with pytest.raises(_lib.SSError):          # a forced list that contradicts the processor inside the block
        eng = _FakeDecodeEngine(img)
        eng.prefill(eng.embed[torch.tensor([5])])
        eng.generate_img_block(100, 5, [img[0], 77])
    # slots
    for n_steps in (3, 30, 70, 150):
        prompts = [[5, 6], [7], [8, 9, 10], [11]]
        forced = [[40, img[0]], [41, 42, 43, 44, 45, 46, img[0]], [], [47, 2]]
        ref = _FakeDecodeEngine(img, n_seq=4)
        for b in range(4):
            ref.select(b).prefill(ref.embed[torch.tensor(prompts[b])])
        ns = ref.generate_batch(n_steps, [p[-1] for p in prompts], forced)
        want = [(ref.select(b).gen_ids.tolist(), ref.select(b).hidden_rows, list(ref.fed[b])) for b in range(4)]
        eng = _FakeDecodeEngine(img, n_seq=4)
        for b in range(4):
            eng.select(b).prefill(eng.embed[torch.tensor(prompts[b])])
        ids, hids = eng.generate_batch_img_block(n_steps, [p[-1] for p in prompts], forced)
        for b in range(4):
            assert ids[b] == want[b][0] and len(ids[b]) == ns[b], (n_steps, b)
            assert torch.equal(hids[b], want[b][1]) and eng.fed[b] == want[b][2], (n_steps, b)


def test_llm_generate_assembles_block_and_sequential_runs_identically(monkeypatch):
    """``LlamaForCausalLM.generate`` (the mirror's glue around the engine): sequences, per-step hidden_states, the cache
    bookkeeping attributes and the ``max_new_tokens`` guard are the same whether the forced image-token run is decoded as one
    block or token by token — driven on CPU with the toy engine above."""
    from seedstory import _lib, ops
    from src.models_clm.generation import AutoImageTokenGenerationProcessor
    from src.models_clm.modeling_llama_xformer import LlamaConfig, LlamaForCausalLM
    _lib.lib()
    monkeypatch.setattr(ops, "gather_rows", lambda table, ids: table[ids.long()])
    monkeypatch.setattr(ops, "gemm", lambda a, w, **kw: a @ w.t())
    img = list(range(3000, 3066))

    class Eng(_FakeDecodeEngine):
        max_new = 512

        def reset(self):
            self.fed[0] = []

        def lengths(self):
            return (len(self.fed[0]), len(self.fed[0]))

        def set_lengths(self, kv, pos):
            self.fed[0] = self.fed[0][:kv]

        def load_past_key_values(self, past):
            self.fed[0] = list(past)

        def past_key_values(self):
            return tuple(self.fed[0])

    class Tok:
        def encode(self, s, add_special_tokens=False):
            return list(img)

    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SEEDSTORY_IMG_BLOCK", mode)
        m = LlamaForCausalLM(LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=1, vocab_size=50))
        eng = Eng(img)
        monkeypatch.setattr(m, "engine_for_generation", lambda ids, eng=eng: eng)
        proc = [AutoImageTokenGenerationProcessor(tokenizer=Tok())]
        ids = torch.tensor([[1, 40, 41, 42]])
        emb = ids.float().unsqueeze(-1)
        m.use_kv_cache_head, m.kv_cache_head = True, None
        o = m.generate(input_ids=ids, inputs_embeds=emb, logits_processor=proc, max_new_tokens=120, forced_tokens=[60, 61, img[0]])
        assert eng.img_block_enabled() == (mode == "1")
        outs[mode] = (o.sequences.tolist(), torch.cat([h[0].reshape(-1, 2) for h in o.hidden_states[1:]]), m.kv_cache_head,
                      m.past_key_values, o.hidden_states[0][0].shape)
        with pytest.raises(ValueError):
            m.generate(input_ids=ids, inputs_embeds=emb, logits_processor=proc, max_new_tokens=513)
    a, b = outs["1"], outs["0"]
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and a[4] == b[4]
    seq = a[0][0]
    assert seq[:4] == [1, 40, 41, 42] and seq[4:7] == [60, 61, img[0]] and seq[7:7 + 65] == img[1:] and len(seq) == 4 + 120
    assert a[2] == 4 + 119 and len(a[3]) == 4 + 119            # everything but the last generated token is cached


def test_splitk_plan_and_bench_slot_groups():
    """Host-only logic of round 3: (1) the split-K plan of the small-M LLaMA projections (ss_gemm_splitk_workspace_bytes is
    pure host code: S slices of M x N fp32; 0 = the regular tiles) — it splits where the column count leaves most CUs
    idle (N = 4096) and nowhere else; (2) bench.py's decode groups (<= 8 slots per engine since round 4)."""
    from seedstory import _lib
    f = _lib.lib().ss_gemm_splitk_workspace_bytes
    H, I = 4096, 11008
    assert f(264, H, H) == 3 * 264 * H * 4 and f(264, H, I) == 3 * 264 * H * 4            # 192 workgroups of 128x64 -> 3 K ranges
    assert f(460, H, H) == 2 * 460 * H * 4 and f(512, H, I) == 2 * 512 * H * 4            # 256 workgroups -> 2 K ranges
    assert f(264, 3 * H, H) == 0 and f(264, 2 * I, H) == 0                                # >= 256 workgroups of 128x128 already
    assert f(128, H, H) == 0 and f(66, H, I) == 0 and f(513, H, H) == 0 and f(1024, H, I) == 0   # outside 128 < M <= 512
    assert f(264, H, 512) == 0 and f(264, 100, H) == 0                                    # short K / ragged N: regular path
    sys.path.insert(0, ROOT)
    import bench
    assert [bench.slot_groups(n) for n in (1, 2, 3, 4, 6, 8, 9, 16)] == [[1], [2], [3], [4], [6], [8], [5, 4], [8, 8]]
    assert [bench.slot_groups(n, 4) for n in (4, 6, 8)] == [[4], [3, 3], [4, 4]]         # the round-3 grouping (--max-slots 4)
