"""Does overlapping INDEPENDENT UNet forwards buy throughput?  One captured forward (batch UB) replayed alone vs two captured
forwards replayed concurrently on two streams (same weights, separate activations): per-forward time = T / number of forwards.
    python tools/unet_concurrent.py   -> gpurun_out/unet_concurrent.json"""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch
from seedstory import _lib
from seedstory.diffusion import UNet2DConditionModel, timestep_embedding
DEV, dt = "cuda:0", torch.bfloat16
UB = int(os.environ.get("SS_UNET_BATCH", "8"))
unet = UNet2DConditionModel().to(DEV, dt).init_synthetic(1)
ctx = torch.randn(UB, 64, 2048, device=DEV, dtype=dt)
cond = {"text_embeds": torch.randn(UB, 1280, device=DEV, dtype=dt), "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * UB, dtype=torch.float32)}
temb = timestep_embedding(torch.full((UB,), 500.0), unet.cfg["block_out_channels"][0]).to(device=DEV, dtype=dt)
xs = [torch.randn(UB, 4, 128, 128, device=DEV, dtype=dt) for _ in range(2)]
unet(xs[0], None, ctx, added_cond_kwargs=cond, return_dict=False, temb_in=temb)
torch.cuda.synchronize()
graphs = []
for i in range(2):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        eps = unet(xs[i], None, ctx, added_cond_kwargs=cond, return_dict=False, temb_in=temb)[0]
    graphs.append((g, eps))
torch.cuda.synchronize()
s = [torch.cuda.Stream(), torch.cuda.Stream()]


def timed(fn, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(e0)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


def one(e0):
    graphs[0][0].replay()


def two_seq(e0):
    graphs[0][0].replay()
    graphs[1][0].replay()


def two_conc(e0):
    cur = torch.cuda.current_stream()
    for i in range(2):
        s[i].wait_stream(cur)
        with torch.cuda.stream(s[i]):
            graphs[i][0].replay()
    for i in range(2):
        cur.wait_stream(s[i])


res = {}
for rnd in range(3):
    for name, fn, k in (("one", one, 1), ("two_sequential", two_seq, 2), ("two_concurrent", two_conc, 2)):
        res.setdefault(name, []).append(timed(fn) / k)
out = {k: round(statistics.median(v), 3) for k, v in res.items()}
print(out)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "unet_concurrent.json"), "w"))
