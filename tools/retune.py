"""Re-measure the GEMM entries of the shipped tile table with the current candidate list (round 6: + the ping-pong tiles).

    python tools/retune.py [--min-m 256] [--out gpurun_out/tune_gfx950.json]

Every non-conv entry of seedstory/tune_gfx950.json is re-tuned through the C ABI's ss_gemm_tune (sustained mode, rotating
weights); conv entries are kept.  Prints old -> new per shape; writes the new table + a log.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch  # noqa: E402
from seedstory import _lib, tune  # noqa: E402

GEGLU = {(10240, 1280), (5120, 640)}     # the UNet's ff1 products (N, K): value / gate pairs folded in the epilogue


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-m", type=int, default=256)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tune_gfx950.json"))
    ap.add_argument("--conv", action="store_true", help="also re-tune the conv entries")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    old = json.load(open(tune.DEFAULT_TABLE))
    rows = old["entries"]
    t0 = time.time()
    log = []
    for r in rows:
        dt, M, N, K, cin, su, cH, cW, cfg, swz = r
        if cin:
            if not a.conv:
                continue
            stride, up = su // 2, su % 2
            Hin, Win = (2 * cH, 2 * cW) if up else (cH, cW)
            Ho, Wo = (Hin + 2 - 3) // stride + 1, (Win + 2 - 3) // stride + 1
            B = M // (Ho * Wo)
            nbytes = lib.ss_conv3x3_tune_workspace_bytes(B, cH, cW, cin, N, stride, up, dt)
            ws = tune._workspace(nbytes, dev)
            us = C.c_float()
            _lib.check(lib.ss_conv3x3_tune(B, cH, cW, cin, N, stride, up, dt, ws.data_ptr(), ws.numel(),
                                           torch.cuda.current_stream().cuda_stream, C.byref(us)), "ss_conv3x3_tune")
            new = tune.lookup(M, N, K, dt, (cin, cH, cW, stride, up))
        else:
            if M < a.min_m:
                continue
            epi = _lib.EPI_GEGLU_PAIR if (N, K) in GEGLU else 0
            nbytes = lib.ss_gemm_tune_workspace_bytes(M, N, K, dt)
            ws = tune._workspace(nbytes, dev)
            us = C.c_float()
            _lib.check(lib.ss_gemm_tune(M, N, K, epi, dt, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream,
                                        C.byref(us)), "ss_gemm_tune")
            new = tune.lookup(M, N, K, dt)
        tf = 2.0 * M * N * K / (us.value * 1e-6) / 1e12
        log.append({"shape": [M, N, K], "conv": [cin, su, cH, cW], "old": [cfg, swz], "new": list(new), "us": round(us.value, 2), "tflops": round(tf, 1)})
        print("%-28s conv=%-4d old %3d/%d -> new %3d/%d  %8.1f us %7.1f TF" % ((M, N, K), cin, cfg, swz, new[0], new[1], us.value, tf), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    n = tune.save_table(a.out, note=old["note"] + "; round 6: GEMM entries re-measured by tools/retune.py with the ping-pong 256x256 tiles (cfg 54 / 55) among the candidates")
    json.dump(log, open(a.out.replace(".json", "_retune_log.json"), "w"), indent=0)
    changed = sum(1 for e in log if e["old"][0] != e["new"][0])
    print("re-tuned %d entries (%d changed tile), table %d entries, %.0f s" % (len(log), changed, n, time.time() - t0))


if __name__ == "__main__":
    main()
