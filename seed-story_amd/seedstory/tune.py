"""Tile-configuration table of the MFMA GEMM / implicit-GEMM conv kernels (``ss_tune_*`` in the C ABI).

``ss_gemm`` / ``ss_conv3x3`` never time or allocate: they look the shape up in a per-process table and
fall back to a closed-form rule.  This module owns the table's life cycle on the host side:

* ``load_default_table()`` — imports ``tune_gfx950.json`` (measured on MI355X, shipped with the package;
  ``SEEDSTORY_TUNE_TABLE`` overrides the path) when the library is first loaded, so the tile choice of
  every shape on the path is deterministic data, not a side effect of the first call;
* ``ensure_gemm`` / ``ensure_conv`` — called by ``seedstory.ops`` before a launch: a shape that has no
  entry is tuned ONCE, explicitly (``ss_gemm_tune``: synchronising, in a workspace this module owns and
  can release), unless tuning is disabled (``set_tuning("gemm_autotune", 0)``) or the stream is being
  captured;
* ``save_table`` / ``load_table`` — JSON export / import of the table.
"""
import ctypes as C
import json
import os

import torch

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_TABLE = os.path.join(_HERE, "tune_gfx950.json")
_seen = set()
_ws = None
_log = []          # (kind, shape, cfg, swz, best_us) of every shape tuned in this process


def _workspace(nbytes, device):
    global _ws
    if _ws is None or _ws.numel() < nbytes or _ws.device != device:
        _ws = None
        _ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return _ws


def release_workspace():
    """Free the tuning workspace (a few hundred MB) once the shapes of interest are tuned."""
    global _ws
    _ws = None


def _enabled():
    return _lib.get_tuning("gemm_autotune", 1) != 0 and not _lib.get_tuning("gemm_cfg", 0) \
        and not torch.cuda.is_current_stream_capturing()


def lookup(M, N, K, dtype_code, conv=(0, 0, 0, 0, 0)):
    out = (C.c_int32 * 2)()
    rc = _lib.lib().ss_tune_lookup(M, N, K, conv[0], conv[1], conv[2], conv[3], conv[4], dtype_code, out)
    return None if rc != 0 else (int(out[0]), int(out[1]))


def ensure_gemm(M, N, K, dtype_code, epi=0, device=None):
    if M <= 128 or dtype_code == _lib.SS_F32 or K % 8:
        return
    key = ("g", dtype_code, (M + 127) // 128 * 128, N, K)
    if key in _seen:
        return
    if lookup(M, N, K, dtype_code) is None:
        if not _enabled():
            return                      # not recorded: retried when tuning becomes possible
        lib = _lib.lib()
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        nbytes = lib.ss_gemm_tune_workspace_bytes(M, N, K, dtype_code)
        ws = _workspace(nbytes, device)
        us = C.c_float()
        _lib.check(lib.ss_gemm_tune(M, N, K, int(epi) & (_lib.EPI_GELU | _lib.EPI_GEGLU_PAIR), dtype_code, ws.data_ptr(),
                                    ws.numel(), torch.cuda.current_stream().cuda_stream, C.byref(us)), "ss_gemm_tune")
        cfg = lookup(M, N, K, dtype_code)
        _log.append(("gemm", (M, N, K), cfg[0], cfg[1], float(us.value)))
    _seen.add(key)


def ensure_conv(B, H, W, Cin, Cout, stride, up, dtype_code, device=None):
    if dtype_code == _lib.SS_F32 or Cin % 8:
        return
    Hin, Win = (2 * H, 2 * W) if up else (H, W)
    Ho, Wo = (Hin + 2 - 3) // stride + 1, (Win + 2 - 3) // stride + 1
    M = B * Ho * Wo
    if M <= 128:
        return
    key = ("c", dtype_code, B, H, W, Cin, Cout, stride, int(up))
    if key in _seen:
        return
    conv = (Cin, H, W, stride, int(up))
    if lookup(M, Cout, 9 * Cin, dtype_code, conv) is None:
        if not _enabled():
            return
        lib = _lib.lib()
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        nbytes = lib.ss_conv3x3_tune_workspace_bytes(B, H, W, Cin, Cout, stride, int(up), dtype_code)
        ws = _workspace(nbytes, device)
        us = C.c_float()
        _lib.check(lib.ss_conv3x3_tune(B, H, W, Cin, Cout, stride, int(up), dtype_code, ws.data_ptr(), ws.numel(),
                                       torch.cuda.current_stream().cuda_stream, C.byref(us)), "ss_conv3x3_tune")
        cfg = lookup(M, Cout, 9 * Cin, dtype_code, conv)
        _log.append(("conv", (B, H, W, Cin, Cout, stride, int(up)), cfg[0], cfg[1], float(us.value)))
    _seen.add(key)


def tuned_log():
    return list(_log)


def export_table():
    lib = _lib.lib()
    n = lib.ss_tune_export(None, 0)
    buf = (C.c_int32 * (10 * max(n, 1)))()
    n = min(n, lib.ss_tune_export(buf, n))
    return [[int(buf[i * 10 + j]) for j in range(10)] for i in range(n)]


def import_table(rows):
    rows = [r for r in rows if len(r) == 10]
    if not rows:
        return 0
    buf = (C.c_int32 * (10 * len(rows)))(*[int(v) for r in rows for v in r])
    _lib.check(_lib.lib().ss_tune_import(buf, len(rows)), "ss_tune_import")
    return len(rows)


def save_table(path=DEFAULT_TABLE, note=""):
    rows = export_table()
    with open(path, "w") as f:
        json.dump({"arch": "gfx950", "note": note,
                   "record": ["dtype", "M_bucket", "N", "K", "conv_Cin", "2*stride+up", "conv_H", "conv_W", "cfg", "xcd_group"],
                   "entries": rows}, f, indent=0)
    return len(rows)


def load_table(path):
    with open(path) as f:
        d = json.load(f)
    return import_table(d.get("entries", []))


def load_default_table():
    path = os.environ.get("SEEDSTORY_TUNE_TABLE", DEFAULT_TABLE)
    if path and os.path.exists(path):
        try:
            return load_table(path)
        except Exception as ex:      # a damaged table must not take the library down: the closed-form rule remains
            import sys
            print("seedstory: could not load tune table %s (%r)" % (path, ex), file=sys.stderr)
    return 0
