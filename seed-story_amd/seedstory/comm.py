"""Device context and RCCL exchange wrappers over the C ABI (``ss_create`` / ``ss_destroy`` / ``ss_rccl_*``).

The Python host normally exchanges through ``torch.distributed`` (seedstory/parallel.py); these classes are the thin
mirror of the flat C entry points a non-torch host binds, and what the GPU tests drive them through.
"""
import ctypes as C

import torch

from . import _lib

_DT = {torch.float32: _lib.SS_F32, torch.bfloat16: _lib.SS_BF16, torch.float16: _lib.SS_F16}


class Context:
    """Owner of the per-device library state (tuning knobs + GEMM tile table).  ``with Context(0) as ctx: ...``"""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().ss_create(int(device), C.byref(self._h)), "ss_create")

    def info(self):
        out = (C.c_int64 * 3)()
        _lib.check(_lib.lib().ss_context_info(self._h, out), "ss_context_info")
        return {"device": int(out[0]), "cu_count": int(out[1]), "hbm_bytes": int(out[2])}

    def close(self):
        if self._h:
            _lib.lib().ss_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def rccl_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    _lib.check(_lib.lib().ss_rccl_unique_id(buf), "ss_rccl_unique_id")
    return buf.raw


class RcclComm:
    """One rank of an RCCL communicator created from a 128-byte unique id (rank 0 makes it, the host ships it)."""

    def __init__(self, unique_id: bytes, nranks: int, rank: int):
        assert len(unique_id) == 128
        self._h = C.c_void_p()
        self.rank, self.nranks = rank, nranks
        _lib.check(_lib.lib().ss_rccl_init(unique_id, nranks, rank, C.byref(self._h)), "ss_rccl_init")

    @staticmethod
    def _args(t: torch.Tensor):
        assert t.is_cuda and t.is_contiguous()
        return t.data_ptr(), t.numel(), _DT[t.dtype], torch.cuda.current_stream(t.device).cuda_stream

    def send(self, t, peer):
        p, n, d, s = self._args(t)
        _lib.check(_lib.lib().ss_rccl_send(self._h, p, n, d, peer, s), "ss_rccl_send")

    def recv(self, t, peer):
        p, n, d, s = self._args(t)
        _lib.check(_lib.lib().ss_rccl_recv(self._h, p, n, d, peer, s), "ss_rccl_recv")

    def bcast(self, t, root):
        p, n, d, s = self._args(t)
        _lib.check(_lib.lib().ss_rccl_bcast(self._h, p, n, d, root, s), "ss_rccl_bcast")

    def close(self):
        if self._h:
            _lib.lib().ss_rccl_destroy(self._h)
            self._h = C.c_void_p()
