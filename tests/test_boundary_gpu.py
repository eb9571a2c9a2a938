"""GPU tests of the boundary's housekeeping entry points: the per-device context (ss_create / ss_destroy) and the
RCCL wrappers (single-rank communicator: the box has one GPU; the 2-rank exchange logic is covered on CPU with gloo in
test_host_cpu.py and on the multi-GPU node by ``bench.py --partition slots``)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_context_create_info_destroy_keeps_tile_table():
    from seedstory import _lib, comm, tune
    with pytest.raises(_lib.SSError):
        comm.Context(99)
    ctx = comm.Context(0)
    info = ctx.info()
    assert info["device"] == 0 and info["cu_count"] == 256 and info["hbm_bytes"] > 200 * 2 ** 30
    tune.load_default_table()
    n = len(tune.export_table())
    assert n > 0
    ctx.close()
    # the tile table is process-global (ops take no handle): other users of the library keep their tuned tiles
    assert len(tune.export_table()) == n


def test_rccl_single_rank_bcast_and_argument_errors():
    from seedstory import _lib, comm
    uid = comm.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    c = comm.RcclComm(uid, 1, 0)
    t = torch.arange(4096, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    ref = t.clone()
    c.bcast(t, 0)
    torch.cuda.synchronize()
    assert torch.equal(t, ref)
    with pytest.raises(_lib.SSError):
        c.send(t, 0)                                # peer == self
    with pytest.raises(_lib.SSError):
        c.recv(t, 3)                                # peer out of range
    c.close()
