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


def test_slot_ring_and_replicas_under_rccl_world_size_1(tmp_path):
    """The first multi-GPU run must not be the first RCCL run: ``bench.py --partition slots`` and the replica mode on THIS
    box's one GPU with a real ``nccl`` (= RCCL) process group of world size 1 — communicator initialisation, the
    header + flat-payload broadcasts of the slot ring, barrier and max-over-ranks all go through RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561",
               SS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra in (["--partition", "slots"], []):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
               "--stories-per-gpu", "2", "--diffusion-steps", "2", "--story-len", "3", "--no-cpu-baseline", "--no-batch1"] + extra
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 1 and line["value"] > 0
        if extra:
            assert line["backend"] == "nccl" and line["config"]["partition"] == "slots"
