"""Multi-GPU partitioning of the story path — one process per GPU,
``torch.distributed`` over RCCL (backend "nccl" on ROCm; "gloo" in the CPU tests).

Two levels, no all-reduce anywhere on the data path:

* story level (``stories_for_rank``): the outer loop over stories (gen_george.py:152) has no
  cross-iteration state -> stories are dealt round-robin to ranks, zero communication;
* slot level (``slot_owner`` / ``broadcast_feature`` / ``send_feature``): inside one story, MLLM step
  t+1 needs only ``img_gen_feat_t`` (gen_george.py:224), never the rendered pixels, so image slot t
  is rendered by rank ``t mod N`` and the only message is the 2 MiB regressed feature [1,256,4096]
  (plus, when replicas mirror the MLLM context, the KV cache slab 0.5 MiB x S).
"""
import torch
import torch.distributed as dist


def stories_for_rank(n_stories, rank, world):
    return list(range(rank, n_stories, world))


def slot_owner(slot, world, mllm_rank=0):
    """Rank that renders image slot `slot` (round-robin, starting after the MLLM rank so that the
    MLLM recurrence and the first render overlap when world > 1)."""
    return (mllm_rank + 1 + slot) % world if world > 1 else 0


def slots_for_rank(n_slots, rank, world, mllm_rank=0):
    return [s for s in range(n_slots) if slot_owner(s, world, mllm_rank) == rank]


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


def broadcast_kv(engine_k, engine_v, length, src=0, group=None):
    """Broadcast the live part of the KV slab [L, H, cap, hd] (first `length` slots)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        k = engine_k[:, :, :length].contiguous()
        v = engine_v[:, :, :length].contiguous()
        dist.broadcast(k, src=src, group=group)
        dist.broadcast(v, src=src, group=group)
        if dist.get_rank(group) != src:
            engine_k[:, :, :length].copy_(k)
            engine_v[:, :, :length].copy_(v)
    return engine_k, engine_v


def max_over_ranks(seconds, device=None, group=None):
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
