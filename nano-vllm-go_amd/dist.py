"""Multi-GPU plumbing for the one way this path shards in round 1: DATA PARALLEL OVER SEQUENCES.

Sequences in a ModelRunner.Run batch are independent (the reference loops them serially with
per-sequence caches, nanovllm/tensor_model_runner.go:58), so each rank (one process per GPU) owns the
sequences with seq_id % world == rank — sticky, because the KV slab lives on the owning GPU — holds a
full replica of the weights, and no collective touches the data path.  torch.distributed (backend
"nccl" = RCCL on ROCm, "gloo" in the CPU tests) is used only for the rendezvous, the barrier around
the timed region, the max-over-ranks of the timings and the gather of the sampled token ids.
"""
from __future__ import annotations

import os


def owner(seq_id: int, world: int) -> int:
    return int(seq_id) % int(world)


def shard(seq_ids, rank: int, world: int):
    """indices (into seq_ids) of the sequences this rank owns."""
    return [i for i, s in enumerate(seq_ids) if owner(s, world) == rank]


def init(backend: str, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if dist.is_initialized():
        return dist
    if device is not None:
        dist.init_process_group(backend, device_id=device)
    else:
        dist.init_process_group(backend)
    return dist


def max_over_ranks(values, device="cpu"):
    """elementwise max of a list of floats over all ranks (1 rank: identity)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t.cpu()]


def gather_tokens(seq_ids, local_idx, local_tokens, device="cpu"):
    """Every rank gets the next-token list for ALL sequences of the batch, in batch order
    (what ModelRunner.Run returns to the engine): all_reduce(sum) of a vector that each rank fills
    only at its own positions."""
    import torch
    import torch.distributed as dist
    out = torch.zeros(len(seq_ids), dtype=torch.int64, device=device)
    for i, tok in zip(local_idx, local_tokens):
        out[i] = int(tok)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return [int(x) for x in out.cpu()]
