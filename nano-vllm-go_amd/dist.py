"""Multi-GPU plumbing: the rank launcher, rendezvous and timing reductions bench.py runs on, and DATA PARALLEL OVER
SEQUENCES, the way the path shards for models that fit one GPU (tensor parallel lives in the library: nvl_tp_*).

Sequences in a ModelRunner.Run batch are independent (the reference loops them serially with
per-sequence caches, nanovllm/tensor_model_runner.go:58), so each rank (one process per GPU) owns the
sequences with seq_id % world == rank — sticky, because the KV slab lives on the owning GPU — holds a
full replica of the weights, and no collective touches the data path.  torch.distributed (backend
"nccl" = RCCL on ROCm, "gloo" in the CPU tests) is used only for the rendezvous, the barrier around
the timed region, the max-over-ranks of the timings and the gather of the sampled token ids.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_local_ranks(argv, n: int, timeout: float | None = None) -> int:
    """Start `n` rank processes of `argv` (a python script + its arguments) on this node, one per GPU, the way
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` would: fresh child processes with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set.  MUST be called before the calling process
    touches the GPU (no HIP call, no torch.cuda.is_available()): the parent only waits, it never becomes a rank
    and never execs.  Children inherit stdout/stderr (rank 0 prints the result line).  Returns the largest
    exit code; if a rank fails the others are terminated."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this pool
        procs.append(subprocess.Popen([sys.executable, *argv], env=env))
    rc = 0
    try:
        import time
        t0 = time.monotonic()
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                rc = max(rc, abs(code))
                if code != 0:                    # one rank died: the others would wait in a collective forever
                    for q in live:
                        q.terminate()
            if timeout is not None and time.monotonic() - t0 > timeout:
                for q in live:
                    q.terminate()
                rc = max(rc, 124)
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    return rc


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
