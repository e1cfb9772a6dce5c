"""The N>1 path on CPU: world_size-2 gloo.  Data parallel over sequences (nano-vllm-go_amd/dist.py): each rank
runs ModelRunner.Run semantics on the sequences it owns (here with the CPU oracle standing in for the
device model — the sharding, barrier, max-reduce and token gather are what is under test) and every rank
ends up with the same next-token list as a single-process run."""
import importlib
import os
import sys
from pathlib import Path

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    pkg = importlib.import_module("nano-vllm-go_amd")
    from oracle import purego_oracle as O
    dist = pkg.dist.init("gloo")
    cfg = pkg.synth.tiny_config("llama")
    om = O.OracleModel(cfg, pkg.synth.make_weights(cfg, seed=5, scale=0.05))
    r = np.random.default_rng(0)
    seq_ids = [11, 4, 7, 20, 9]
    prompts = [r.integers(0, cfg["vocab_size"], n).tolist() for n in (5, 3, 8, 2, 6)]
    mine = pkg.dist.shard(seq_ids, rank, world)
    dist.barrier()
    toks = [O.argmax(om.forward_with_cache(prompts[i], om.new_cache(), 0)[-1]) for i in mine]
    dist.barrier()
    t = pkg.dist.max_over_ranks([float(rank + 1), 0.5])
    allt = pkg.dist.gather_tokens(seq_ids, mine, toks)
    q.put((rank, mine, allt, t))
    dist.destroy_process_group()


def test_dp_over_sequences_world2_gloo(pkg, oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    cfg = pkg.synth.tiny_config("llama")
    om = oracle.OracleModel(cfg, pkg.synth.make_weights(cfg, seed=5, scale=0.05))
    r = np.random.default_rng(0)
    prompts = [r.integers(0, cfg["vocab_size"], n).tolist() for n in (5, 3, 8, 2, 6)]
    want = [oracle.argmax(om.forward_with_cache(p, om.new_cache(), 0)[-1]) for p in prompts]
    (r0, mine0, all0, t0), (r1, mine1, all1, t1) = res
    assert sorted(mine0 + mine1) == [0, 1, 2, 3, 4] and not set(mine0) & set(mine1)
    assert all0 == want and all1 == want           # every rank sees the whole batch's tokens, in batch order
    assert t0 == t1 == [2.0, 0.5]                  # max over ranks
