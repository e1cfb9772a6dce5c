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


def _bench(*args, timeout=180):
    import json
    import subprocess
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, [json.loads(ln) for ln in lines], p.stderr


def test_bench_started_plainly_launches_its_own_ranks():
    """`python bench.py --gpus N` with no external launcher (what the driver runs) must itself start N rank processes:
    the control flow (spawn before any GPU call, rendezvous, barrier, max over ranks, ONE line from rank 0) on CPU."""
    rc, lines, err = _bench("--gpus", "2", "--launch-check")
    assert rc == 0, err
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["max_rank_plus_1"] == 2.0
    rc, lines, err = _bench("--gpus", "1", "--launch-check")
    assert rc == 0 and lines[0]["n_gpus"] == 1


def test_launcher_propagates_a_failing_rank(pkg, tmp_path):
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\nr = int(os.environ['RANK'])\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "if r == 1: sys.exit(7)\ntime.sleep(30)\n")
    import time
    t0 = time.monotonic()
    rc = pkg.dist.launch_local_ranks([str(script)], 3)
    assert rc == 7 or rc == 15          # rank 1's code; the sleeping ranks were terminated, not waited for
    assert time.monotonic() - t0 < 20
