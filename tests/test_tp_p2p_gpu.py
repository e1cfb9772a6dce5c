"""The hand-written tensor-parallel all-reduce (csrc/tp_p2p.h: one-shot direct peer stores for decode-sized payloads,
reduce-scatter + all-gather for prefill-sized ones, residual add fused) with the hardware this box has: ONE GPU.
Two rank PROCESSES share device 0 and exchange the hipIpcMemHandles of their comm buffers, exactly as two GPUs of a node
would (one process per GPU); what this validates is the protocol — handle exchange, inbox addressing, arrival counters,
parity double-buffering, the fused residual add, rank-order sums — and the sharded arithmetic against the un-sharded CPU
oracle.  Cross-GPU visibility over xGMI and the link rates are UNMEASURED on hardware (no multi-GPU node was available)."""
import multiprocessing as mp
import os
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
TOL = {"f32": 1e-4, "bf16": 1.5e-2}


def _rank(rank, tp, precision, oneshot_rows, conn):
    import importlib
    sys.path.insert(0, str(ROOT))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    pkg = importlib.import_module("nano-vllm-go_amd")
    try:
        pkg.lib().nvl_set_tuning(23, oneshot_rows)
        cfg = pkg.synth.tiny_config("llama")
        w = pkg.synth.make_weights(cfg, seed=9, scale=0.05)
        m = pkg.HipTransformerModel(cfg, w, precision=precision, max_seqs=2, max_batch_tokens=256, tp_rank=rank, tp_size=tp)
        conn.send(m.tp_p2p_export())
        m.tp_p2p_attach(conn.recv())
        r = np.random.default_rng(2)
        prompt = r.integers(0, cfg["vocab_size"], 90).tolist()
        out = [m.forward_with_cache(prompt, seq_id=1, pos_offset=0)]
        pos = len(prompt)
        for t in r.integers(0, cfg["vocab_size"], 5).tolist():
            out.append(m.forward_with_cache([t], seq_id=1, pos_offset=pos, all_logits=False))
            pos += 1
        conn.send(("ok", out))
        m.close()
    except Exception as e:      # noqa: BLE001
        conn.send(("err", repr(e)))


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("oneshot_rows", [64, 0])      # 64: decode steps one-shot, the 90-row prefill two-shot; 0: everything two-shot
def test_p2p_allreduce_two_rank_processes_on_one_gpu(gpu, oracle, precision, oneshot_rows):
    tp = 2
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(tp)]
    procs = [ctx.Process(target=_rank, args=(r, tp, precision, oneshot_rows, pipes[r][1])) for r in range(tp)]
    for p in procs:
        p.start()
    try:
        handles = []
        for r in range(tp):
            assert pipes[r][0].poll(120), "rank did not export its handle"
            handles.append(pipes[r][0].recv())
        assert all(isinstance(h, bytes) and len(h) == 64 for h in handles), handles
        for r in range(tp):
            pipes[r][0].send(handles)
        res = []
        for r in range(tp):
            assert pipes[r][0].poll(180), "rank hung"
            kind, val = pipes[r][0].recv()
            assert kind == "ok", val
            res.append(val)
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.terminate()
    cfg = gpu.synth.tiny_config("llama")
    w = gpu.synth.make_weights(cfg, seed=9, scale=0.05)
    om = oracle.OracleModel(cfg, w)
    r = np.random.default_rng(2)
    prompt = r.integers(0, cfg["vocab_size"], 90).tolist()
    kv = om.new_cache()
    want = [om.forward_with_cache(prompt, kv, 0)]
    pos = len(prompt)
    for t in r.integers(0, cfg["vocab_size"], 5).tolist():
        want.append(om.forward_with_cache([t], kv, pos)[-1:])
        pos += 1
    for step, wv in enumerate(want):
        for rk in range(tp):
            assert rel_err(res[rk][step], wv) <= TOL[precision], (step, rk)
        assert np.array_equal(res[0][step], res[1][step]), step       # every rank holds the same residual stream
