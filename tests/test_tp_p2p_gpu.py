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


def _rank_loop(rank, tp, conn):
    """prefill, then 10 decode steps stepwise (one nvl_forward per step: replayed as hipGraphs from the third step on) and 12
    more in the fused device loop — the tensor-parallel decode path with the all-reduce launches inside the captured graphs."""
    import importlib
    sys.path.insert(0, str(ROOT))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    pkg = importlib.import_module("nano-vllm-go_amd")
    try:
        cfg = pkg.synth.tiny_config("llama")
        w = pkg.synth.make_weights(cfg, seed=9, scale=0.05)
        m = pkg.HipTransformerModel(cfg, w, precision="bf16", max_seqs=4, max_batch_tokens=256, tp_rank=rank, tp_size=tp)
        conn.send(m.tp_p2p_export())
        m.tp_p2p_attach(conn.recv())
        r = np.random.default_rng(3)
        prompts = [r.integers(0, cfg["vocab_size"], n).tolist() for n in (33, 7, 20)]
        ids = [0, 1, 2]
        out = {}
        for graphs in (0, 1):
            pkg.lib().nvl_set_tuning(21, graphs)
            for i in ids:
                m.seq_reset(i)
            m.reset_stats()
            lg, am = m.forward_batch(ids, prompts, [0, 0, 0])
            logits, pos = [lg], [len(q) for q in prompts]
            for _ in range(10):
                lg, am = m.forward_batch(ids, [[int(t)] for t in am], pos)
                logits.append(lg); pos = [q + 1 for q in pos]
            fused = m.decode_greedy(ids, am, 12)
            out[graphs] = (logits, fused, m.stats()["graph_replays"])
        conn.send(("ok", out))
        m.close()
    except Exception as e:      # noqa: BLE001
        conn.send(("err", repr(e)))


def _rank_timeout(rank, tp, conn):
    """rank 1 shows up late for one forward call: rank 0's all-reduce gives up (bounded spin) and the call fails; after
    nvl_tp_p2p_rearm on both ranks the group works again."""
    import importlib
    import time
    sys.path.insert(0, str(ROOT))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    pkg = importlib.import_module("nano-vllm-go_amd")
    try:
        pkg.lib().nvl_set_tuning(29, 300)                      # wait at most ~0.3 s for a peer
        cfg = pkg.synth.tiny_config("llama")
        w = pkg.synth.make_weights(cfg, seed=9, scale=0.05)
        m = pkg.HipTransformerModel(cfg, w, precision="bf16", max_seqs=2, max_batch_tokens=256, tp_rank=rank, tp_size=tp)
        conn.send(m.tp_p2p_export())
        m.tp_p2p_attach(conn.recv())
        prompt = np.random.default_rng(4).integers(0, cfg["vocab_size"], 12).tolist()
        first = None
        if rank == 1:
            time.sleep(2.5)
        try:
            m.forward_with_cache(prompt, seq_id=1, pos_offset=0)
            first = "ok"
        except pkg.NvlError as e:
            first = "timeout" if "timed out" in str(e) else repr(e)
        conn.send(("phase1", first))
        assert conn.recv() == "rearm"
        m.tp_p2p_rearm()
        conn.send(("rearmed", None))
        assert conn.recv() == "go"
        m.seq_reset(1)
        got = m.forward_with_cache(prompt, seq_id=1, pos_offset=0)
        conn.send(("ok", got))
        m.close()
    except Exception as e:      # noqa: BLE001
        conn.send(("err", repr(e)))


def _spawn(target, tp):
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(tp)]
    procs = [ctx.Process(target=target, args=(r, tp, pipes[r][1])) for r in range(tp)]
    for p in procs:
        p.start()
    handles = []
    for r in range(tp):
        assert pipes[r][0].poll(120), "rank did not export its handle"
        handles.append(pipes[r][0].recv())
    for r in range(tp):
        pipes[r][0].send(handles)
    return procs, pipes


def _reap(procs):
    for p in procs:
        p.join(30)
        if p.is_alive():
            p.terminate()


def test_p2p_decode_runs_inside_replayed_graphs_and_the_fused_loop(gpu):
    """Round-2 finding: graph replay and the deferred RMSNorm were switched off for tp > 1.  The all-reduce's call state now
    lives on the device (tp_p2p.h), so the captured decode passes — all-reduce launches included — replay; the all-reduce
    launch carries the norm.  Graphs on == graphs off bit for bit, ranks identical, replays counted."""
    procs, pipes = _spawn(_rank_loop, 2)
    try:
        res = []
        for r in range(2):
            assert pipes[r][0].poll(240), "rank hung"
            kind, val = pipes[r][0].recv()
            assert kind == "ok", val
            res.append(val)
    finally:
        _reap(procs)
    for rk in range(2):
        eager, graph = res[rk][0], res[rk][1]
        assert eager[2] == 0 and graph[2] >= 8 + 10, (eager[2], graph[2])          # stepwise passes + fused-loop steps were replayed
        for a, b in zip(eager[0], graph[0]):
            assert np.array_equal(a, b)
        assert np.array_equal(eager[1], graph[1])
    for a, b in zip(res[0][1][0], res[1][1][0]):
        assert np.array_equal(a, b)                                                 # every rank holds the same logits
    assert np.array_equal(res[0][1][1], res[1][1][1])


def test_p2p_timeout_fails_the_call_and_rearm_recovers(gpu, oracle):
    procs, pipes = _spawn(_rank_timeout, 2)
    try:
        first = []
        for r in range(2):
            assert pipes[r][0].poll(240), "rank hung"
            kind, val = pipes[r][0].recv()
            assert kind == "phase1", (kind, val)
            first.append(val)
        assert first[0] == "timeout", first           # rank 0 waited ~0.3 s for rank 1 and gave up: an error, not a hang
        for r in range(2):
            pipes[r][0].send("rearm")
        for r in range(2):
            assert pipes[r][0].poll(60) and pipes[r][0].recv()[0] == "rearmed"
        for r in range(2):
            pipes[r][0].send("go")
        outs = []
        for r in range(2):
            assert pipes[r][0].poll(120), "rank hung after the re-arm"
            kind, val = pipes[r][0].recv()
            assert kind == "ok", val
            outs.append(val)
    finally:
        _reap(procs)
    cfg = gpu.synth.tiny_config("llama")
    w = gpu.synth.make_weights(cfg, seed=9, scale=0.05)
    om = oracle.OracleModel(cfg, w)
    prompt = np.random.default_rng(4).integers(0, cfg["vocab_size"], 12).tolist()
    want = om.forward_with_cache(prompt, om.new_cache(), 0)
    assert rel_err(outs[0], want) <= TOL["bf16"] and np.array_equal(outs[0], outs[1])
