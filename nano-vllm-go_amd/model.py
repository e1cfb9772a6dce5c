"""Host-side mirror of tensor.TransformerModel (purego/tensor/generic_model.go:4-19) over the C ABI:
same method names and argument meaning as the reference (ForwardWithCache, GetLogitsForLastToken),
device-resident weights and KV slabs underneath."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _cfg_struct(cfg: dict) -> L.ModelConfigC:
    c = L.ModelConfigC()
    for k in ("vocab_size", "hidden", "num_layers", "num_heads", "num_kv_heads", "head_dim", "ffn_dim",
              "max_seq_len", "num_experts", "num_experts_per_tok"):
        setattr(c, k, int(cfg.get(k, 0)))
    c.attention_type = L.ATTN[cfg["attention_type"]]
    c.norm_type = L.NORM[cfg["norm_type"]]
    c.position_type = L.POS[cfg["position_type"]]
    c.activation_type = L.ACT[cfg["activation_type"]]
    c.block_style = L.BLOCK[cfg["block_style"]]
    c.rope_base = float(cfg.get("rope_base", 10000.0))
    c.norm_eps = float(cfg.get("norm_eps", 1e-5))
    c.tied_embedding = int(bool(cfg.get("tied_embedding", False)))
    c.use_moe = int(bool(cfg.get("use_moe", False)))
    for k in ("embedding_multiplier", "attention_multiplier", "residual_multiplier", "logits_scaling"):
        setattr(c, k, float(cfg.get(k, 0.0)))
    for k in ("mamba_expand", "mamba_state_size", "mamba_num_heads", "mamba_head_dim", "mamba_n_groups", "mamba_conv_kernel"):
        setattr(c, k, int(cfg.get(k, 0)))
    for i, t in enumerate(cfg.get("hybrid_layers") or []):          # config.go:113 HybridLayers
        if t in ("mamba", "mamba2"):
            c.mamba_layer_mask[i >> 6] |= 1 << (i & 63)
    return c


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HipTransformerModel:
    """tensor.TransformerModel on one MI355X.  `tensors` is {(slot, layer): array} in the reference's
    post-load layout (2-D weights [in, out]); arrays may be numpy fp32 or torch tensors (fp32/bf16,
    host or device) — torch device tensors are handed over by pointer with no host copy."""

    def __init__(self, cfg: dict, tensors: dict | None, *, device: int = 0, precision: str = "bf16",
                 max_seqs: int = 8, max_batch_tokens: int | None = None, tp_rank: int = 0, tp_size: int = 1,
                 tp_force_single: bool = False, kv_num_blocks: int = 0, kv_block_size: int = 0):
        """tensors=None defers finalize(): upload() each tensor, then call finalize().
        tp_size > 1 makes this the tp_rank-th tensor-parallel shard: pass the FULL tensors, the library keeps
        its slice; join the shards with tp_init (RCCL, one process per GPU) or attach_local_group (tests)."""
        self.cfg = dict(cfg)
        self.lib = L.lib()
        self.h = C.c_void_p()
        opts = L.RuntimeOptsC(device=device, precision=L.PRECISION[precision], max_seqs=max_seqs,
                              max_batch_tokens=max_batch_tokens or cfg["max_seq_len"], tp_rank=tp_rank, tp_size=tp_size)
        if tp_force_single:
            opts.tp_force_single = 1
        # kv_num_blocks > 0: paged KV — the host's block manager (nanovllm/block_manager.go) owns the cache
        opts.kv_num_blocks, opts.kv_block_size = int(kv_num_blocks), int(kv_block_size)
        self.kv_block_size = (kv_block_size or 256) if kv_num_blocks else 0
        self._c = _cfg_struct(cfg)
        L.check(self.lib.nvl_create(C.byref(self._c), C.byref(opts), C.byref(self.h)))
        self.V = cfg["vocab_size"]
        self.H = cfg["hidden"]
        if tensors is not None:
            for (slot, layer), arr in tensors.items():
                self.upload(slot, layer, arr)
            self.finalize()

    @classmethod
    def from_pretrained(cls, path: str, **kw):
        """tensor.LoadModelFromDirectory (generic_loader.go:1016-1040) without the fp32 host model: config.json (or
        model_info.json) through nvl_load_config_json, weights through nvl_load_safetensors (mmap -> device)."""
        import os
        c = L.ModelConfigC()
        for name in ("config.json", "model_info.json"):
            cp = os.path.join(path, name)
            if os.path.exists(cp):
                L.check(L.lib().nvl_load_config_json(cp.encode(), C.byref(c)))
                break
        else:
            raise FileNotFoundError(f"no config.json / model_info.json in {path}")
        inv = {k: {v: n for n, v in tbl.items()} for k, tbl in
               (("attention_type", L.ATTN), ("norm_type", L.NORM), ("position_type", L.POS),
                ("activation_type", L.ACT), ("block_style", L.BLOCK))}
        cfg = {}
        for name, _ in L.ModelConfigC._fields_:
            v = getattr(c, name)
            if name == "mamba_layer_mask":
                if v[0] | v[1]:
                    cfg["hybrid_layers"] = ["mamba" if (v[i >> 6] >> (i & 63)) & 1 else "attention" for i in range(c.num_layers)]
                continue
            cfg[name] = inv[name][v] if name in inv else (bool(v) if name in ("tied_embedding", "use_moe") else v)
        m = cls(cfg, None, **kw)
        L.check(m.lib.nvl_load_safetensors(m.h, path.encode()), m.h)
        m.finalize()
        return m

    def finalize(self):
        L.check(self.lib.nvl_finalize(self.h), self.h)

    # -- tensor parallel group ---------------------------------------------------------------------
    @staticmethod
    def tp_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        L.check(L.lib().nvl_tp_get_unique_id(buf, 128))
        return buf.raw

    def tp_init(self, unique_id: bytes):
        L.check(self.lib.nvl_tp_init(self.h, C.c_char_p(unique_id), len(unique_id)), self.h)

    def tp_p2p_export(self) -> bytes:
        """IPC handle (64 bytes) of this rank's all-reduce comm buffer (nvl_tp_p2p_export); gather them in rank order."""
        buf = C.create_string_buffer(64)
        L.check(self.lib.nvl_tp_p2p_export(self.h, buf, 64), self.h)
        return buf.raw

    def tp_p2p_attach(self, handles):
        """handles: the tp_size exported handles in rank order (nvl_tp_p2p_attach): switches the all-reduce to direct peer stores."""
        blob = b"".join(handles)
        L.check(self.lib.nvl_tp_p2p_attach(self.h, C.c_char_p(blob), 64), self.h)

    def tp_p2p_rearm(self):
        """nvl_tp_p2p_rearm: after a timed-out all-reduce, on every rank, while no rank is inside a forward call."""
        L.check(self.lib.nvl_tp_p2p_rearm(self.h), self.h)

    @staticmethod
    def attach_local_group(models):
        arr = (C.c_void_p * len(models))(*[m.h.value for m in models])
        L.check(L.lib().nvl_tp_attach_local(arr, len(models)))

    # -- weights -----------------------------------------------------------------------------
    def upload(self, slot: str, layer: int, arr, layout: int | None = None):
        if layout is None:
            layout = L.LAYOUT_OUT_IN if slot in L.OUT_IN_SLOTS else L.LAYOUT_IN_OUT
        ptr, dtype, shape, keep = _as_pointer(arr)
        if slot in L.ONE_D:
            rows, cols = int(np.prod(shape)), 1
        elif slot in ("moe_in", "moe_out"):
            rows, cols = shape[0] * shape[1], shape[2]
        else:
            rows, cols = shape
        L.check(self.lib.nvl_upload_tensor(self.h, L.SLOT_ID[slot], int(layer), ptr, dtype, rows, cols, layout), self.h)
        del keep

    # -- sequences (replace kvCaches map[int64]*KVCache, tensor_model_runner.go:13) ----------------
    def seq_reset(self, seq_id: int):
        L.check(self.lib.nvl_seq_reset(self.h, seq_id), self.h)

    def seq_close(self, seq_id: int):
        L.check(self.lib.nvl_seq_close(self.h, seq_id), self.h)

    def seq_close_all(self):
        L.check(self.lib.nvl_seq_close_all(self.h), self.h)

    def seq_len(self, seq_id: int) -> int:
        return self.lib.nvl_seq_len(self.h, seq_id)

    # -- forward --------------------------------------------------------------------------------
    def forward_batch(self, seq_ids, token_lists, pos_offsets, *, all_logits=False, want_logits=True):
        """nvl_forward: returns (logits [rows, V] or None, argmax [n_seqs])."""
        n = len(seq_ids)
        ids = np.asarray(seq_ids, np.int64)
        lens = np.asarray([len(t) for t in token_lists], np.int32)
        toks = np.concatenate([np.asarray(t, np.int32) for t in token_lists]).astype(np.int32)
        pos = np.asarray(pos_offsets, np.int32)
        rows = int(lens.sum()) if all_logits else n
        logits = np.empty((rows, self.V), np.float32) if want_logits else None
        am = np.empty(n, np.int32)
        L.check(self.lib.nvl_forward(self.h, n, _ptr(ids), _ptr(toks), _ptr(lens), _ptr(pos),
                                     L.FWD_ALL_LOGITS if all_logits else 0, _ptr(logits), _ptr(am)), self.h)
        return logits, am

    def forward_with_cache(self, token_ids, seq_id: int, pos_offset: int, *, all_logits=True):
        """TransformerModel.ForwardWithCache (generic_model.go:276): logits [S, V] for one sequence.
        The *KVCache argument of the reference is the sequence's device slot, named by seq_id;
        pos_offset == 0 with a fresh id (or after seq_reset) is the reference's kvCache == nil."""
        if self.seq_len(seq_id) < 0:
            L.check(self.lib.nvl_seq_open(self.h, seq_id), self.h)
        logits, _ = self.forward_batch([seq_id], [list(token_ids)], [pos_offset], all_logits=all_logits)
        return logits

    @staticmethod
    def get_logits_for_last_token(logits):
        """GetLogitsForLastToken (generic_model.go:595-604)."""
        return logits[-1].copy()

    def greedy(self, prompt, max_tokens: int, seq_id: int = 0):
        """cmd/ask/main.go:287-360 generateResponse with argmax on the device (no EOS stop)."""
        self.seq_reset(seq_id)
        all_tokens = list(prompt)
        _, am = self.forward_batch([seq_id], [all_tokens], [0], want_logits=False)
        out = [int(am[0])]
        all_tokens.append(out[-1])
        for _ in range(max_tokens - 1):
            _, am = self.forward_batch([seq_id], [[all_tokens[-1]]], [len(all_tokens) - 1], want_logits=False)
            out.append(int(am[0]))
            all_tokens.append(out[-1])
        return out

    def forward_paged(self, token_lists, pos_offsets, block_tables, *, all_logits=False, want_logits=True):
        """nvl_forward_paged: new tokens of each sequence at pos_offsets, KV through the sequences' block tables
        (Sequence.BlockTable).  Returns (logits [rows, V] or None, argmax [n_seqs])."""
        n = len(token_lists)
        lens = np.asarray([len(t) for t in token_lists], np.int32)
        toks = np.concatenate([np.asarray(t, np.int32) for t in token_lists]).astype(np.int32)
        pos = np.asarray(pos_offsets, np.int32)
        tbl = np.concatenate([np.asarray(b, np.int32) for b in block_tables]).astype(np.int32)
        off = np.concatenate([[0], np.cumsum([len(b) for b in block_tables])]).astype(np.int32)
        rows = int(lens.sum()) if all_logits else n
        logits = np.empty((rows, self.V), np.float32) if want_logits else None
        am = np.empty(n, np.int32)
        L.check(self.lib.nvl_forward_paged(self.h, n, _ptr(toks), _ptr(lens), _ptr(pos), _ptr(tbl), _ptr(off),
                                           L.FWD_ALL_LOGITS if all_logits else 0, _ptr(logits), _ptr(am)), self.h)
        return logits, am

    def decode_greedy_paged(self, first_tokens, positions, n_steps: int, block_tables):
        """nvl_decode_greedy_paged -> [n_steps, n_seqs]; the tables must cover positions + n_steps tokens."""
        n = len(first_tokens)
        first = np.ascontiguousarray(first_tokens, dtype=np.int32)
        pos = np.ascontiguousarray(positions, dtype=np.int32)
        tbl = np.concatenate([np.asarray(b, np.int32) for b in block_tables]).astype(np.int32)
        off = np.concatenate([[0], np.cumsum([len(b) for b in block_tables])]).astype(np.int32)
        out = np.empty((n_steps, n), dtype=np.int32)
        L.check(self.lib.nvl_decode_greedy_paged(self.h, n, _ptr(first), _ptr(pos), n_steps, _ptr(tbl), _ptr(off), _ptr(out)),
                self.h)
        return out

    def get_kv_paged(self, block_table, n_tokens: int, layer: int):
        tbl = np.ascontiguousarray(block_table, dtype=np.int32)
        nkv = self.cfg["num_heads"] if self.cfg["attention_type"] == "mha" else (
            1 if self.cfg["attention_type"] == "mqa" else self.cfg["num_kv_heads"])
        k = np.empty((nkv, n_tokens, self.cfg["head_dim"]), np.float32)
        v = np.empty_like(k)
        L.check(self.lib.nvl_get_kv_paged(self.h, _ptr(tbl), tbl.size, n_tokens, layer, _ptr(k), _ptr(v)), self.h)
        return k, v

    def decode_greedy(self, seq_ids, first_tokens, n_steps: int):
        """nvl_decode_greedy: n_steps greedy decode steps with the token feedback on the device -> [n_steps, n_seqs]."""
        n = len(seq_ids)
        ids = np.ascontiguousarray(seq_ids, dtype=np.int64)
        first = np.ascontiguousarray(first_tokens, dtype=np.int32)
        out = np.empty((n_steps, n), dtype=np.int32)
        L.check(self.lib.nvl_decode_greedy(self.h, n, _ptr(ids), _ptr(first), n_steps, _ptr(out)), self.h)
        return out

    def greedy_fused(self, prompt, max_tokens: int, seq_id: int = 0):
        """greedy() with the whole decode loop in one nvl_decode_greedy call."""
        self.seq_reset(seq_id)
        _, am = self.forward_batch([seq_id], [list(prompt)], [0], want_logits=False)
        out = [int(am[0])]
        if max_tokens > 1:
            out += [int(t) for t in self.decode_greedy([seq_id], [out[0]], max_tokens - 1)[:, 0]]
        return out

    def decode_sampled(self, seq_ids, first_tokens, n_steps: int, histories, uniforms, *, temperature=1.0, top_p=1.0,
                       top_k=0, repetition_penalty=1.2):
        """nvl_decode_sampled: n_steps sampled decode steps on the device -> [n_steps, n_seqs].  histories[i] = the
        sequence's token ids so far (ending with first_tokens[i]); uniforms [n_steps, n_seqs]."""
        from .ops import _histories
        n = len(seq_ids)
        ids = np.ascontiguousarray(seq_ids, dtype=np.int64)
        first = np.ascontiguousarray(first_tokens, dtype=np.int32)
        ptrs, lens, keep = _histories(histories, n)
        u = np.ascontiguousarray(uniforms, dtype=np.float32).reshape(n_steps, n)
        out = np.empty((n_steps, n), dtype=np.int32)
        sp = L.sampling_params(temperature, top_p, top_k, repetition_penalty)
        L.check(self.lib.nvl_decode_sampled(self.h, n, _ptr(ids), _ptr(first), n_steps, C.byref(sp), ptrs, _ptr(lens), _ptr(u),
                                            _ptr(out)), self.h)
        return out

    def sample(self, histories, uniforms, *, temperature=1.0, top_p=1.0, top_k=0, repetition_penalty=1.2):
        """nvl_sample: tensor.SampleWithHistory on the logits rows the last forward left on the device."""
        from .ops import _histories
        n = len(uniforms)
        ptrs, lens, keep = _histories(histories, n)
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        out = np.empty(n, np.int32)
        sp = L.sampling_params(temperature, top_p, top_k, repetition_penalty)
        L.check(self.lib.nvl_sample(self.h, n, C.byref(sp), ptrs, _ptr(lens), _ptr(u), _ptr(out)), self.h)
        return out

    # -- debug / parity taps ------------------------------------------------------------------------
    def set_debug(self, keep_hidden: bool):
        L.check(self.lib.nvl_set_debug(self.h, int(keep_hidden)), self.h)

    def get_hidden(self, n_tokens: int):
        out = np.empty((self.cfg["num_layers"], n_tokens, self.H), np.float32)
        for li in range(self.cfg["num_layers"]):
            L.check(self.lib.nvl_get_hidden(self.h, li, _ptr(out[li]), n_tokens * self.H), self.h)
        return out

    def get_mamba_state(self, seq_id: int, layer: int):
        """Mamba2Layer.SSMState of the sequence for a Mamba2 layer, [heads, head_dim, state]."""
        c = self.cfg
        hd = c.get("mamba_head_dim") or c["mamba_expand"] * c["hidden"] // c["mamba_num_heads"]
        out = np.empty((c["mamba_num_heads"], hd, c["mamba_state_size"]), np.float32)
        L.check(self.lib.nvl_get_mamba_state(self.h, seq_id, layer, _ptr(out)), self.h)
        return out

    def get_kv(self, seq_id: int, layer: int):
        T = self.seq_len(seq_id)
        at = self.cfg["attention_type"]
        nkv = self.cfg["num_heads"] if at == "mha" else (1 if at == "mqa" else self.cfg["num_kv_heads"])
        k = np.empty((nkv, T, self.cfg["head_dim"]), np.float32)
        v = np.empty_like(k)
        L.check(self.lib.nvl_get_kv(self.h, seq_id, layer, _ptr(k), _ptr(v)), self.h)
        return k, v

    # -- measurement -------------------------------------------------------------------------------
    def set_profile(self, on: bool):
        L.check(self.lib.nvl_set_profile(self.h, int(on)), self.h)

    def stats(self) -> dict:
        s = L.StatsC()
        L.check(self.lib.nvl_get_stats(self.h, C.byref(s)), self.h)
        return {k: getattr(s, k) for k, _ in L.StatsC._fields_}

    def kernel_stats(self) -> list:
        """nvl_get_kernel_stats: [{site, phase ('prefill'|'decode'), launches, ms, flops, bytes}, ...]."""
        n = self.lib.nvl_get_kernel_stats(self.h, None, 0)
        arr = (L.KernelStatC * max(n, 1))()
        n = min(n, self.lib.nvl_get_kernel_stats(self.h, arr, n))
        return [dict(site=self.lib.nvl_kernel_site_name(arr[i].site).decode(), phase=("prefill", "decode")[arr[i].phase],
                     launches=int(arr[i].launches), ms=arr[i].ms, flops=arr[i].flops, bytes=arr[i].bytes) for i in range(n)]

    def reset_stats(self):
        L.check(self.lib.nvl_reset_stats(self.h), self.h)

    def close(self):
        if self.h:
            self.lib.nvl_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            import sys
            if sys.is_finalizing():      # interpreter shutdown: the HIP runtime may already be gone (a leaked handle must
                return                   # not turn into nvl_destroy -> hipFree after teardown); the OS reclaims the memory
            self.close()
        except Exception:
            pass


def _as_pointer(arr):
    """(pointer, nvl dtype, shape, keepalive) for numpy arrays or torch tensors."""
    if isinstance(arr, np.ndarray):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        return a.ctypes.data_as(C.c_void_p), L.DTYPE_F32, a.shape, a
    import torch  # plumbing only: device memory handed over by pointer
    t = arr.contiguous()
    if t.dtype == torch.float32:
        dt = L.DTYPE_F32
    elif t.dtype == torch.bfloat16:
        dt = L.DTYPE_BF16
    elif t.dtype == torch.float16:
        dt = L.DTYPE_F16
    else:
        raise TypeError(t.dtype)
    if t.is_cuda:
        torch.cuda.synchronize(t.device)
    return C.c_void_p(t.data_ptr()), dt, tuple(t.shape), t
