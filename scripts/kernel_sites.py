"""Shared by the profile post-processing scripts: label every dispatch of a rocprofv3 trace of bench.py with the launch
SITE the library's own per-site statistics use (nvl_kernel_site_name: qkv_proj, attention, o_proj, ffn_up, ffn_down,
lm_head ...) and with its phase (prefill / decode), from the kernel name, its template arguments and its neighbours —
the O and FFN-down projections share one kernel, so do the QKV projection and the LM head of a decode step."""
import re
import subprocess


def demangle(names):
    names = list(names)
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def _unmangle(n):
    """c++filt does not know the __bf16 mangling (DF16b): read _ZN3nvl<len><name>I<template args>E... by hand."""
    m = re.match(r"_ZN3nvl(\d+)", n)
    if not m:
        return n
    ln = int(m.group(1)); start = m.end()
    base, rest = n[start:start + ln], n[start + ln:]
    if not rest.startswith("I"):
        return base
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        if rest.startswith("Li", i) or rest.startswith("Lb", i) or rest.startswith("Lj", i):
            j = rest.index("E", i)
            v = rest[i + 2:j]
            args.append(("true" if v == "1" else "false") if rest[i + 1] == "b" else v)
            i = j + 1
        elif rest.startswith("DF16b", i):
            args.append("bf16"); i += 5
        elif rest[i] == "f":
            args.append("float"); i += 1
        else:
            break
    return f"{base}<{', '.join(args)}>"


def short(name):
    n = name[:name.find("(")] if "(" in name else name
    n = n.replace("void nvl::", "").replace("nvl::", "").strip()
    return _unmangle(n) if n.startswith("_ZN3nvl") else n


def _targs(name):
    m = re.search(r"<(.*)>", name)
    return [a.strip() for a in m.group(1).split(",")] if m else []


def gemm_epi(name):
    """EPI template argument of a projection kernel (0 STORE, 1 RESID, 2 SWIGLU, 3 GELU, 4 QKV), or None."""
    a = _targs(name)
    try:
        if "gemm_bf16_pp_kernel" in name:
            return int(a[0])
        if "gemm_bf16_kernel" in name:
            return int(a[5])
        if "gemm_skinny" in name:
            return int(a[3])
        if "gemm_f32_kernel" in name:
            return int(a[0])
    except (IndexError, ValueError):
        return None
    return None


def label_passes(rows, name_key="Kernel_Name"):
    """rows: dispatches in time order (dicts).  Adds r['site'], r['phase'], r['kname'] (demangled, short); returns the
    list of passes (lists of rows) that contain a projection."""
    dm = demangle({r[name_key] for r in rows})
    for r in rows:
        r["kname"] = short(dm[r[name_key]])
    passes, cur = [], []
    for r in rows:
        cur.append(r)
        n = r["kname"]
        if "argmax_final_kernel" in n or "decode_seam_kernel" in n or "sample_row_kernel" in n:
            passes.append(cur)
            cur = []
    if cur:
        passes.append(cur)
    passes = [p for p in passes if any("gemm" in r["kname"] for r in p)]
    for p in passes:
        # a decode step of an MQA model (Falcon: 71 query heads per KV head) runs the prefill attention kernel too (the decode
        # kernel takes groups of <= 16 heads): such a pass is told from a prefill by its projections, which are all on the
        # weight-streaming decode kernels
        tiles = any("gemm_bf16_pp_kernel" in r["kname"] or "gemm_bf16_kernel" in r["kname"] or "gemm_f32_kernel" in r["kname"] for r in p)
        phase = "prefill" if any("attn_prefill" in r["kname"] or "attn_f32" in r["kname"] for r in p) and tiles and \
            not any("attn_decode" in r["kname"] for r in p) else "decode"
        last = None
        for i, r in enumerate(p):
            n = r["kname"]
            nxt = p[i + 1]["kname"] if i + 1 < len(p) else ""
            site = "other"
            epi = gemm_epi(n)
            if "attn_" in n:
                site = "attention"
            elif epi is not None:
                if epi == 4:
                    site = "qkv_proj"
                elif epi in (2, 3):
                    site = "ffn_up"
                elif epi == 1:
                    site = "o_proj" if last == "attention" else "ffn_down"
                else:   # plain store: QKV (decode form: attention or rope_kv follows), LM head (argmax follows), MoE router
                    if "argmax" in nxt or "scale_rows" in nxt:
                        site = "lm_head"
                    elif "attn_" in nxt or "rope_kv" in nxt:
                        site = "qkv_proj"
                    elif last == "moe_up":          # the grouped expert-down GEMM (moe_combine follows it)
                        site = "moe_down"
                    elif "moe_" in nxt:
                        site = "moe_router"
                    else:
                        site = "other_gemm"
                if site == "ffn_up" and any("moe_" in q["kname"] for q in p):
                    site = "moe_up"
            elif "norm" in n:
                site = "norm"
            elif "moe_" in n:
                site = "moe_plan"
            elif "embed" in n or "decode_seam" in n:
                site = "embed"
            elif "argmax" in n:
                site = "argmax"
            elif "rope_kv" in n:
                site = "rope_kv"
            r["site"], r["phase"] = site, phase
            if site in ("attention", "ffn_up", "moe_up", "moe_down", "o_proj", "ffn_down", "qkv_proj"):
                last = site
    return passes
