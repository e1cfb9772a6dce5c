"""Load-order probe (GPU box): the library before the first `import torch` — fixed by _lib.lib() importing torch first.
usage: dbg_nodev.py lib_first | torch_first"""
import importlib, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
pkg = importlib.import_module("nano-vllm-go_amd")
mode = sys.argv[1]
L = pkg.lib()
print("device_count", L.nvl_device_count(), flush=True)
import torch
if mode == "torch_first":
    torch.zeros(1, device="cuda")
cfg = dict(pkg.synth.FULL_CONFIGS["granite-3.0-1b-a400m"]); cfg["num_layers"] = 3; cfg["vocab_size"] = 4096
hm = pkg.HipTransformerModel(cfg, None, precision="bf16", max_seqs=5, max_batch_tokens=160)
print("model created", flush=True)
try:
    g = torch.Generator(device=torch.device("cuda", 0))
    print(mode, "generator OK", flush=True)
except Exception as e:
    print(mode, "generator FAILED:", str(e)[:100], flush=True)
    try:
        print("is_available", torch.cuda.is_available(), torch.cuda.device_count(), flush=True)
    except Exception as e2:
        print("is_available failed", e2)
