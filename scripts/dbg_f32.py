import importlib, sys, numpy as np
sys.path.insert(0, '.')
from oracle import purego_oracle as O
p = importlib.import_module('nano-vllm-go_amd')
def run(fam, prec, n, debug=True):
    cfg = p.synth.tiny_config(fam)
    w = p.synth.make_weights(cfg, seed=7, scale=0.05)
    om = O.OracleModel(cfg, w)
    toks = np.random.default_rng(1).integers(0, cfg["vocab_size"], n).tolist()
    want = om.forward_with_cache(toks, om.new_cache(), 0)
    hm = p.HipTransformerModel(cfg, w, precision=prec, max_seqs=4, max_batch_tokens=256)
    hm.set_debug(debug)
    got = hm.forward_with_cache(toks, 5, 0)
    if debug: hm.get_hidden(n)
    err = np.abs(got - want).max(axis=1) / np.abs(want).max()
    bad = np.nonzero(err > (1e-4 if prec == "f32" else 1.5e-2))[0]
    print(fam, prec, n, "bad rows", bad[:12], "n", len(bad), "max", err.max())
    if len(bad):
        r = bad[0]; cols = np.nonzero(np.abs(got[r]-want[r]) > 1e-3)[0]
        print("   row", r, "bad cols", cols[:10], len(cols), got[r, cols[:4]], want[r, cols[:4]])
    hm.close()
for fam in ["llama", "gpt2", "falcon", "granite_moe"]:
    for prec in ["f32", "bf16"]:
        run(fam, prec, 37)
for fam in ["llama", "gpt2", "falcon", "granite_moe"]:
    for prec in ["f32", "bf16"]:
        run(fam, prec, 131)
