import importlib, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from oracle import purego_oracle as O
p = importlib.import_module('nano-vllm-go_amd')
def rel(a,b): return float(np.abs(a-b).max()/np.abs(b).max())
for fam in ["llama","gpt2","falcon","granite_moe"]:
    cfg = p.synth.tiny_config(fam)
    w = p.synth.make_weights(cfg, seed=7, scale=0.05)
    om = O.OracleModel(cfg, w)
    toks = np.random.default_rng(1).integers(0, cfg["vocab_size"], 37).tolist()
    want, wh = om.forward_with_cache(toks, om.new_cache(), 0, want_hidden=True)
    for prec in ("f32","bf16"):
        hm = p.HipTransformerModel(cfg, w, precision=prec, max_seqs=2, max_batch_tokens=64)
        hm.set_debug(True)
        got = hm.forward_with_cache(toks, 1, 0)
        gh = hm.get_hidden(len(toks))
        print(fam, prec, "logits", f"{rel(got,want):.2e}", "hidden", [f"{rel(gh[i],wh[i]):.1e}" for i in range(cfg['num_layers'])])
        hm.close()
