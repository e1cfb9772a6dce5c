import importlib, sys
import numpy as np
sys.path.insert(0, '.')
p = importlib.import_module('nano-vllm-go_amd')
r = np.random.default_rng(0)
M, K, N = 4096, 2048, 2048
a = r.standard_normal((M, K), dtype=np.float32)
b = r.standard_normal((K, N), dtype=np.float32) * 0.05
outs = {}
for tile in (1, 3, 5):
    old = p.lib().nvl_set_tuning(0, tile)
    outs[tile] = p.ops.mat_mul(a, b, precision="bf16")
    p.lib().nvl_set_tuning(0, old)
for t in (3, 5):
    d = np.abs(outs[t] - outs[1])
    print("tile", t, "vs 1: equal", np.array_equal(outs[t], outs[1]), "max abs diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "max", np.abs(outs[1]).max())
for rep in range(3):
    old = p.lib().nvl_set_tuning(0, 5)
    again = p.ops.mat_mul(a, b, precision="bf16")
    p.lib().nvl_set_tuning(0, old)
    print("tile 5 repeat equal:", np.array_equal(again, outs[5]))
