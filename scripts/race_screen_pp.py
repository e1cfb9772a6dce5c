"""Race screen for the ping-pong GEMM (a new synchronisation structure): many repeats at several shapes, every result
compared bit for bit with the lock-step 128x128 kernel (same accumulation order => identical bits)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = importlib.import_module('nano-vllm-go_amd')
r = np.random.default_rng(1)
bad = 0
for (M, K, N, reps) in ((8192, 2048, 8192, 25), (16384, 192, 4096, 25), (4100, 4096, 4300, 25), (65536, 256, 1024, 15), (2048, 8192, 32768, 10)):
    a = r.standard_normal((M, K), dtype=np.float32)
    b = r.standard_normal((K, N), dtype=np.float32) * 0.05
    old = p.lib().nvl_set_tuning(0, 1)
    ref = p.ops.mat_mul(a, b, precision="bf16")
    p.lib().nvl_set_tuning(0, 5)
    n_bad = 0
    for i in range(reps):
        got = p.ops.mat_mul(a, b, precision="bf16")
        if not np.array_equal(got, ref):
            n_bad += 1
            d = np.abs(got - ref)
            print("  MISMATCH rep", i, "max diff", d.max(), "count", int((d > 0).sum()), flush=True)
    p.lib().nvl_set_tuning(0, old)
    bad += n_bad
    print(f"M={M} K={K} N={N}: {reps - n_bad}/{reps} identical", flush=True)
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
