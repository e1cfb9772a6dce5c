"""Mid-size prefill batches (M = 2048 / 4096 / 8192): lock-step 256x128 (2) and 256x256 (3) vs ping-pong 256x256 (5) and 256x128 (6)."""
import importlib, os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = importlib.import_module('nano-vllm-go_amd')
L = p.lib()
def bench(M, N, K, epi, tile, iters=20):
    us = C.c_float()
    rc = L.nvl_bench_gemm(0, M, N, K, epi, tile, 0, iters, C.byref(us))
    return None if rc else us.value
for M in (2048, 4096, 8192):
    for name, N, K, epi in (("qkv", 3072, 2048, 0), ("o", 2048, 2048, 1), ("w1", 16384, 2048, 2), ("w2", 2048, 8192, 1)):
        row = []
        for tile in (0, 1, 2, 3, 5):
            us = bench(M, N, K, epi, tile)
            row.append(f"t{tile}: {2*M*N*K/us/1e6:6.0f}")
        print(f"M={M:5d} {name:4s}: " + "  ".join(row) + "  TF/s", flush=True)
