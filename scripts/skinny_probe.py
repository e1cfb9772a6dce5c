"""Decode-shape GEMM probe: time vs N at M=32 (slope = streaming rate, intercept = fixed cost)."""
import importlib, sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, '..')
p = importlib.import_module('nano-vllm-go_amd')
L = p.lib()
def bench(M, N, K, epi, bnt=0, ks=0, iters=30):
    us = C.c_float()
    rc = L.nvl_bench_gemm(0, M, N, K, epi, bnt, ks, iters, C.byref(us))
    return None if rc else us.value
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for epi, name in ((0, "store"), (2, "swiglu"), (1, "resid")):
    for K in (2048, 8192):
        for N in (2048, 4096, 16384, 65536, 131072):
            if N * K * 2 > 1.2e9: continue
            us = bench(M, N, K, epi)
            print(f"M={M} {name:6s} N={N:6d} K={K:5d}: {us:8.1f} us  {N*K*2/us/1e6:6.2f} TB/s", flush=True)
