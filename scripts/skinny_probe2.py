"""Wide vs narrow decode GEMM forms at the Llama shapes, by M and ksplit."""
import importlib, sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, '..')
p = importlib.import_module('nano-vllm-go_amd')
L = p.lib()
def bench(M, N, K, epi, bnt=0, ks=0, iters=30):
    us = C.c_float()
    rc = L.nvl_bench_gemm(0, M, N, K, epi, bnt, ks, iters, C.byref(us))
    return None if rc else us.value
for M in (8, 16, 32, 64):
    for name, N, K, epi in (("w1", 16384, 2048, 2), ("lm", 128256, 2048, 0), ("w1-8b", 28672, 4096, 2)):
        row = []
        for form, ks in ((2 if epi == 2 else 1, 0), (8, 8), (8, 4), (8, 2)):
            us = bench(M, N, K, epi, form, ks)
            row.append(f"form{form}/ks{ks}: {us:7.1f} us {N*K*2/us/1e6:5.2f} TB/s")
        print(f"M={M:2d} {name:6s}: " + "  ".join(row), flush=True)
