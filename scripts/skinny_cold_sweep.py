"""Decode GEMM forms with COLD weights (nvl_bench_gemm rotates > 512 MiB of weight copies): narrow form NTW 1/2/4 and
wide form, by K split, at the Llama-3.2-1B projection shapes."""
import importlib, sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, '..')
p = importlib.import_module('nano-vllm-go_amd')
L = p.lib()
def bench(M, N, K, epi, form, ks, iters=40):
    us = C.c_float()
    rc = L.nvl_bench_gemm(0, M, N, K, epi, form, ks, iters, C.byref(us))
    return None if rc else us.value
shapes = [("qkv", 3072, 2048, 0), ("o", 2048, 2048, 1), ("w1", 16384, 2048, 2), ("w2", 2048, 8192, 1), ("lm", 128256, 2048, 0)]
for M in (int(x) for x in (sys.argv[1:] or ["32"])):
    for name, N, K, epi in shapes:
        forms = ((2, 8) if epi == 2 else (1, 2, 4, 8))
        for form in forms:
            if form == 8 and N < 16384: continue
            row = []
            for ks in (1, 2, 4, 8, 16):
                if form == 8 and ks > 8: continue
                us = bench(M, N, K, epi, form, ks, iters=20 if N > 100000 else 40)
                row.append("   n/a" if us is None else f"{us:6.1f}")
            print(f"M={M:2d} {name:4s} form{form}: ks1,2,4,8,16 = {' '.join(row)} us   ({N*K*2/1e6:.1f} MB)", flush=True)
