"""Run ONE projection GEMM shape repeatedly (for rocprofv3 --pmc / --kernel-trace runs).
usage: gemm_one.py M N K epi tile iters"""
import ctypes as C, importlib, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = importlib.import_module('nano-vllm-go_amd')
M, N, K, epi, tile, iters = map(int, sys.argv[1:7])
us = C.c_float()
rc = p.lib().nvl_bench_gemm(0, M, N, K, epi, tile, 0, iters, C.byref(us))
print("rc", rc, f"{us.value:.1f} us  {2*M*N*K/us.value/1e6:.1f} TF/s")
