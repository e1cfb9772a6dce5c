import importlib, sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, '..')
p = importlib.import_module('nano-vllm-go_amd')
L = p.lib()
def bench(M,N,K,epi,bnt=0,ks=0,iters=50):
    us = C.c_float()
    rc = L.nvl_bench_gemm(0,M,N,K,epi,bnt,ks,iters,C.byref(us))
    if rc: return None
    return us.value
shapes = [("qkv",3072,2048,0),("o",2048,2048,1),("w1",16384,2048,2),("w2",2048,8192,1),("lm",128256,2048,0)]
for M in ():
    for name,N,K,epi in shapes:
        for bnt in ((2,) if epi==2 else (1,2)):
            row=[]
            for ks in (2,4,8,16):
                us = bench(M,N,K,epi,bnt,ks)
                row.append("   n/a " if us is None else f"{us:7.1f}")
            gb = N*K*2/1e9
            best = min([float(x) for x in row if 'n/a' not in x])
            print(f"M={M} {name:4s} N={N:6d} K={K:5d} bnt={bnt} ksplit(2,4,8,16): {' '.join(row)} us  best {gb/best*1e6/1e3:6.2f} TB/s")
# prefill shapes
for name,M,N,K,epi in [("qkv",16384,3072,2048,0),("o",16384,2048,2048,1),("w1",16384,16384,2048,2),("w2",16384,2048,8192,1),("sq4k",4096,4096,4096,0),("sq8k",8192,8192,8192,0)]:
    row = []
    for tile in (1, 3, 5):
        us = bench(M,N,K,epi,bnt=tile,iters=10)
        row.append(f"tile{tile}: {us:8.1f} us {2*M*N*K/us/1e6:7.1f} TF/s")
    print(f"prefill {name:5s} M={M} N={N} K={K}: " + "   ".join(row))
