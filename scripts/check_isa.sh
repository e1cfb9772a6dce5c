#!/bin/bash
# ISA invariants of the hand-scheduled kernels that the compiler can silently break (run after touching attn.h / gemm.h;
# hipcc cross-compiles without a GPU, ~1 min).  Checks, for attn_prefill_bf16_kernel<64> and <128>:
#   * M0 is written only by lds_dma16's inline assembly (it cannot be declared as a clobber: attn.h);
#   * the steady-state tile loop holds no compiler-inserted "s_waitcnt vmcnt(0)" (that would wait for the tile issued a
#     moment ago and serialise the LDS-DMA ring): the only vmcnt(0) in the kernel are the Q-fragment wait before the loop
#     and the hand-written one for the last tile;
#   * no register spills.
# And the register budgets of the decode kernels (from the code object metadata in the same assembly): round 3 lost 5-14 % of
# decode throughput on three configs to changes that were only measured on the headline one — the MHA / small-batch
# decode-attention instantiations went from two waves per SIMD to one, the 64-row-group projections spilled 34-54 registers.
set -e
S=${TMPDIR:-/tmp}/nvllm_check.s
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=6 -Wno-unused-function -Wno-unused-variable --cuda-device-only -S nano-vllm-go_amd/csrc/nvllm.hip -o $S 2>/dev/null
rc=0
for hd in 64 128; do
  K=$S.attn$hd
  awk "/^_ZN3nvl24attn_prefill_bf16_kernelILi${hd}EEEvNS_8AttnArgsE:/,/s_endpgm/" $S > $K
  m0=$(grep "m0" $K | grep -vc "s_mov_b32 m0" || true)
  loop0=$(awk '/Loop Header: Depth=1/,0' $K | grep -B1 "s_waitcnt vmcnt(0)" | grep -c "ASMSTART" || true)
  all0=$(awk '/Loop Header: Depth=1/,0' $K | grep -c "s_waitcnt vmcnt(0)" || true)
  spills=$(grep -A40 "\.name: *_ZN3nvl24attn_prefill_bf16_kernelILi${hd}EEEvNS_8AttnArgsE$" $S | grep -m1 vgpr_spill_count | awk '{print $2}')
  echo "attn_prefill<$hd>: m0 outside asm=$m0  vmcnt(0) after loop entry: $all0 (of them hand-written: $loop0)  vgpr spills=$spills"
  [ "$m0" = 0 ] && [ "$spills" = 0 ] && [ "$all0" = "$loop0" ] || rc=1
done
python3 - $S <<'PY' || rc=1
import re, sys
meta = open(sys.argv[1]).read()
kern = {}
for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
    g = lambda k: int(re.search(r"\." + k + r": *(\d+)", blk).group(1))
    name = re.search(r"\.name: *(\S+)", blk).group(1)
    agpr = int(re.match(r" *(\d+)", blk).group(1))
    kern[name] = dict(vgpr=g("vgpr_count"), agpr=agpr, spill=g("vgpr_spill_count"))
bad = 0
def check(pattern, what, ok):
    global bad
    hits = [k for k in kern if re.search(pattern, k)]
    if not hits:
        print("no kernel matches", pattern); bad += 1
    for k in hits:
        r = kern[k]
        good = ok(r)
        print(("ok  " if good else "BAD ") + what + ": " + k[:96] + "  vgpr(+agpr) %d spills %d" % (r["vgpr"], r["spill"]))
        bad += 0 if good else 1
# decode attention, hd 64 (fused and plain): two waves per SIMD (<= 256 unified registers), nothing spilled
check(r"attn_decode_bf16_kernelILi64ELi[248]ELb[01]ELb0E", "decode attention hd 64, 2 waves/SIMD", lambda r: r["vgpr"] <= 256 and r["spill"] == 0)
check(r"attn_decode_bf16_kernelILi128ELi8ELb[01]ELb0E", "decode attention hd 128, 8 waves", lambda r: r["vgpr"] <= 256 and r["spill"] <= 8)
# decode projections: the 64-row-group instantiations (PASSES) and the wide form
check(r"gemm_skinny_bf16_kernelILi4ELi[124]ELi2ELi[0-4]E\w+Lb1ELb0ELb0E", "narrow projection, 64-row groups", lambda r: r["spill"] <= 12)
check(r"gemm_skinny_bf16_kernelILi[12]ELi[124]ELi4ELi[0-4]E\w+Lb0ELb0ELb[01]E", "narrow projection, <= 32 rows", lambda r: r["spill"] <= 12)
check(r"gemm_skinny_wide_bf16_kernelILi[12]ELi[24]ELi4E", "wide projection, <= 32 rows", lambda r: r["spill"] <= 12)
sys.exit(1 if bad else 0)
PY
exit $rc
