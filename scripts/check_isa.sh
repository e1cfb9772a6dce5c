#!/bin/bash
# ISA invariants of the hand-scheduled kernels that the compiler can silently break (run after touching attn.h / gemm.h;
# hipcc cross-compiles without a GPU, ~1 min).  Checks, for attn_prefill_bf16_kernel<64> and <128>:
#   * M0 is written only by lds_dma16's inline assembly (it cannot be declared as a clobber: attn.h);
#   * the steady-state tile loop holds no compiler-inserted "s_waitcnt vmcnt(0)" (that would wait for the tile issued a
#     moment ago and serialise the LDS-DMA ring): the only vmcnt(0) in the kernel are the Q-fragment wait before the loop
#     and the hand-written one for the last tile;
#   * no register spills.
set -e
S=${TMPDIR:-/tmp}/nvllm_check.s
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable --cuda-device-only -S nano-vllm-go_amd/csrc/nvllm.hip -o $S 2>/dev/null
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
exit $rc
