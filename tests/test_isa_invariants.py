"""ISA invariants of the hand-scheduled prefill attention kernel (scripts/check_isa.sh): hipcc cross-compiles without a
GPU, so this runs in the CPU suite.  What it pins: M0 is written only by the LDS-DMA inline assembly (it cannot be
declared as a clobber), the compiler has put no `s_waitcnt vmcnt(0)` into the tile loop (that serialises the LDS-DMA
ring: the kernel then still computes the right thing, only ~25 % slower, so no numerical test would notice), no spills.
And the register budgets of the decode kernels (two waves per SIMD for the hd-64 decode attention, bounded spills in the
decode projections): round 3 lost 5-14 % of decode throughput on GPT-2, Llama-3-8B and the 64-row-group projections to
changes measured only on the headline config — no numerical test notices those either."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="no hipcc")
def test_prefill_attention_isa_invariants(tmp_path):
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "check_isa.sh")], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "attn_prefill<64>" in r.stdout and "attn_prefill<128>" in r.stdout
    assert "BAD" not in r.stdout and r.stdout.count("\nok  ") >= 60
