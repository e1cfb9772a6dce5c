"""Compiles csrc/ into lib/libnvllm_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent


def build(force: bool = False) -> Path:
    args = ["make", "-C", str(_PKG / "csrc"), "-s"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    out = _PKG / "lib" / "libnvllm_hip.so"
    assert out.exists(), out
    return out


if __name__ == "__main__":
    print(build())
