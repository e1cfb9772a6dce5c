"""The drop-in boundary without a GPU: the library loads, exports every symbol include/nvllm.h declares,
its structs match the ctypes mirrors, and it refuses to compute without a device (no CPU fallback)."""
import ctypes as C
import re

import pytest


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    names = pkg.declared_symbols()
    assert len(names) >= 30 and "nvl_forward" in names and "nvl_runner_run" in names
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.nvl_abi_version() == 3


def test_header_cites_the_reference_for_every_entry_point(pkg):
    text = pkg._lib.HEADER.read_text()
    for must in ("generic_model.go:276-480", "tensor_model_runner.go:55-97", "model_runner.go:9-16",
                 "generic_loader.go", "cmd/ask/main.go:389-402", "tensor.go:62-88", "rope.go:153-205", "moe.go:43-128"):
        assert must in text, must
    assert 'extern "C"' in text and "torch" not in re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def test_struct_mirrors_match_the_header(pkg):
    L = pkg._lib
    lib = pkg.lib()
    assert C.sizeof(L.ModelConfigC) == lib.nvl_sizeof(0) == 144     # 13 x i32, pad, f64, f32, i32, 3 x i32, 4 x f32, 6 x i32 (Mamba2), 2 x u64
    assert C.sizeof(L.RuntimeOptsC) == lib.nvl_sizeof(1) == 40
    assert C.sizeof(L.StatsC) == lib.nvl_sizeof(2) == 16 * 8
    assert C.sizeof(L.SamplingParamsC) == lib.nvl_sizeof(3) == 16
    assert [n for n, _ in L.ModelConfigC._fields_][:3] == ["vocab_size", "hidden", "num_layers"]
    assert len(L.SLOTS) == 33 and L.SLOT_ID["moe_out"] == 24 and L.SLOT_ID["mamba_out_proj"] == 32      # NVL_T_COUNT


def test_no_cpu_fallback(pkg):
    L = pkg.lib()
    if L.nvl_device_count() > 0:
        pytest.skip("a GPU is visible here")
    cfg = pkg.synth.tiny_config("llama")
    with pytest.raises(pkg.NvlError) as e:
        pkg.HipTransformerModel(cfg, pkg.synth.make_weights(cfg))
    assert e.value.code == -8                                      # NVL_ERR_NO_DEVICE
    import numpy as np
    with pytest.raises(pkg.NvlError) as e:
        pkg.ops.mat_mul(np.zeros((1, 64), np.float32), np.zeros((64, 16), np.float32))
    assert e.value.code == -8


def test_product_package_never_imports_the_oracle(pkg):
    from pathlib import Path
    root = Path(pkg.__file__).resolve().parent
    for p in list(root.rglob("*.py")) + list(root.rglob("*.h")) + list(root.rglob("*.hip")) + list(root.rglob("Makefile")):
        text = p.read_text()
        assert "purego_oracle" not in text and "import oracle" not in text and "from oracle" not in text, p


def test_create_validates_options_before_touching_a_device(pkg):
    """A zero-valued host options struct (max_seqs = 0) must be refused, not turned into a 0-block KV cache
    (the check runs before the device probe, so it is testable without a GPU)."""
    import ctypes as C
    L = pkg._lib
    lib = pkg.lib()
    from importlib import import_module
    cfgc = import_module("nano-vllm-go_amd.model")._cfg_struct(pkg.synth.tiny_config("llama"))
    h = C.c_void_p()
    for bad in (dict(max_seqs=0), dict(max_seqs=-3), dict(max_seqs=2, max_batch_tokens=-1), dict(max_seqs=2, kv_num_blocks=-1)):
        opts = L.RuntimeOptsC(device=0, precision=0, tp_size=1, **bad)
        assert lib.nvl_create(C.byref(cfgc), C.byref(opts), C.byref(h)) == -1, bad     # NVL_ERR_INVALID
        assert not h.value
        assert b"nvl_create" in lib.nvl_last_error(None)
