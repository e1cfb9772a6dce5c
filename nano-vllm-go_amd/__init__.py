"""nano-vllm-go_amd — MI355X-native (gfx950) replacement for nano-vllm-go's purego/tensor
prefill + decode forward path, behind the C ABI of include/nvllm.h.

The directory name is the project's; it is not a Python identifier, so import it with
    importlib.import_module("nano-vllm-go_amd")
This package holds only what the path needs: csrc/ (HIP kernels + C ABI) and the host-side mirror of
the reference's TransformerModel / TensorModelRunner interface.  It never imports oracle/.
"""
from . import _lib, config, dist, ops, synth  # noqa: F401
from ._lib import NvlError, declared_symbols, lib  # noqa: F401
from .model import HipTransformerModel  # noqa: F401
from .runner import HipModelRunner, HipPagedModelRunner, Sequence  # noqa: F401
