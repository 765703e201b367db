"""fhe-linformer_amd — MI355X-native RNS-CKKS evaluation engine behind the FHE-Linformer `FHEController` API.

The product is the HIP/C++ shared library ``libfhelin_amd.so`` (sources in ``csrc/``, C ABI in
``include/fhelin.h``).  This Python package is only a ctypes binding used by tests and bench.py.
There is no CPU fallback: importing works without a GPU (so the C-ABI can be inspected), but every
evaluation call fails loudly when the library or a HIP device is missing.
"""
from .capi import (  # noqa: F401
    FhelinError, Params, Engine, load_library, library_path, PRESETS, circuit_rotation_indices,
)
