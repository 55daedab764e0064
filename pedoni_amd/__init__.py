"""pedoni_amd -- MI355X-native backend for qt2/pedoni's per-timestep pedestrian update.

Only what the hot path needs lives here: ``csrc/`` (hand-written gfx950 kernels, the
C-ABI of ``include/pedoni_hip.h`` and the C++ host mirror of pedoni-simulator's
``Simulator``) and thin ctypes bindings.  There is no CPU fallback: every compute entry
point raises when the HIP library or a GPU is missing.
"""
from .abi import HipModel, PedoniError, Options, load_library, library_path  # noqa: F401

__all__ = ["HipModel", "PedoniError", "Options", "load_library", "library_path"]
