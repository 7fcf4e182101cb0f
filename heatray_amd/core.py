"""Loader for the product library libhrcore.so (hand-written HIP behind include/hrcore.h).

There is no CPU fallback: if the library is missing or no HIP device is usable the calls fail.
"""
import ctypes
import os

from ._ffi import ABI_SYMBOLS, Engine, EngineError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HRCORE_LIB") or os.path.join(_HERE, "csrc", "libhrcore.so")  # HRCORE_LIB: A/B builds in experiments
_LIB = None


def load_library():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _LIB = ctypes.CDLL(LIB_PATH)
        missing = [s for s in ABI_SYMBOLS if not hasattr(_LIB, "hr_" + s)]
        if missing and not (os.environ.get("HRCORE_LIB") and os.environ.get("HRCORE_ALLOW_OLD_ABI")):  # A/B against older builds only
            raise EngineError(f"libhrcore.so lacks C-ABI symbols: {missing}")
    return _LIB


def create_engine(device_id=0, rank=0, world=1, tile_size=32, stream=None, collect_stats=False, time_kernels=False, memory_budget=0):
    from ._ffi import HR_CTX_COLLECT_STATS, HR_CTX_TIME_KERNELS
    flags = (HR_CTX_COLLECT_STATS if collect_stats else 0) | (HR_CTX_TIME_KERNELS if time_kernels else 0)
    return Engine(load_library(), "hr_", device_id=device_id, rank=rank, world=world, tile_size=tile_size,
                  stream=stream, flags=flags, memory_budget=memory_budget)
