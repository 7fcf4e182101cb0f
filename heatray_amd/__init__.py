"""heatray_amd — MI355X-native per-pass ray kernel behind the HeatrayRenderer / RLWrapper API.

Layout:
  csrc/    hand-written HIP kernels + the C-ABI library libhrcore.so (include/hrcore.h)
  host/    C++ drop-in layer mirroring PassGenerator / Scene / Mesh / Material / Light /
           openrl::Texture / openrl::PixelPackBuffer over the C-ABI
  core.py  loader for libhrcore.so (fails loudly when the extension is missing)
  host.py, scenes.py, tiles.py  Python twins of the host logic used by bench.py and the tests
"""
__all__ = ["core", "host", "scenes"]
