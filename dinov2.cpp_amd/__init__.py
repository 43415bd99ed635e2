"""dinov2.cpp_amd -- MI355X-native DINOv2 forward behind the dino_model_load / dino_predict surface.

The directory name contains a dot, so import it through `__graft_entry__.load_package()` (registers
the package as `dinov2_cpp_amd`).  Sub-modules:

  gguf_writer  GGUF v3 writer + ggml block quantisers (offline tooling, numpy)
  synth        seeded synthetic checkpoints in the reference's GGUF schema
  api          ctypes binding of the C-ABI (include/dinov2_hip.h) + dino_* mirror of the reference API
"""
from . import gguf_writer, synth  # noqa: F401
