"""MI355X-native hot path of MLAgg-UNet 2D training (reference: aticejiang/MLAgg-UNet).

Layout:
  csrc/   hand-written gfx950 HIP kernels + the C ABI (include/mlagg_hip.h) -> libmlagg_hip.so
  _lib.py ctypes binding of that ABI (fails loudly when the library is missing)
  ops.py  torch.autograd.Function wrappers (selective scan, local / pooled differential attention)
  shims.py drop-in modules named like the reference's third-party imports
  model.py the network behind nnUNetTrainer_MLAgg_2D_dt_MS.build_network_architecture
  trainer.py train step / loss / DDP wiring mirroring nnUNetTrainer.train_step
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
