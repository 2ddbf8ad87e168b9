"""caphn: MI355X-native hot path of the hypernetwork-conditioned captioning step.

  caphn._lib        ctypes binding of libcaphn.so (C ABI: include/caphn.h)
  caphn.ops         tensor-level wrappers of the C entry points
  caphn.functional  torch.autograd.Function wrappers (module API)
  caphn.engine      fused training step (hypernet -> decoder -> CE -> backward -> clip -> Adam)
  caphn.dp          data-parallel gradient exchange (rank-1 factor all-gather + dense all-reduce)
"""
from . import config  # noqa: F401
