#!/usr/bin/env python3
"""Per-phase shader-clock profile of the persistent recurrent kernels (workgroup 0), canonical size."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd")); sys.path.insert(0, ROOT)
from caphn import ops
dev = "cuda"
dims = ops.DecDims(128, 20, 49, 2048, 200, 200, 200, 9684)
p = {n: (torch.rand(s, device=dev) - 0.5) * 0.14 for n, s in dims.param_shapes().items()}
feats = torch.relu(torch.randn(128, 49, 2048, device=dev)) * 0.45
caps = torch.randint(1, 9684, (128, 20), device=dev)
ws = ops.decoder_workspace(dims, dev)
import ctypes
from caphn import _lib
for rot in (0, 1):
  _lib.load().caphn_tune(3, rot)
  print("=== row rotation", rot)
  for _ in range(3):
      logits, alphas = ops.decoder_forward(dims, p, feats, caps, ws)
      lo, dl = ops.cross_entropy_fwd_bwd(logits, caps, 0)
      grads = {n: torch.empty(s, device=dev) for n, s in dims.param_shapes().items()}
      ops.decoder_backward(dims, p, feats, caps, dl, grads, ws)
  torch.cuda.synchronize()
  # workspace tail (csrc/decoder.hip, layout()): ... | prof: 64 floats | rowmap: up4(B*T + 4) ints |
  tail = ((128 * 20 + 4 + 3) // 4) * 4
  wsf = ws.view(torch.float32)
  prof = wsf[wsf.numel() - tail - 64: wsf.numel() - tail].contiguous().view(torch.int64).cpu().tolist()
  if not any(prof[:16]):
      print("all counters are zero: build csrc/recurrent_gru.hip with -DCAPHN_REC_PROFILE (phase stamps are compiled out by default)")
  fw, bw = prof[:8], prof[8:16]
  names_f = ["A matvec W_hh,U_a", "B scores", "C softmax", "D1 alpha.G partial", "D2 gates"]
  names_b = ["P1 cell pointwise", "P2 dalpha=G.dgi", "P3 softmax bwd", "P4 d(uah) partial", "P4b sum", "P5 matvec^T", "P6 sum"]
  print("forward  (cycles per step, 20 steps): total %.0f" % (sum(fw) / 20))
  for n, v in zip(names_f, fw): print(f"   {n:22s} {v/20:9.0f}")
  print("backward (cycles per step): total %.0f" % (sum(bw) / 20))
  for n, v in zip(names_b, bw): print(f"   {n:22s} {v/20:9.0f}")
