#!/usr/bin/env python3
"""Time of the three attention kernels on CATR's shapes (bs 64, 8 heads of 32)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd"))
from caphn import ops  # noqa: E402


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    dev = "cuda"
    bs, nh, dm = 64, 8, 256
    for name, tq, tk, causal in (("enc self", 49, 49, False), ("dec self", 128, 128, True), ("cross", 128, 49, False)):
        q, k, v = torch.randn(tq, bs, dm, device=dev), torch.randn(tk, bs, dm, device=dev), torch.randn(tk, bs, dm, device=dev)
        mask = torch.full((tq, tk), float("-inf"), device=dev).triu(1) if causal else None
        o, lse = ops.attention_fwd(q, k, v, nh, mask, None)
        d_o = torch.randn_like(o)
        tf = timed(lambda: ops.attention_fwd(q, k, v, nh, mask, None))
        tb = timed(lambda: ops.attention_bwd(q, k, v, nh, mask, None, o, lse, d_o))
        flop_f = 4.0 * bs * nh * tq * tk * 32
        print(f"{name:9s} tq {tq:3d} tk {tk:3d}: forward {tf:6.1f} us ({flop_f/tf/1e6:5.2f} TFLOP/s)  backward (dq + dk/dv) {tb:6.1f} us")


if __name__ == "__main__":
    main()
