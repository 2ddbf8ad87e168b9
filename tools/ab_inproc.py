#!/usr/bin/env python3
"""A/B of one FusedTrainer attribute INSIDE one process: blocks of steps with the attribute alternating between two values (the boxes'
HBM speed drifts by up to 10 % between processes, more than most schedule changes are worth).
  python tools/ab_inproc.py <attribute | tune:KEY> <value A> <value B> [rounds=8] [steps per block=40]
Prints ms/step per block and the medians."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hypernet-image-captioning_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402
from hypernet_attention import HyperNet  # noqa: E402
from caphn.engine import FusedTrainer  # noqa: E402


def parse(v):
    return {"True": True, "False": False}.get(v, int(v) if v.lstrip("-").isdigit() else v)


def main():
    attr, va, vb = sys.argv[1], parse(sys.argv[2]), parse(sys.argv[3])
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    block = int(sys.argv[5]) if len(sys.argv) > 5 else 40
    dev = torch.device("cuda", 0)
    B, T, P, D, F, E, H, V = 128, 20, 49, 2048, 200, 200, 200, 9684
    torch.manual_seed(1234)
    net = HyperNet(F, E, H, V, bench._Vocab()).to(dev)
    tr = FusedTrainer(net, lr=1e-3, max_norm=5.0)
    trs = None
    if attr.startswith("init:"):          # a construction-time switch (environment variable): TWO trainers, alternating blocks
        var = attr.split(":")[1]
        trs = []
        for v in (va, vb):
            os.environ[var] = str(v)
            torch.manual_seed(1234)
            trs.append(FusedTrainer(HyperNet(F, E, H, V, bench._Vocab()).to(dev), lr=1e-3, max_norm=5.0))
        del os.environ[var]
    from caphn import _lib
    lib = _lib.load()
    cur = [tr]
    if trs is not None:
        def setv(v):
            cur[0] = trs[0] if v == va else trs[1]
    elif attr.startswith("tune:"):          # a caphn_tune key instead of a trainer attribute
        key = int(attr.split(":")[1])

        def setv(v):
            assert lib.caphn_tune(key, int(v)) == 0
    else:
        assert hasattr(tr, attr), attr

        def setv(v):
            setattr(tr, attr, v)
    batches = bench.synth_batches(4, B, T, P, D, V, dev, seed=1234)
    j = [0]

    def run(n):
        for _ in range(n):
            f, c = batches[j[0] % 4]
            nf, nc = batches[(j[0] + 1) % 4]
            cur[0].step(f, c, style_token=4 + j[0] % 3, next_style_token=4 + (j[0] + 1) % 3, next_features=nf, next_captions=nc)
            j[0] += 1
    if trs is not None:
        for t in trs:
            cur[0] = t
            run(40)
    else:
        run(60)
    res = {0: [], 1: []}
    for r in range(rounds):
        for k, v in ((0, va), (1, vb)):
            setv(v)
            run(4)
            torch.cuda.synchronize()
            t = time.perf_counter()
            run(block)
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t) / block * 1e3)
        print(f"round {r}: {attr}={va!r}: {res[0][-1]:.4f}   {attr}={vb!r}: {res[1][-1]:.4f}   (B - A: {(res[1][-1] - res[0][-1]) * 1e3:+.1f} us)")
    a, b = float(np.median(res[0])), float(np.median(res[1]))
    print(f"median {attr}={va!r}: {a:.4f} ms/step   {attr}={vb!r}: {b:.4f} ms/step   B - A = {(b - a) * 1e3:+.1f} us")


if __name__ == "__main__":
    main()
