#!/usr/bin/env python3
"""hypernet.py's literal configuration -- HyperNet(200, 150, 9684, vocab, 2, 'gru'), B=128, T=20: 2.8 G hypernet
parameters (11 GB fp32).  Runs forward + backward through the drop-in modules, checks theta and sampled gradients
against plain PyTorch on the same device, and times the phases (tuning aid / DESIGN.md numbers)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd"))
from hypernet import HyperNet  # noqa: E402


class Vocab:
    w2i = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3, "factual": 4}

    def __call__(self, w):
        return self.w2i.get(w, 3)


def main():
    dev = "cuda"
    if os.environ.get("CAPHN_TUNE"):                       # e.g. CAPHN_TUNE="1=0,0=0" -> caphn_tune(1, 0), caphn_tune(0, 0)
        from caphn import _lib
        for kv in os.environ["CAPHN_TUNE"].split(","):
            k, v = kv.split("=")
            assert _lib.load().caphn_tune(int(k), int(v)) == 0
    torch.manual_seed(0)
    E, H, V, L, B, T = 200, 150, 9684, 2, 128, 20
    with torch.device(dev):
        net = HyperNet(E, H, V, Vocab(), num_layers=L, type="gru")
    nparam = sum(p.numel() for p in net.hn_heads.parameters()) + sum(p.numel() for p in net.hn_base.parameters())
    print(f"hypernet parameters: {nparam/1e9:.3f} G ({nparam*4/1e9:.1f} GB), heads {[(h[0].out_features, h[2].out_features) for h in net.hn_heads]}")
    feats = torch.randn(B, E, device=dev, requires_grad=True)
    caps = torch.randint(0, V, (B, T), device=dev)
    style = torch.tensor([4], device=dev)

    def step():
        for p in net.parameters():
            p.grad = None
        cap = net(net.captioner.embed(style))
        torch.manual_seed(1)
        logits = cap(feats, caps, True)
        loss = F.cross_entropy(logits.view(-1, V), caps.view(-1))
        loss.backward()
        return loss, logits

    loss, logits = step()
    torch.cuda.synchronize()
    # reference: plain PyTorch on the same device
    with torch.no_grad():
        x = net.captioner.embed.weight[4:5]
        base = F.leaky_relu(F.linear(F.leaky_relu(F.linear(x, net.hn_base[0].weight, net.hn_base[0].bias)), net.hn_base[2].weight, net.hn_base[2].bias))
        outs, acts = [], []
        for h in net.hn_heads:
            a = F.leaky_relu(F.linear(base, h[0].weight, h[0].bias)); acts.append(a)
            outs.append(F.linear(a, h[2].weight, h[2].bias).flatten())
        theta_ref = torch.cat(outs)
    n0 = sum(getattr(net.captioner.lstm_cell, n).numel() for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
    got = torch.cat([getattr(net.captioner.lstm_cell, n).detach().flatten() for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")])
    print("theta (first cell) max |diff| vs torch:", float((got - theta_ref[:n0]).abs().max()), " |theta| max", float(theta_ref.abs().max()))
    # rank-1 structure of the big second-layer gradient: dW2_0 = dtheta_0 (x) a_0
    g = net.hn_heads[0][2].weight.grad
    db = net.hn_heads[0][2].bias.grad
    rows = torch.randint(0, g.shape[0], (64,), device=dev)
    print("dW2_0 rank-1 check max |diff|:", float((g[rows] - db[rows, None] * acts[0][0][None, :]).abs().max()), " |g| max", float(g.abs().max()))
    print("dead heads (layers.0.*) gradient abs sum:", float(sum(net.hn_heads[i][2].weight.grad.abs().sum() for i in range(4, 8))))
    print("loss", float(loss))
    for name, fn in [("hypernet forward (11 GB GEMV)", lambda: net(net.captioner.embed(style))), ("full forward+backward", step)]:
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{name}: {dt*1e3:.2f} ms" + (f"  ({nparam*4/dt/1e12:.2f} TB/s)" if "GEMV" in name else ""))
    # the fused step (caphn.engine_plain): forward, loss, backward and Adam with rank-1 second-layer gradients
    from caphn.engine_plain import FusedPlainTrainer
    for p in net.parameters():
        p.grad = None
    torch.cuda.empty_cache()
    tr = FusedPlainTrainer(net, lr=1e-6)
    live = sum(w.numel() for w in tr.W2)
    h0 = torch.rand(B, H, device=dev)
    f2 = feats.detach()
    for prefetch in (False, True):
        for _ in range(3):
            out = tr.step(f2, caps, 4, h0, None, next_style_token=4 if prefetch else None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            out = tr.step(f2, caps, 4, h0, None, next_style_token=4 if prefetch else None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        bpp = 28 if prefetch else 32
        if prefetch:
            import json
            print(json.dumps({"metric": "training images/sec, hypernet.py HyperNet(200, 150, 9684, vocab, 2, 'gru') (BASELINE config 2 as literally named)",
                              "value": B / dt, "unit": "images/s", "n_gpus": 1, "ms_per_step": dt * 1e3, "higher_is_better": True, "dtype": "f32",
                              "data": "synthetic", "config": {"workload": "fused step: hypernet forward (in the Adam pass), DecoderGRU 2 layers, cross "
                                                                          "entropy, backward, Adam; dead heads skipped", "per_gpu_batch": B, "T": T,
                                                              "hypernet_params": nparam, "live_second_layer_params": live},
                              "roofline": {"bound": "hbm", "achieved": live * 28 / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                                           "frac": live * 28 / dt / 8e12, "traffic": None,
                                           "note": "whole step over the 28 B per live second-layer parameter it must move (VJP 4 + Adam 24)"}}))
        print(f"fused step ({'next theta in the Adam pass' if prefetch else 'separate forward GEMV'}): {dt*1e3:.2f} ms = "
              f"{B/dt:.0f} img/s; live second-layer parameters {live/1e9:.3f} G of {nparam/1e9:.3f} G, "
              f"{bpp} B each -> {live*bpp/dt/1e12:.2f} TB/s;  loss {float(out[0]):.4f}")


if __name__ == "__main__":
    main()
