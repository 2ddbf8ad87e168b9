#!/usr/bin/env python3
"""BASELINE config 5 at its per-GPU share (bs = 512 / 8 = 64): Caption = generated 1x1 projection -> 6+6 pre-norm
transformer (d 256, 8 heads, ff 2048) -> MLP head over 30522 tokens, T = 128, cross entropy; forward + backward
through the drop-in modules (module API: the loss and the optimiser stay the caller's).  Side measurement for DESIGN.md."""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd"))
from baseline.caption import Caption  # noqa: E402
from caphn import functional as CF  # noqa: E402
from baseline.configuration import Config  # noqa: E402
from baseline.transformer import build_transformer  # noqa: E402


class Backbone(torch.nn.Module):
    """precomputed ResNet feature maps stand where the backbone goes (the trunk is out of scope)."""

    def forward(self, samples):
        src, pos = samples
        return [(src, torch.zeros(src.shape[0], src.shape[2], src.shape[3], dtype=torch.bool, device=src.device))], [pos]


def main():
    dev = "cuda"
    torch.manual_seed(0)
    bs = int(os.environ.get("CATR_BS", "64"))
    hw = int(os.environ.get("CATR_HW", "7"))
    cfg = Config(dropout=float(os.environ.get("CATR_DROPOUT", "0.1")))       # baseline/configuration.py:28
    net = Caption(Backbone(), build_transformer(cfg), cfg.hidden_dim, cfg.vocab_size, hyper_emb=10).to(dev)
    n_tr = sum(p.numel() for p in net.transformer.parameters())
    n_hy = sum(p.numel() for p in net.input_proj.parameters())
    print(f"transformer {n_tr/1e6:.1f} M, MLP {sum(p.numel() for p in net.mlp.parameters())/1e6:.1f} M, projection hypernet {n_hy/1e6:.1f} M parameters")
    src = torch.randn(bs, 2048, hw, hw, device=dev)
    pos = torch.randn(bs, cfg.hidden_dim, hw, hw, device=dev)
    T = cfg.max_position_embeddings
    tgt = torch.randint(1, cfg.vocab_size, (bs, T), device=dev)
    tmask = torch.zeros(bs, T, dtype=torch.bool, device=dev)
    x = torch.zeros(10, device=dev); x[3] = 1.0

    from caphn.optim import FusedAdam
    opt = FusedAdam(net.parameters(), lr=1e-4)               # cc_train_catr.py:56-61: Adam over every parameter
    with_opt = os.environ.get("CATR_OPT", "1") == "1"
    if os.environ.get("CATR_RANK1", "1") == "1":
        opt.adopt_rank1(net.input_proj)                      # the 550 M-parameter head of the generated projection

    def step():
        opt.zero_grad()
        out = net((src, pos), tgt, tmask, x)
        loss = CF.cross_entropy(out.reshape(-1, cfg.vocab_size), tgt.reshape(-1))       # cc_train_catr.py: nn.CrossEntropyLoss()
        loss.backward()
        if with_opt:
            opt.step()
        return loss

    for _ in range(2):
        loss = step()
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    what = "forward+backward+Adam" if with_opt else "forward+backward"
    # algorithmic matrix work of one forward (multiply-add = 2 flop), x3 for forward + both backward contractions
    d, ff, V, S, rows_t, rows_s = cfg.hidden_dim, cfg.dim_feedforward, cfg.vocab_size, hw * hw, bs * T, bs * hw * hw
    dh = d // cfg.nheads
    enc = cfg.enc_layers * (4 * rows_s * d * d + 2 * rows_s * d * ff + 2 * rows_s * S * d) * 2
    dec = cfg.dec_layers * (6 * rows_t * d * d + 2 * rows_s * d * d + 2 * rows_t * d * ff + 2 * rows_t * T * d + 2 * rows_t * S * d) * 2
    head = (rows_t * d * 512 + rows_t * 512 * 512 + rows_t * 512 * V) * 2 + rows_s * 2048 * d * 2
    flops = 3.0 * (enc + dec + head)
    print(f"matrix work {flops/1e12:.2f} TFLOP per step (fp32) -> {flops/dt/1e12:.1f} TFLOP/s = {flops/dt/157.3e12:.2f} of the 157.3 TFLOP/s "
          f"fp32 MFMA peak (each fp32 product is 6 bf16 MFMA products here: {6*flops/dt/1e12:.0f} TFLOP/s of bf16 work, "
          f"{6*flops/dt/2.5e15:.2f} of the 2.5 PFLOP/s bf16 peak)")
    import json
    print(json.dumps({"metric": "training images/sec, CATR + hypernet-generated proj weights (BASELINE config 5), per-GPU share of bs=512 at DP=8",
                      "value": bs / dt, "unit": "images/s", "n_gpus": 1, "ms_per_step": dt * 1e3, "higher_is_better": True, "dtype": "f32",
                      "data": "synthetic", "step": what,
                      "config": {"workload": "Caption: generated 1x1 input_proj -> 6+6 pre-norm transformer (d 256, 8 heads, ff 2048) -> MLP head, "
                                             "cross entropy, Adam", "per_gpu_batch": bs, "positions": hw * hw, "T": T, "V": cfg.vocab_size,
                                 "dropout": cfg.dropout, "launch": "eager (module API)"},
                      "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flops / dt / 157.3e12,
                                   "traffic": None, "note": "whole step's fp32 matrix work over the fp32 MFMA peak; each fp32 product is six bf16 MFMA products"}}))
    print(f"bs {bs}, {hw}x{hw} positions, T {T}, dropout {cfg.dropout}: {what} {dt*1e3:.1f} ms = {bs/dt:.0f} img/s; loss {float(loss.detach()):.4f}; "
          f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")


if __name__ == "__main__":
    main()
