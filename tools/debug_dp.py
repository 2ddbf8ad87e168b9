#!/usr/bin/env python3
"""Debug aid: two ranks on ONE GPU over gloo, identical minibatches; per trial compare the exchanged gradient arena with 2 x the local
one (region by region) and the gathered factors with the local ones.  Prints the first mismatching regions."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hypernet-image-captioning_amd")):
    sys.path.insert(0, p)


def worker(rank, world, port, trials, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import caphn_oracle as O
    from hypernet_attention import HyperNet
    from models.decoderlstm import AttentionGru
    from caphn.engine import FusedTrainer
    from caphn import dp, _lib
    _lib.load().caphn_tune(13, 40)          # deterministic gradients: local passes are bit-reproducible
    dims = O.Dims(D=24, F=12, E=12, H=12, V=40, he=5)
    p = O.init_params(dims, seed=3)

    class V:
        w2i = {"<pad>": 0}

        def __call__(self, w):
            return 3
    net = HyperNet(dims.F, dims.E, dims.H, dims.V, V(), cc=True, hyper_emb=dims.he)
    net.captioner = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, p=0.0)
    net.load_state_dict(p, strict=False)
    tr = FusedTrainer(net.to("cuda:0"), lr=1e-3, max_norm=0.5)
    batch = O.synth_batch(dims, B=3, T=6, P=5, seed=100)
    x = torch.zeros(dims.he, device="cuda:0"); x[1] = 1.0
    f, c = batch["features"].cuda(), batch["captions"].cuda()
    bad = []
    active = dp.active
    for t in range(trials):
        dp.active = lambda group=None: False
        tr.forward_backward(f, c, x_style=x)
        torch.cuda.synchronize()
        g_local = tr.flat_g.clone()
        acts_local = tr._acts.clone()
        dp.active = active
        tr.forward_backward(f, c, x_style=x)
        gfac, acts_all = tr._exchange()
        torch.cuda.synchronize()
        g = tr.flat_g
        for name, (lo, hi) in tr._buckets.items():
            if hi > lo:
                d = float((g[lo:hi] - 2 * g_local[lo:hi]).abs().max())
                if d > 1e-7:
                    # where inside the bucket?
                    idx = int((g[lo:hi] - 2 * g_local[lo:hi]).abs().argmax()) + lo
                    owner = [n for n, (o, k, _) in tr.offs.items() if o <= idx < o + k]
                    ratio = float(g[idx] / g_local[idx]) if float(g_local[idx]) != 0 else float("nan")
                    bad.append((t, name, d, owner, ratio))
        for r in range(world):
            d = float((gfac[r] - g_local[:tr.theta_size]).abs().max())
            if d > 0:
                bad.append((t, f"gfac[{r}]", d, None, None))
            d = float((acts_all[r] - acts_local).abs().max())
            if d > 0:
                bad.append((t, f"acts[{r}]", d, None, None))
        dist.barrier()
    q.put((rank, bad[:12], len(bad)))
    dist.destroy_process_group()


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, trials, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for _ in procs:
        rank, bad, n = q.get(timeout=600)
        print(f"rank {rank}: {n} mismatches in {trials} trials")
        for b in bad:
            print("   ", b)
    for pr in procs:
        pr.join(60)


if __name__ == "__main__":
    main()
