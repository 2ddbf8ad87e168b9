"""Data-parallel path of the fused engine with TWO processes sharing the one GPU of the test box
(gloo backend moving CUDA tensors; the driver's multi-GPU bench uses the same code over RCCL).

  * identical minibatches on both ranks  ==  single-process training (averaging equal gradients; the
    rank-2 factor sum / 2 equals the rank-1 gradient)
  * different minibatches and domains: the replicas stay bit-identical after every step.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, PKG

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, same_batch, q, flickr=False):
    for p in (REPO, PKG, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import caphn_oracle as O
        from hypernet_attention import HyperNet
        from models.decoderlstm import AttentionGru
        from caphn.engine import FusedTrainer
        dims = O.Dims(D=24, F=12, E=12, H=12, V=40, he=5)
        p = O.init_params(dims, seed=3)

        class V:
            w2i = {"<pad>": 0}

            def __call__(self, w):
                return 3
        if flickr:       # bench.py's configuration: x = embed[style token], next step's theta from the Adam pass
            dims = O.Dims(D=24, F=12, E=12, H=12, V=40, he=12)
            p = O.init_params(dims, seed=3)
        net = HyperNet(dims.F, dims.E, dims.H, dims.V, V(), cc=not flickr, hyper_emb=dims.he)
        net.captioner = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, p=0.0)
        net.load_state_dict(p, strict=False)
        tr = FusedTrainer(net.to("cuda:0"), lr=1e-3, max_norm=0.5)
        seed = 100 if same_batch else 100 + rank
        batch = O.synth_batch(dims, B=3, T=6, P=5, seed=seed)
        x = torch.zeros(dims.he, device="cuda:0")
        x[1 if same_batch else rank + 1] = 1.0
        losses = []
        feats_dev, caps_dev = batch["features"].cuda(), batch["captions"].cuda()
        for _ in range(4):
            if flickr:      # bench.py's configuration: next-step theta in the Adam pass + next precompute on a side stream
                tok = 4 + rank
                l = tr.step(feats_dev, caps_dev, style_token=tok, next_style_token=tok, next_features=feats_dev)
            else:
                l = tr.step(batch["features"].cuda(), batch["captions"].cuda(), x_style=x)
            losses.append(float(l[0]))
        flat = torch.cat([tr.flat_p] + [w.data.flatten() for w in tr.W2]).cpu().numpy().copy()   # pickled by value
        q.put((rank, "ok", losses, flat))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc(), None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _run(same_batch, flickr=False):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, same_batch, q, flickr)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    res = [(r[0], r[1], r[2], torch.from_numpy(r[3]) if r[3] is not None else None) for r in res]
    for pr in procs:
        pr.join(timeout=60)
    assert all(r[1] == "ok" for r in res), [r[2] for r in res if r[1] != "ok"]
    return res


@pytest.mark.timeout(600)
def test_dp_two_ranks_different_batches_stay_in_sync():
    res = _run(same_batch=False)
    assert torch.equal(res[0][3], res[1][3])                    # bit-identical replicas
    assert res[0][2] != res[1][2]                               # the ranks saw different data
    assert res[0][2][-1] < res[0][2][0]


@pytest.mark.timeout(600)
def test_dp_flickr_style_tokens_with_prefetch():
    """bench.py's multi-GPU configuration: a different style token per rank, next theta fused into Adam."""
    res = _run(same_batch=False, flickr=True)
    assert torch.equal(res[0][3], res[1][3])
    assert res[0][2] != res[1][2] and res[0][2][-1] < res[0][2][0]


@pytest.mark.timeout(600)
def test_dp_two_ranks_same_batch_equals_single_process():
    from oracle import caphn_oracle as O
    from hypernet_attention import HyperNet
    from models.decoderlstm import AttentionGru
    from caphn.engine import FusedTrainer
    res = _run(same_batch=True)
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
    losses = [float(tr.step(batch["features"].cuda(), batch["captions"].cuda(), x_style=x)[0]) for _ in range(4)]
    flat = torch.cat([tr.flat_p] + [w.data.flatten() for w in tr.W2]).cpu()
    assert max(abs(a - b) for a, b in zip(losses, res[0][2])) < 2e-5, (losses, res[0][2])
    assert float((flat - res[0][3]).abs().max()) < 1e-4
    assert torch.equal(res[0][3], res[1][3])
