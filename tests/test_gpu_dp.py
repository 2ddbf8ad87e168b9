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


def _worker(rank, world, port, same_batch, q, flickr=False, backend="gloo", steps=4):
    for p in (REPO, PKG, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    calls = {"all_reduce": 0, "all_gather": 0, "wait": 0}
    try:
        torch.cuda.set_device(0)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from caphn import dp
        if world == 1:
            dp.FORCE = True          # CAPHN_FORCE_COLLECTIVES: a one-rank group still goes through every collective call
            orig_ar, orig_ag = dist.all_reduce, dist.all_gather_into_tensor

            class _W:               # counts Work.wait() calls on the handles the engine receives
                def __init__(self, w):
                    self.w = w

                def wait(self):
                    calls["wait"] += 1
                    return self.w.wait()

            def ar(*a, **k):
                calls["all_reduce"] += 1
                w = orig_ar(*a, **k)
                return _W(w) if w is not None else None

            def ag(*a, **k):
                calls["all_gather"] += 1
                w = orig_ag(*a, **k)
                return _W(w) if w is not None else None
            dist.all_reduce, dist.all_gather_into_tensor = ar, ag
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
        for _ in range(steps):
            if flickr:      # bench.py's configuration: next-step theta in the Adam pass + next precompute on a side stream
                tok = 4 + rank
                # (captions announced too: the split front -- G / gate GEMM beside the last rank-1 pass, packed W_hh from that pass)
                l = tr.step(feats_dev, caps_dev, style_token=tok, next_style_token=tok, next_features=feats_dev, next_captions=caps_dev)
                assert tr._pre_key is not None and tr._pre_key[-1] == 3
            else:
                l = tr.step(batch["features"].cuda(), batch["captions"].cuda(), x_style=x)
            losses.append(float(l[0]))
        flat = torch.cat([tr.flat_p] + [w.data.flatten() for w in tr.W2]).cpu().numpy().copy()   # pickled by value
        named = {n: t.detach().cpu().numpy().copy() for n, t in net.named_parameters()}
        q.put((rank, "ok", losses, flat, named, dict(calls)))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc(), None, None, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _run(same_batch, flickr=False, world=2, backend="gloo", steps=4):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, same_batch, q, flickr, backend, steps)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    res = [(r[0], r[1], r[2], torch.from_numpy(r[3]) if r[3] is not None else None, r[4], r[5]) for r in res]
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


def _single_process_reference(flickr, steps=4):
    """The same training run in this process with no process group at all."""
    from oracle import caphn_oracle as O
    from hypernet_attention import HyperNet
    from models.decoderlstm import AttentionGru
    from caphn.engine import FusedTrainer
    dims = O.Dims(D=24, F=12, E=12, H=12, V=40, he=12 if flickr else 5)
    p = O.init_params(dims, seed=3)

    class V:
        w2i = {"<pad>": 0}

        def __call__(self, w):
            return 3
    net = HyperNet(dims.F, dims.E, dims.H, dims.V, V(), cc=not flickr, hyper_emb=dims.he)
    net.captioner = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, p=0.0)
    net.load_state_dict(p, strict=False)
    tr = FusedTrainer(net.to("cuda:0"), lr=1e-3, max_norm=0.5)
    batch = O.synth_batch(dims, B=3, T=6, P=5, seed=100)
    x = torch.zeros(dims.he, device="cuda:0"); x[1] = 1.0
    f, c = batch["features"].cuda(), batch["captions"].cuda()
    losses = []
    for _ in range(steps):
        if flickr:
            losses.append(float(tr.step(f, c, style_token=4, next_style_token=4, next_features=f, next_captions=c)[0]))
        else:
            losses.append(float(tr.step(f, c, x_style=x)[0]))
    return losses, torch.cat([tr.flat_p] + [w.data.flatten() for w in tr.W2]).cpu()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("flickr", [False, True])
def test_rccl_code_path_with_a_forced_one_rank_group(flickr):
    """backend "nccl" (= RCCL) with world size 1 and dp.FORCE (CAPHN_FORCE_COLLECTIVES=1): init_process_group(device_id=...),
    all_gather_into_tensor, the bucketed asynchronous all_reduce calls issued from the communication stream behind the
    backward's milestones, and Work.wait() all execute on the GPU.  A one-rank exchange is the identity, so the run must
    reproduce a run without any process group."""
    res = _run(same_batch=True, flickr=flickr, world=1, backend="nccl")
    losses, flat = _single_process_reference(flickr)
    calls = res[0][5]
    assert calls["all_gather"] == 1 * 4 and calls["all_reduce"] == 3 * 4 and calls["wait"] == 4 * 4, calls      # 1 all-gather + 3 all-reduces per step
    # (not bit-identical in general: split-K weight gradients and the embedding scatter accumulate with fp32 atomics)
    assert max(abs(a - b) for a, b in zip(losses, res[0][2])) < 1e-5, (losses, res[0][2])
    assert float((flat - res[0][3]).abs().max()) < 2e-5


@pytest.mark.timeout(600)
def test_dp_two_ranks_different_batches_match_the_oracle_stepping_on_mean_gradients():
    """Two ranks, a different minibatch AND a different domain row each, three steps of the fused engine (factor all-gather,
    bucketed all-reduce, Gram-matrix clip norm, rank-2 Adam pass) == the oracle stepping on the rank-MEAN of its own dense
    gradients (DistributedDataParallel semantics), parameter by parameter."""
    from oracle import caphn_oracle as O
    steps = 3
    res = _run(same_batch=False, steps=steps)
    dims = O.Dims(D=24, F=12, E=12, H=12, V=40, he=5)
    p = O.init_params(dims, seed=3)
    batches = [O.synth_batch(dims, B=3, T=6, P=5, seed=100 + r) for r in range(2)]
    xs = []
    for r in range(2):
        x = torch.zeros(dims.he); x[r + 1] = 1.0
        xs.append(x)
    state = {}
    for step in range(1, steps + 1):
        gs = [O.forward_backward(dims, p, xs[r], batches[r]["features"], batches[r]["captions"])[4] for r in range(2)]
        avg = {n: (gs[0][n] + gs[1][n]) / 2 for n in O.trainable_names(p)}
        O.train_step(dims, p, state, step, xs[0], batches[0]["features"], batches[0]["captions"], lr=1e-3, max_norm=0.5,
                     grads_override=avg)
    got = res[0][4]
    bad = {}
    for n in O.trainable_names(p):
        # v_a.bias has a mathematically zero gradient (softmax is shift invariant): Adam turns its rounding noise into
        # +-lr per step, on either side
        tol = 3 * 1.1e-3 if n.endswith("v_a.bias") else 1e-5
        d = float((torch.from_numpy(got[n]).double() - p[n].double()).abs().max())
        if not d < tol:
            bad[n] = d
    assert not bad, (bad, res[0][2], res[1][2])
    assert torch.equal(res[0][3], res[1][3])
