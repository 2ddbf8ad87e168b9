"""Data-parallel exchange (caphn.dp) with two processes on CPU over gloo.

Each rank runs the oracle forward/backward on ITS OWN minibatch and ITS OWN domain row, then
exchanges gradients the way the fused engine does: all-gather of the rank-1 factors (dtheta, hypernet
activations) + one all-reduce of the flat dense buffer.  Checked against dense DDP averaging:
  * sum_r dtheta_r (x) a_r / R  ==  mean over ranks of the dense second-layer weight gradients
  * the Gram identity for the global gradient norm
  * after clip + Adam with the averaged gradients both ranks hold identical parameters.
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, PKG


def _worker(rank, world, port, q):
    for p in (REPO, PKG, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import caphn_oracle as O
        from caphn import dp
        dims = O.Dims(D=24, F=12, E=12, H=12, V=40, he=5)
        p = O.init_params(dims, seed=3)                      # replicated parameters
        batch = O.synth_batch(dims, B=3, T=6, P=5, seed=100 + rank)
        x = torch.zeros(dims.he); x[rank + 1] = 1.0          # one domain per rank-batch
        loss, _, _, _, g = O.forward_backward(dims, p, x, batch["features"], batch["captions"])
        heads = O.head_layout(dims)
        # ---- what the engine exchanges
        with torch.no_grad():
            base = torch.nn.functional.leaky_relu(torch.nn.functional.linear(
                torch.nn.functional.leaky_relu(torch.nn.functional.linear(x, p["hn_base.0.weight"], p["hn_base.0.bias"]), 0.01),
                p["hn_base.2.weight"], p["hn_base.2.bias"]), 0.01)
            acts = [torch.nn.functional.leaky_relu(torch.nn.functional.linear(
                base, p[f"hn_heads.{i}.0.weight"], p[f"hn_heads.{i}.0.bias"]), 0.01) for i in range(4)]
        pack = torch.cat([g["dtheta"]] + acts)
        allp = dp.all_gather_factors(pack)                    # [R, L]
        assert allp.shape == (world, pack.numel()) and torch.equal(allp[rank], pack)
        dense_names = [n for n in O.trainable_names(p) if not (n.startswith("hn_heads.") and n.endswith(".2.weight"))]
        flat = torch.cat([g[n].flatten() for n in dense_names])
        work = dp.all_reduce_dense(flat, async_op=True)
        work.wait()
        # ---- reference: dense DDP averaging of every gradient
        avg = {}
        for n in O.trainable_names(p):
            t = g[n].clone()
            dist.all_reduce(t)
            avg[n] = t / world
        # dense buffer
        off = 0
        for n in dense_names:
            k = g[n].numel()
            assert torch.allclose(flat[off:off + k].view_as(g[n]) / world, avg[n], atol=1e-7), n
            off += k
        # rank-1 factors reproduce the averaged second-layer weight gradients
        th_off, a_off, gram = 0, dims.theta_size, 0.0
        for i, (_, k, w) in enumerate(heads):
            gf = allp[:, th_off:th_off + w]
            af = allp[:, a_off:a_off + k]
            dense = torch.einsum("rm,rk->mk", gf, af) / world
            assert torch.allclose(dense, avg[f"hn_heads.{i}.2.weight"], atol=1e-7), i
            gram += float(((gf.double() @ gf.double().t()) * (af.double() @ af.double().t())).sum()) / world ** 2
            th_off += w; a_off += k
        dense_sq = sum(float((avg[n].double() ** 2).sum()) for n in dense_names)
        tot_ref, coef = O.clip_coef([avg[n] for n in O.trainable_names(p)], 0.5)
        assert abs((dense_sq + gram) ** 0.5 - tot_ref) < 1e-6 * tot_ref
        # ---- replicas stay identical after the update
        state = {}
        O.train_step(dims, p, state, 1, x, batch["features"], batch["captions"], lr=1e-3, max_norm=0.5, grads_override=avg)
        chk = torch.cat([p[n].flatten() for n in O.trainable_names(p)])
        both = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(both, chk)
        assert torch.equal(both[0], both[1])
        q.put((rank, "ok", float(loss)))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp_exchange_world2_gloo():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=240) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
    assert res[0][2] != res[1][2]            # the ranks really saw different batches


def test_single_process_is_identity():
    from caphn import dp
    t = torch.arange(6.0)
    assert dp.world() == 1
    assert torch.equal(dp.all_gather_factors(t)[0], t)
    assert dp.all_reduce_dense(t) is None
