"""N4 data parallel: the CATR modules are ordinary autograd nn.Modules, so torch's DistributedDataParallel (RCCL on a
multi-GPU node; gloo moving CUDA tensors between two processes on the one GPU of the test box) averages their gradients,
and caphn.optim.FusedAdam keeps the replicas identical.  Two ranks on different minibatches == one process on the
concatenated minibatch (mean loss over twice the rows)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, PKG

pytestmark = pytest.mark.gpu


class _Backbone(torch.nn.Module):
    def forward(self, samples):
        src, pos = samples
        return [(src, torch.zeros(src.shape[0], src.shape[2], src.shape[3], dtype=torch.bool, device=src.device))], [pos]


def _model_and_data():
    from baseline.caption import Caption
    from baseline.configuration import Config
    from baseline.transformer import build_transformer
    cfg = Config(hidden_dim=32, max_position_embeddings=10, dropout=0.0, vocab_size=60, enc_layers=1, dec_layers=1,
                 dim_feedforward=48, nheads=4)
    torch.manual_seed(21)
    net = Caption(_Backbone(), build_transformer(cfg), cfg.hidden_dim, cfg.vocab_size, hyper_emb=10, in_channels=40)
    g = torch.Generator().manual_seed(5)
    data = []
    for _ in range(2):
        data.append((torch.randn(3, 40, 2, 3, generator=g), torch.randn(3, 32, 2, 3, generator=g),
                     torch.randint(1, 60, (3, 10), generator=g)))
    x = torch.zeros(10); x[2] = 1.0
    return cfg, net, data, x


def _steps(net, batches, x, cfg, ddp=None):
    import torch.nn.functional as F
    from caphn.optim import FusedAdam
    opt = FusedAdam(net.parameters(), lr=1e-3)
    model = ddp if ddp is not None else net
    losses = []
    for _ in range(3):
        opt.zero_grad()
        src, pos, tgt = batches
        out = model((src, pos), tgt, torch.zeros_like(tgt, dtype=torch.bool), x)
        loss = F.cross_entropy(out.reshape(-1, cfg.vocab_size), tgt.reshape(-1))
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses, torch.cat([p.detach().flatten() for p in net.parameters()]).cpu()


def _worker(rank, world, port, q):
    for p in (REPO, PKG, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        cfg, net, data, x = _model_and_data()
        net = net.to("cuda:0")
        ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0])
        batch = tuple(t.to("cuda:0") for t in data[rank])
        losses, flat = _steps(net, batch, x.to("cuda:0"), cfg, ddp)
        q.put((rank, "ok", losses, flat.numpy().copy()))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc(), None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_two_ranks_equal_one_process_on_the_joint_batch():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=60)
    assert all(r[1] == "ok" for r in res), [r[2] for r in res if r[1] != "ok"]
    flat0, flat1 = torch.from_numpy(res[0][3]), torch.from_numpy(res[1][3])
    assert torch.equal(flat0, flat1)                                         # replicas stay identical
    cfg, net, data, x = _model_and_data()
    net = net.to("cuda:0")
    joint = tuple(torch.cat([data[0][i], data[1][i]]).to("cuda:0") for i in range(3))
    losses, flat = _steps(net, joint, x.to("cuda:0"), cfg)
    for i in range(3):                                                       # the loss trajectory is that of the joint batch
        assert abs(losses[i] - 0.5 * (res[0][2][i] + res[1][2][i])) < 5e-5, (i, losses, res[0][2], res[1][2])
    # Adam normalises by sqrt(v): where a gradient is ~0 its last bits decide the sign of a full lr-sized move, so single
    # parameters may differ by O(lr) while the bulk agrees to rounding
    diff = (flat - flat0).abs()
    assert float(diff.max()) < 3 * 1e-3 and float(diff.mean()) < 2e-6 and float((diff < 5e-5).float().mean()) > 0.995
    assert losses[-1] < losses[0]
