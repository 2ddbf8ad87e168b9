"""FusedTrainer over a captioner with num_layers = 2 (models/decoderlstm.py:34-36, :65-67, :101-103; train_gru.py:114 default):
the extra GRUCells' parameters live in the arena and are updated by the fused clip + Adam.  Checked against the module-API loop
(training_step -> backward -> caphn.optim.FusedAdam with the same clip), whose pieces are pinned separately: the layered decoder
against the reference module's vectors (tests/test_gpu_layers.py), the optimiser against torch.optim.Adam / the oracle
(tests/test_gpu_module_optim.py)."""
import pytest
import torch

from helpers import maxdiff
from oracle import caphn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _V:
    w2i = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3, "factual": 4, "humorous": 5, "romantic": 6}

    def __call__(self, w):
        return self.w2i.get(w, 3)


def _net(dims, layers):
    from hypernet_attention import HyperNet
    from models.decoderlstm import AttentionGru
    torch.manual_seed(3)
    net = HyperNet(dims.F, dims.E, dims.H, dims.V, _V(), cc=True, hyper_emb=dims.he)
    net.captioner = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, num_layers=layers, p=0.0)
    return net.to(DEV)


def test_fused_trainer_with_two_layers_follows_the_module_api_loop():
    from caphn.engine import FusedTrainer
    from caphn.optim import FusedAdam
    from caphn import functional as CF
    dims = O.Dims(D=24, F=12, E=12, H=12, V=40, he=3)
    a, b = _net(dims, 2), _net(dims, 2)
    b.load_state_dict(a.state_dict())
    assert len(a.captioner.layers) == 1
    tr = FusedTrainer(a, lr=1e-3, max_norm=0.5)
    assert any(n.startswith("captioner.layers.0.") for n in tr.optimizer_param_names())
    (opt,), _ = b.configure_optimizers()
    # the reference's HyperNet list has no captioner.layers (its captioner has one layer): add them, as train_gru.py's
    # Adam(captioner.parameters()) would
    opt.add_param_group({"params": list(b.captioner.layers.parameters())})
    for g in opt.param_groups:
        g["lr"] = 1e-3
    opt.max_norm = 0.5
    xs = [torch.eye(dims.he)[i % dims.he].to(DEV) for i in range(3)]
    for i in range(3):
        bt = O.synth_batch(dims, B=3, T=6, P=5, seed=40 + i)
        f, c = bt["features"].to(DEV), bt["captions"].to(DEV)
        la = tr.step(f, c, x_style=xs[i])
        opt.zero_grad()
        cap = b.forward(xs[i])
        pred, _ = cap(f, c, 0.0)
        lb = CF.cross_entropy(pred.view(-1, dims.V), c.view(-1), ignore_index=0)
        lb.backward()
        assert b.captioner.layers[0].weight_hh.grad is not None and float(b.captioner.layers[0].weight_hh.grad.abs().sum()) > 0
        opt.step()
        assert abs(float(la[0]) - float(lb)) < 2e-5, (i, float(la[0]), float(lb))
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for n in pa:
        if n.startswith("captioner.gru."):
            continue
        tol = 3.5e-3 if n.endswith("v_a.bias") else 3e-5
        assert maxdiff(pa[n].detach(), pb[n].detach()) < tol, n
