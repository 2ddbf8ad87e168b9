"""N2 inside the fused engine: HyperNetCC's trainable domain front-ends (cc_train_hypernet.py:93-109, used :136-149,
optimised :110-122) as part of FusedTrainer -- parameters in the arena and the Adam pass, the hypernet's input row computed
and back-propagated by libcaphn kernels.  Against the oracle (oracle.frontend_forward + train_step): loss and EVERY gradient
of the first step at 2e-6, then three optimiser steps over changing domains with the next step's theta produced in the
Adam pass."""
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

    def __len__(self):
        return len(self.w2i)


@pytest.mark.parametrize("mode", ["embedding", "histograme", "JSD"])
def test_front_end_trains_inside_the_fused_engine(mode):
    from cc_train_hypernet import HyperNetCC
    from models.decoderlstm import AttentionGru
    from caphn.engine import FusedTrainer
    torch.manual_seed(0)
    domains = ["news\n", "sport\n", "travel\n"]
    dims = O.Dims(D=24, F=12, E=12, H=12, V=40, he=10)
    vocab = _V()
    feats_dom = None
    if mode == "histograme":
        feats_dom = {d: torch.rand(len(vocab) + 1).tolist() for d in domains}
    if mode == "JSD":
        feats_dom = {d: torch.randn(2).tolist() for d in domains}
    net = HyperNetCC(dims.F, dims.E, dims.H, dims.V, vocab, domains, lr=1e-3, hyper_emb=10, embedding=mode, domain_features=feats_dom)
    net.hypernet.captioner = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, p=0.0)
    net = net.to(DEV)
    p = {k[len("hypernet."):]: v.detach().cpu().clone() for k, v in net.state_dict().items() if k.startswith("hypernet.")}
    p = {k: v for k, v in p.items() if not k.startswith("captioner.gru.")}
    p.update({"embed." + k: v.detach().cpu().clone() for k, v in net.embed.state_dict().items()})
    tr = FusedTrainer.from_cc(net, lr=1e-3, max_norm=0.5)
    names = tr.optimizer_param_names()
    assert names[16:16 + len(list(net.embed.parameters()))] == ["embed." + n for n, _ in net.embed.named_parameters()]   # :111-113
    seq = ["sport", "news", "sport"]
    batches = [O.synth_batch(dims, B=3, T=6, P=5, seed=20 + i) for i in range(3)]

    def oinp(d):
        return net.dict_domain[d] if mode == "embedding" else torch.tensor(net.dict_domain[d], dtype=torch.float32)
    # ---- first step: loss and every gradient
    f0, c0 = batches[0]["features"].to(DEV), batches[0]["captions"].to(DEV)
    loss = tr.forward_backward(f0, c0, domain_input=net.domain_input(seq[0]), validate=True)
    ref_loss, _, _, _, g = O.forward_backward(dims, p, None, batches[0]["features"], batches[0]["captions"], frontend=(mode, oinp(seq[0])))
    assert abs(float(loss[0]) - float(ref_loss)) < 2e-6
    for n in O.trainable_names(p):
        if n.startswith("hn_heads.") and n.endswith(".2.weight"):
            got = tr.w2_grad_dense(int(n.split(".")[1]))
        else:
            got = tr.grad(n)
        want = g[n] if g[n] is not None else torch.zeros_like(p[n])
        assert maxdiff(got.cpu(), want) < 2e-6, n
    assert any(float(tr.grad(n).abs().sum()) > 0 for n in names if n.startswith("embed."))
    tr.optimizer_step(next_domain_input=net.domain_input(seq[1]))
    state = {}
    O.train_step(dims, p, state, 1, None, batches[0]["features"], batches[0]["captions"], lr=1e-3, max_norm=0.5,
                 frontend=(mode, oinp(seq[0])))
    # ---- two more steps, domains change, next theta comes out of the Adam pass
    for i in (1, 2):
        nxt = net.domain_input(seq[i + 1]) if i + 1 < len(seq) else None
        l = tr.step(batches[i]["features"].to(DEV), batches[i]["captions"].to(DEV), domain_input=net.domain_input(seq[i]),
                    next_domain_input=nxt)
        ref, *_ = O.train_step(dims, p, state, i + 1, None, batches[i]["features"], batches[i]["captions"], lr=1e-3, max_norm=0.5,
                               frontend=(mode, oinp(seq[i])))
        assert abs(float(l[0]) - float(ref)) < 2e-5, (i, float(l[0]), float(ref))
    sd = dict(net.hypernet.named_parameters())
    sd.update({"embed." + n: t for n, t in net.embed.named_parameters()})
    for n in O.trainable_names(p):
        tol = 3.5e-3 if n.endswith("v_a.bias") else 3e-5        # Adam on a mathematically zero gradient: +-lr per step
        assert maxdiff(sd[n].detach().cpu(), p[n]) < tol, n
    assert tr.step_count == 3
