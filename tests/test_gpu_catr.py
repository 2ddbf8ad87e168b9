"""N4, stage 1 (parity UNPINNED by the reference: it has no hypernet for CATR): the hypernet-generated 1x1 input
projection and the MLP vocabulary head of baseline/caption.py through libcaphn, against torch's conv2d / linear on
the same weights and torch autograd through an fp64 restatement of the hypernetwork (oracle.hyper_forward)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import caphn_oracle as O
from helpers import maxdiff

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref_generate(mod, x):
    p = {k: v.detach().double().cpu().requires_grad_(True) for k, v in mod.hyper_named_tensors().items()}
    theta = O.hyper_forward(p, x.double().cpu().reshape(1, -1), n_heads=len(mod.hn_heads))
    return p, theta.reshape(-1)


@pytest.mark.parametrize("cin,hid,he,bs,h,w", [(48, 16, 10, 3, 5, 4), (2048, 256, 10, 2, 7, 7), (2048, 256, 200, 2, 3, 3)])
def test_generated_projection_matches_conv2d(cin, hid, he, bs, h, w):
    from baseline.caption import HyperInputProj
    torch.manual_seed(cin + he)
    mod = HyperInputProj(cin, hid, he).to(DEV)
    src = torch.randn(bs, cin, h, w, device=DEV, requires_grad=True)
    x = torch.randn(he, device=DEV)
    out = mod(src, x)
    assert out.shape == (bs, hid, h, w)
    p, theta = _ref_generate(mod, x)
    wr, br = theta[:hid * cin].view(hid, cin, 1, 1), theta[hid * cin:]
    s64 = src.detach().double().cpu().requires_grad_(True)
    ref = F.conv2d(s64, wr, br)
    scale = float(ref.detach().abs().max())
    assert maxdiff(out.detach().cpu().double(), ref.detach()) < 2e-6 * max(1.0, scale)
    g = torch.randn(bs, hid, h, w, dtype=torch.float64)
    out.backward(g.float().to(DEV))
    ref.backward(g)
    assert maxdiff(src.grad.cpu().double(), s64.grad) < 2e-6 * max(1.0, float(s64.grad.abs().max()))
    for name, t in mod.hyper_named_tensors().items():
        want = p[name].grad
        tol = 3e-6 * max(1.0, float(want.abs().max()))
        if t.numel() > 4_000_000:                      # the big second-layer weight: rows sampled, norm in full
            rows = torch.arange(0, t.shape[0], max(1, t.shape[0] // 97))
            assert maxdiff(t.grad[rows.to(DEV)].cpu().double(), want[rows]) < tol, name
            assert abs(float(t.grad.double().norm()) - float(want.norm())) < 1e-5 * float(want.norm()) + 1e-9, name
        else:
            assert maxdiff(t.grad.cpu().double(), want) < tol, name


def test_mlp_head_matches_torch():
    from baseline.caption import MLP
    torch.manual_seed(5)
    mlp = MLP(64, 96, 301, 3).to(DEV)
    x = torch.randn(7, 5, 64, device=DEV, requires_grad=True)
    y = mlp(x)
    ref_p = [p.detach().double().cpu().requires_grad_(True) for p in mlp.parameters()]
    x64 = x.detach().double().cpu().requires_grad_(True)
    hcur = x64
    for i in range(3):
        hcur = F.linear(hcur, ref_p[2 * i], ref_p[2 * i + 1])
        if i < 2:
            hcur = F.relu(hcur)
    assert maxdiff(y.detach().cpu().double(), hcur.detach()) < 2e-6
    g = torch.randn_like(hcur)
    y.backward(g.float().to(DEV))
    hcur.backward(g)
    assert maxdiff(x.grad.cpu().double(), x64.grad) < 2e-6
    for p, r in zip(mlp.parameters(), ref_p):
        assert maxdiff(p.grad.cpu().double(), r.grad) < 5e-6


def test_caption_wiring_and_no_cpu_path():
    from baseline.caption import Caption, HyperInputProj
    from caphn._lib import CaphnError

    class Backbone(torch.nn.Module):
        def forward(self, samples):
            return [(samples, torch.zeros(samples.shape[0], samples.shape[2], samples.shape[3], dtype=torch.bool,
                                          device=samples.device))], [None]

    class Body(torch.nn.Module):             # stands where baseline/transformer.py's Transformer goes: [T, bs, hidden]
        def forward(self, src, mask, pos, tgt, tgt_mask):
            return src.flatten(2).mean(-1).unsqueeze(0).expand(tgt.shape[1], -1, -1)

    torch.manual_seed(0)
    net = Caption(Backbone(), Body(), 16, 50, hyper_emb=10, in_channels=32).to(DEV)
    out = net(torch.randn(2, 32, 3, 3, device=DEV), torch.zeros(2, 6, dtype=torch.long, device=DEV), None,
              torch.randn(10, device=DEV))
    assert out.shape == (2, 6, 50)
    out.sum().backward()
    assert all(p.grad is not None for p in net.input_proj.parameters()) and net.mlp.layers[0].weight.grad is not None
    with pytest.raises(CaphnError):
        HyperInputProj(16, 12, 10)(torch.randn(1, 16, 2, 2), torch.randn(10))


@pytest.mark.parametrize("name", ["catr_prenorm", "catr_postnorm"])
def test_transformer_matches_reference_vectors(name):
    """PINNED: vectors from the reference's own baseline/transformer.py (fp64, dropout 0; tools/make_golden.py
    --only-catr).  The reference's state_dict loads with strict=True; output and every gradient are compared."""
    import json
    import os
    import numpy as np
    from helpers import GOLDEN
    from baseline.transformer import build_transformer
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        m = json.load(f)[name]

    class Cfg:
        pad_token_id = 0; layer_norm_eps = 1e-12; dropout = 0.0
    Cfg.hidden_dim, Cfg.nheads, Cfg.dim_feedforward = m["hidden_dim"], m["nheads"], m["dim_feedforward"]
    Cfg.vocab_size, Cfg.max_position_embeddings = m["vocab_size"], m["max_position_embeddings"]
    Cfg.enc_layers, Cfg.dec_layers, Cfg.pre_norm = m["enc_layers"], m["dec_layers"], m["pre_norm"]
    net = build_transformer(Cfg)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p/")}
    net.load_state_dict(sd, strict=True)
    assert sum(p.numel() for p in net.parameters()) == m["n_params"]
    net = net.to(DEV)
    src = torch.from_numpy(z["src"]).to(DEV).requires_grad_(True)
    hs = net(src, torch.from_numpy(z["mask"]).to(DEV), torch.from_numpy(z["pos"]).to(DEV), torch.from_numpy(z["tgt"]).to(DEV),
             torch.from_numpy(z["tgt_mask"]).to(DEV))
    want = torch.from_numpy(z["hs"])
    assert hs.shape == want.shape
    assert maxdiff(hs.detach().cpu().double(), want.double()) < 2e-5
    (hs * torch.from_numpy(z["R"]).to(DEV)).sum().backward()
    assert maxdiff(src.grad.cpu().double(), torch.from_numpy(z["dsrc"]).double()) < 5e-5
    worst = 0.0
    for k, p in net.named_parameters():
        g = torch.from_numpy(z["g/" + k]).double()
        got = p.grad.cpu().double() if p.grad is not None else torch.zeros_like(g)
        tol = 5e-5 * max(1.0, float(g.abs().max()))
        assert maxdiff(got, g) < tol, (k, maxdiff(got, g), tol)
        worst = max(worst, maxdiff(got, g))
    # training mode with dropout: masks come from libcaphn's hash (not torch's stream) -- finite, different from eval,
    # repeatable under torch.manual_seed
    net.train()
    for mod in net.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.1
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.1
    args = (src.detach(), torch.from_numpy(z["mask"]).to(DEV), torch.from_numpy(z["pos"]).to(DEV),
            torch.from_numpy(z["tgt"]).to(DEV), torch.from_numpy(z["tgt_mask"]).to(DEV))
    from caphn import functional as CF
    torch.manual_seed(1); CF._seed_state["n"] = 0
    a = net(*args)
    torch.manual_seed(1); CF._seed_state["n"] = 0
    b = net(*args)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    assert maxdiff(a.detach().cpu().double(), want.double()) > 1e-3
    a.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)


@pytest.mark.parametrize("max_norm", [None, 0.1])
def test_fused_adam_matches_torch_adam(max_norm):
    """caphn.optim.FusedAdam (cc_train_catr.py:56-61: Adam over every parameter) against torch.optim.Adam
    (+ clip_grad_norm_) on the same gradients, three steps."""
    from caphn.optim import FusedAdam
    g = torch.Generator().manual_seed(9)
    shapes = [(37, 5), (300,), (64, 64), (1,), (2049,)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    opt_a, opt_b = FusedAdam(pa, lr=1e-2, max_norm=max_norm), torch.optim.Adam(pb, lr=1e-2)
    for _ in range(3):
        grads = [torch.randn(s, generator=g).to(DEV) * 0.05 for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        pa[3].grad = None; pb[3].grad = None                   # a parameter without gradient is skipped by both
        if max_norm is not None:
            torch.nn.utils.clip_grad_norm_(pb, max_norm)
        opt_a.step(); opt_b.step()
    for p, q in zip(pa, pb):
        assert maxdiff(p.detach().cpu(), q.detach().cpu()) < 2e-6


def test_cross_entropy_node_matches_torch():
    """caphn.functional.cross_entropy as an autograd node (loss + d logits in one pass, upstream scalar applied on the
    device) against F.cross_entropy, with and without an ignored class, under a non-unit upstream gradient."""
    from caphn import functional as CF
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(6, 11, 301, generator=g)
    tgt = torch.randint(0, 301, (6, 11), generator=g)
    for ignore in (-100, 0):
        a = logits.to(DEV).requires_grad_(True)
        b = logits.double().requires_grad_(True)
        la = CF.cross_entropy(a.view(-1, 301), tgt.to(DEV).view(-1), ignore)
        lb = F.cross_entropy(b.view(-1, 301), tgt.view(-1), ignore_index=ignore)
        assert abs(float(la) - float(lb)) < 2e-6
        (la * 2.5).backward(); (lb * 2.5).backward()
        assert maxdiff(a.grad.cpu().double(), b.grad) < 1e-7


def test_rank1_optimiser_path_equals_dense():
    """FusedAdam.adopt_rank1(HyperInputProj): the generated projection's second-layer weights are updated from the factors
    (dtheta_i, a_i) without a dense gradient; two steps equal the dense path."""
    from baseline.caption import HyperInputProj
    from caphn.optim import FusedAdam

    def run(rank1):
        torch.manual_seed(8)
        mod = HyperInputProj(96, 32, 10).to(DEV)
        opt = FusedAdam(mod.parameters(), lr=1e-2)
        if rank1:
            opt.adopt_rank1(mod)
        g = torch.Generator(device=DEV).manual_seed(2)
        for _ in range(2):
            src = torch.randn(3, 96, 4, 4, device=DEV, generator=g)
            x = torch.randn(10, device=DEV, generator=g)
            opt.zero_grad()
            (mod(src, x) * torch.randn(3, 32, 4, 4, device=DEV, generator=g)).sum().backward()
            if rank1:
                assert mod.hn_heads[0][2].weight.grad is None and mod.hn_heads[0][0].weight.grad is not None
            opt.step()
        return [p.detach().clone() for p in mod.parameters()]

    for a, b in zip(run(True), run(False)):
        assert maxdiff(a.cpu(), b.cpu()) < 2e-6
