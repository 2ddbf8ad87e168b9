"""Host-side logic of the drop-in API on CPU: the weight-injection protocol (utils.py:24-69), the
head-sizing rule (hypernet_attention.py:55-99) and loud failure without a GPU."""
import pytest
import torch
from torch import nn

from oracle import caphn_oracle as O
from helpers import TINY_DIMS, load_case


def test_flip_and_set_semantics_literal():
    import utils
    cell = nn.GRUCell(8, 4)
    order = [n for n, _ in cell.named_parameters()]
    assert order == ["weight_ih", "weight_hh", "bias_ih", "bias_hh"]
    utils.flip_parameters_to_tensors(cell)
    assert cell.registered_parameters_name == order            # SURVEY 8a H3 [measured]
    assert not list(cell.parameters())
    assert all(float(getattr(cell, n).abs().sum()) == 0 for n in order)
    n_tot = 12 * 8 + 12 * 4 + 12 + 12
    theta = torch.arange(n_tot, dtype=torch.float32).reshape(1, -1).requires_grad_(True)
    assert utils.set_all_parameters(cell, theta, detach=True) == n_tot
    assert [n for n, _ in cell.named_parameters()] == order   # re-registered as Parameters (utils.py:57-58)
    assert torch.equal(cell.weight_ih.detach().flatten(), theta[0, :96].detach())
    assert torch.equal(cell.bias_hh.detach(), theta[0, -12:].detach())
    assert cell.weight_ih.is_leaf                               # detached from theta
    # idempotent across steps
    utils.flip_parameters_to_tensors(cell)
    assert cell.registered_parameters_name == order
    assert utils.set_all_parameters(cell, theta * 2, detach=True) == n_tot
    assert torch.equal(cell.weight_hh.detach().flatten(), 2 * theta[0, 96:144].detach())
    # the cell still works as an nn.GRUCell with the injected weights (beam search calls captioner.gru(x,h))
    out = cell(torch.zeros(2, 8), torch.zeros(2, 4))
    assert out.shape == (2, 4)


def test_flip_and_set_attached_mode():
    import utils
    cell = nn.GRUCell(8, 4)
    n_tot = 12 * 8 + 12 * 4 + 24
    theta = torch.randn(1, n_tot, requires_grad=True)
    for _ in range(3):                                           # flip must keep finding the names
        utils.flip_parameters_to_tensors(cell)
        assert utils.set_all_parameters(cell, theta, detach=False) == n_tot
    assert not list(cell.parameters())
    assert not cell.weight_ih.is_leaf
    cell(torch.randn(2, 8), torch.randn(2, 4)).sum().backward()
    assert theta.grad is not None and float(theta.grad.abs().sum()) > 0
    # switching back to the literal mode re-registers Parameters
    utils.flip_parameters_to_tensors(cell)
    utils.set_all_parameters(cell, theta.detach(), detach=True)
    assert len(list(cell.parameters())) == 4


def test_set_skips_weight_and_bias_names_and_child_offset_restart():
    """utils.py:50-53 skips names exactly 'weight'/'bias'; :68 recurses with the SAME theta so a
    child restarts at offset 0 (a reference bug, only reachable through hypernet.py)."""
    import utils

    class Wrap(nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(3, 2)         # names 'weight', 'bias' -> skipped
            self.cell = nn.GRUCell(2, 2)

    m = Wrap()
    utils.flip_parameters_to_tensors(m)
    assert m.lin.registered_parameters_name == ["weight", "bias"]
    theta = torch.arange(100, dtype=torch.float32).reshape(1, -1)
    n = utils.set_all_parameters(m, theta, detach=True)
    assert n == 12 + 12 + 6 + 6
    assert float(m.lin.weight.abs().sum()) == 0                 # untouched zeros
    assert torch.equal(m.cell.weight_ih.detach().flatten(), theta[0, :12])   # child restarted at 0


@pytest.mark.parametrize("name", ["gru_tiny_flickr", "gru_tiny_cc", "gru_odd_cc"])
def test_head_rule_matches_oracle_and_golden(name):
    from hypernet_attention import build_hypernet_layers
    dims = TINY_DIMS[name]
    cell = nn.GRUCell(dims.E + dims.F, dims.H)
    base, heads = build_hypernet_layers(cell, dims.he)
    got = [(h[0].in_features, h[0].out_features, h[2].out_features) for h in heads]
    assert got == O.head_layout(dims)
    assert base[0].in_features == dims.he and base[2].out_features == dims.he
    _, p = load_case(name)
    sd = {"hn_heads." + k: v for k, v in heads.state_dict().items()}
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(p[k].shape), k


def test_head_rule_canonical_sizes():
    from hypernet_attention import build_hypernet_layers
    cell = nn.GRUCell(400, 200)
    base, heads = build_hypernet_layers(cell, 200)
    assert [(h[0].out_features, h[2].out_features) for h in heads] == [(480, 240000), (240, 120000), (200, 600), (200, 600)]
    n = sum(p.numel() for p in base.parameters()) + sum(p.numel() for p in heads.parameters())
    assert n == 144906720
    with pytest.raises(ValueError):
        build_hypernet_layers(nn.GRUCell(2, 1), 200)             # unrunnable branch 1 is rejected


def test_module_names_match_reference_state_dict():
    from models.decoderlstm import AttentionGru, GruNet, DecoderGRU, DecoderRNN
    m = AttentionGru(32, 16, 16, 16, 50)
    keys = set(m.state_dict())
    for k in ["feature_fc.0.weight", "feature_fc.2.bias", "embed.weight", "gru.weight_ih", "gru.bias_hh", "fc.weight",
              "attention.W_a.weight", "attention.U_a.bias", "attention.v_a.weight", "init_h.bias"]:
        assert k in keys
    assert sum(p.numel() for p in AttentionGru(2048, 200, 200, 200, 9684).parameters()) == 4815285
    assert issubclass(GruNet, AttentionGru)
    assert set(DecoderGRU(8, 6, 30, num_layers=2).state_dict()) == {
        'lstm_cell.weight_ih', 'lstm_cell.weight_hh', 'lstm_cell.bias_ih', 'lstm_cell.bias_hh', 'layers.0.weight_ih',
        'layers.0.weight_hh', 'layers.0.bias_ih', 'layers.0.bias_hh', 'fc_out.weight', 'fc_out.bias', 'embed.weight'}
    assert isinstance(DecoderRNN(8, 6, 30).lstm_cell, nn.LSTMCell)


def test_no_cpu_fallback():
    from caphn._lib import CaphnError
    from models.decoderlstm import AttentionGru
    from hypernet_attention import HyperNet

    class V:
        w2i = {"<pad>": 0}

        def __call__(self, w):
            return 3
    m = AttentionGru(32, 16, 16, 16, 50)
    with pytest.raises(CaphnError):
        m(torch.zeros(2, 7, 32), torch.zeros(2, 5, dtype=torch.long))
    net = HyperNet(16, 16, 16, 50, V(), cc=True, hyper_emb=4)
    with pytest.raises(CaphnError):
        net(torch.zeros(4))


def test_catr_generated_projection_head_layout_and_cpu_refusal():
    """N4 stage 1: the hypernet that generates baseline/caption.py:13's 1x1 convolution is sized by the M = 500 rule of
    hypernet_attention.py; like every other module it refuses CPU tensors instead of falling back."""
    import torch
    from baseline.caption import HyperInputProj, MLP
    from caphn._lib import CaphnError
    m = HyperInputProj(2048, 256, hyper_emb=10)
    assert [(h[0].out_features, h[2].out_features) for h in m.hn_heads] == [(1048, 524288), (10, 256)]
    assert m._shape.theta_size == 256 * 2048 + 256
    with pytest.raises(CaphnError):
        m(torch.randn(1, 2048, 2, 2), torch.randn(10))
    with pytest.raises(CaphnError):
        MLP(8, 16, 30, 3)(torch.randn(2, 8))


@pytest.mark.parametrize("name", ["catr_prenorm", "catr_postnorm"])
def test_catr_transformer_state_dict_is_the_references(name):
    """The drop-in baseline.transformer registers exactly the parameters of the reference's Transformer (names and
    shapes taken from the vectors its own classes produced), so a reference checkpoint loads with strict=True."""
    import json
    import os
    import numpy as np
    from conftest import REPO
    from baseline.transformer import build_transformer
    z = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
    with open(os.path.join(REPO, "tests", "golden", "meta.json")) as f:
        m = json.load(f)[name]

    class Cfg:
        pad_token_id = 0; layer_norm_eps = 1e-12; dropout = 0.1
    Cfg.hidden_dim, Cfg.nheads, Cfg.dim_feedforward = m["hidden_dim"], m["nheads"], m["dim_feedforward"]
    Cfg.vocab_size, Cfg.max_position_embeddings = m["vocab_size"], m["max_position_embeddings"]
    Cfg.enc_layers, Cfg.dec_layers, Cfg.pre_norm = m["enc_layers"], m["dec_layers"], m["pre_norm"]
    net = build_transformer(Cfg)
    want = {k[2:]: z[k].shape for k in z.files if k.startswith("p/")}
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == {k: tuple(v) for k, v in want.items()}
    assert sum(p.numel() for p in net.parameters()) == m["n_params"]
    with pytest.raises(Exception):                      # CPU tensors are refused, not computed
        net(torch.zeros(1, Cfg.hidden_dim, 2, 2), torch.zeros(1, 2, 2, dtype=torch.bool), torch.zeros(1, Cfg.hidden_dim, 2, 2),
            torch.zeros(1, Cfg.max_position_embeddings, dtype=torch.long), torch.zeros(1, Cfg.max_position_embeddings, dtype=torch.bool))
