"""BASELINE config 1: the call pattern of train_gru.py on its bs = 8 plumbing case.

train_gru.py:45 builds `GruNet(2048, features, embed, hidden, vocab, num_layers=num_layers, p=0.0)` (a name its own
models/decoderlstm.py never defines; AttentionGru's constructor) and uses the result of `captioner(img_feats,
caps.long(), prob)` as ONE tensor: `.view(-1, vocab_size)` into F.cross_entropy(ignore_index=<pad>) (:84-86, :104-108),
Adam over `captioner.parameters()` (:65-66).  The image encoder is out of scope (synthetic [8, 49, 2048] feature maps).
Checked against the oracle's logits / loss / gradients; the product path is libcaphn (there is no CPU backend)."""
import pytest
import torch
from torch.nn import functional as F

from helpers import maxdiff
from oracle import caphn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(seed=0, V=300):
    from models.decoderlstm import GruNet
    torch.manual_seed(seed)
    net = GruNet(2048, 64, 48, 56, V, num_layers=1, p=0.0)       # train_gru.py:45
    dims = O.Dims(D=2048, F=64, E=48, H=56, V=V, he=48)
    return net, dims


def _oracle_view(net):
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    p = {"captioner." + k: v for k, v in sd.items() if not k.startswith("gru.")}
    cellw = {k[4:]: v for k, v in sd.items() if k.startswith("gru.")}
    return p, cellw


def test_grunet_training_step_pattern_bs8():
    net, dims = _build()
    p, cellw = _oracle_view(net)
    batch = O.synth_batch(dims, B=8, T=12, P=49, seed=5)
    feats, caps = batch["features"], batch["captions"].float()       # the collate yields float captions (data_loader.py:368-383)
    net = net.to(DEV).train()
    caps_pred = net(feats.to(DEV), caps.to(DEV).long(), 0.0)                         # :85 (the 0.0 call)
    assert isinstance(caps_pred, torch.Tensor) and caps_pred.shape == (8, 12, dims.V)
    loss = F.cross_entropy(caps_pred.view(-1, dims.V), caps.to(DEV).view(-1).long(), ignore_index=0)
    loss.backward()
    # oracle
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    cw = {k: v.clone().requires_grad_(True) for k, v in cellw.items()}
    ref, _ = O.decoder_forward(dims, q, cw, feats, caps.long())
    ref_loss = O.caption_loss(ref, caps.long())
    ref_loss.backward()
    assert maxdiff(caps_pred.detach().cpu(), ref.detach()) < 5e-6
    assert abs(float(loss) - float(ref_loss)) < 2e-6
    got = dict(net.named_parameters())
    for k, v in q.items():
        assert maxdiff(got[k[len("captioner."):]].grad.cpu(), v.grad) < 5e-6, k
    for k, v in cw.items():
        assert maxdiff(got["gru." + k].grad.cpu(), v.grad) < 5e-6, k
    # token argmax agrees wherever the oracle's own top-2 margin is not a rounding tie
    top2 = ref.detach().topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2e-5
    assert torch.equal(caps_pred.detach().cpu().argmax(-1)[clear], ref.detach().argmax(-1)[clear])


def test_grunet_adam_steps_follow_the_oracle():
    """configure_optimizers (:65-66): torch.optim.Adam over captioner.parameters(); three steps of the call pattern."""
    net, dims = _build(seed=1)
    p, cellw = _oracle_view(net)
    batch = O.synth_batch(dims, B=8, T=10, P=49, seed=6)
    feats, caps = batch["features"], batch["captions"]
    net = net.to(DEV).train()
    opt = torch.optim.Adam(list(net.parameters()), lr=5e-3)                          # lr of train_gru.py:115
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    cw = {k: v.clone().requires_grad_(True) for k, v in cellw.items()}
    ropt = torch.optim.Adam(list(q.values()) + list(cw.values()), lr=5e-3)
    for step in range(3):
        opt.zero_grad()
        out = net(feats.to(DEV), caps.to(DEV), 0.0)
        loss = F.cross_entropy(out.view(-1, dims.V), caps.to(DEV).view(-1), ignore_index=0)
        loss.backward()
        ropt.zero_grad()
        ref, _ = O.decoder_forward(dims, q, cw, feats, caps)
        rl = O.caption_loss(ref, caps)
        rl.backward()
        assert abs(float(loss) - float(rl)) < 1e-5 * (step + 1), step
        # identical gradients into both optimisers would hide nothing here: the two paths run their own Adam on their own
        # gradients, so only the loss trajectory is compared tightly (Adam amplifies rounding noise of ~0 gradients)
        opt.step(); ropt.step()
    assert float(loss) < 5.8


def test_grunet_validation_step_pattern():
    """validation_step (:95-97) runs the 0.0 call without gradients; the 1.0 (free-running) call of training_step (:84)
    returns a single tensor too."""
    net, dims = _build(seed=2)
    p, cellw = _oracle_view(net)
    batch = O.synth_batch(dims, B=8, T=9, P=49, seed=7)
    feats, caps = batch["features"], batch["captions"]
    net = net.to(DEV).eval()
    with torch.no_grad():
        a = net(feats.to(DEV), caps.to(DEV), 0.0)
        b = net(feats.to(DEV), caps.to(DEV), 1.0)
    ra, _ = O.decoder_forward(dims, p, cellw, feats, caps)
    rb, _ = O.decoder_forward(dims, p, cellw, feats, caps, use_sampling=[True] * 9)
    assert maxdiff(a.cpu(), ra) < 5e-6
    assert maxdiff(b.cpu(), rb) < 2e-5 and torch.equal(b.cpu().argmax(-1), rb.argmax(-1))


@pytest.mark.parametrize("T", [9, 12])
def test_grunet_literal_training_step_trains_through_the_sampled_call(T):
    """train_gru.py:84-86 as written: BOTH calls inside training_step carry gradients --
        loss = 0.5 * CE(captioner(feats, caps, 1.0)) + 0.5 * CE(captioner(feats, caps, 0.0))
    The 1.0 call feeds back argmax(output / 0.5) at every step t >= 1; the argmax is not differentiable, so its gradient is
    the teacher-forced one over the sampled ids.  Against the oracle's autograd through the same loop."""
    net, dims = _build(seed=3)
    p, cellw = _oracle_view(net)
    batch = O.synth_batch(dims, B=8, T=T, P=49, seed=8)
    feats, caps = batch["features"], batch["captions"]
    net = net.to(DEV).train()
    tf = 0.5                                                                         # self.teacher_forcing_proba, :40
    a = net(feats.to(DEV), caps.to(DEV).long(), 1.0)
    b = net(feats.to(DEV), caps.to(DEV).long(), 0.0)
    loss = tf * F.cross_entropy(a.view(-1, dims.V), caps.to(DEV).view(-1), ignore_index=0) + \
        (1 - tf) * F.cross_entropy(b.view(-1, dims.V), caps.to(DEV).view(-1), ignore_index=0)
    loss.backward()
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    cw = {k: v.clone().requires_grad_(True) for k, v in cellw.items()}
    ra, _ = O.decoder_forward(dims, q, cw, feats, caps, use_sampling=[True] * T)
    rb, _ = O.decoder_forward(dims, q, cw, feats, caps)
    rl = tf * O.caption_loss(ra, caps) + (1 - tf) * O.caption_loss(rb, caps)
    rl.backward()
    assert torch.equal(a.detach().cpu().argmax(-1), ra.detach().argmax(-1))            # the sampled trajectories coincide
    assert maxdiff(a.detach().cpu(), ra.detach()) < 2e-5 and maxdiff(b.detach().cpu(), rb.detach()) < 5e-6
    assert abs(float(loss) - float(rl)) < 5e-6
    got = dict(net.named_parameters())
    for k, v in q.items():
        assert maxdiff(got[k[len("captioner."):]].grad.cpu(), v.grad) < 1e-5, k
    for k, v in cw.items():
        assert maxdiff(got["gru." + k].grad.cpu(), v.grad) < 1e-5, k


def test_grunet_two_layers_bs8():
    """train_gru.py's own --num_layers default is 2 (:114): GruNet(..., num_layers=2) adds one GRUCell applied as h = layer(h, h)
    to the initial state and at every step (models/decoderlstm.py:34-36, :65-67, :101-103)."""
    from models.decoderlstm import GruNet
    torch.manual_seed(3)
    V = 300
    net = GruNet(2048, 64, 48, 56, V, num_layers=2, p=0.0)
    dims = O.Dims(D=2048, F=64, E=48, H=56, V=V, he=48)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    p = {"captioner." + k: v for k, v in sd.items() if not k.startswith("gru.") and not k.startswith("layers.")}
    cellw = {k[4:]: v for k, v in sd.items() if k.startswith("gru.")}
    layers = [{k[len("layers.0."):]: v for k, v in sd.items() if k.startswith("layers.0.")}]
    batch = O.synth_batch(dims, B=8, T=12, P=49, seed=7)
    feats, caps = batch["features"], batch["captions"]
    net = net.to(DEV).train()
    out = net(feats.to(DEV), caps.to(DEV), 0.0)
    assert isinstance(out, torch.Tensor) and out.shape == (8, 12, V)
    loss = F.cross_entropy(out.view(-1, V), caps.to(DEV).view(-1), ignore_index=0)
    loss.backward()
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    cw = {k: v.clone().requires_grad_(True) for k, v in cellw.items()}
    lw = [{k: v.clone().requires_grad_(True) for k, v in layers[0].items()}]
    ref, _ = O.decoder_forward(dims, q, cw, feats, caps, layers=lw)
    ref_loss = O.caption_loss(ref, caps)
    ref_loss.backward()
    assert maxdiff(out.detach().cpu(), ref.detach()) < 5e-6 and abs(float(loss) - float(ref_loss)) < 2e-6
    got = dict(net.named_parameters())
    for k, v in q.items():
        assert maxdiff(got[k[len("captioner."):]].grad.cpu(), v.grad) < 5e-6, k
    for k, v in cw.items():
        assert maxdiff(got["gru." + k].grad.cpu(), v.grad) < 5e-6, k
    for k, v in lw[0].items():
        assert maxdiff(got["layers.0." + k].grad.cpu(), v.grad) < 5e-6, k
