"""GPU parity of the drop-in module API (models.decoderlstm / hypernet_attention / utils) and of the
fused training engine against the golden vectors produced by the reference's modules."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import caphn_oracle as O
from helpers import TINY_DIMS, load_case, maxdiff, style_args

pytestmark = pytest.mark.gpu
DEV = "cuda"
CASES = ["gru_tiny_flickr", "gru_tiny_cc", "gru_odd_cc"]


class _Vocab:
    w2i = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3, "factual": 4, "humorous": 5, "romantic": 6}

    def __call__(self, w):
        return self.w2i.get(w, 3)


def build_net(dims, p, cc):
    from hypernet_attention import HyperNet
    from models.decoderlstm import AttentionGru
    net = HyperNet(dims.F, dims.E, dims.H, dims.V, _Vocab(), cc=cc, hyper_emb=dims.he)
    net.captioner = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, p=0.0)   # tiny D instead of 2048
    res = net.load_state_dict(p, strict=False)
    assert all(k.startswith("captioner.gru.") for k in res.missing_keys), res
    assert not res.unexpected_keys, res
    return net.to(DEV)


@pytest.mark.parametrize("name", CASES)
def test_module_api_forward_backward(name):
    from caphn import config
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    net = build_net(dims, p, cc=tok is None)
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV)
    config.DETACH_THETA = False
    xs = net.captioner.embed(torch.tensor([tok], device=DEV)) if tok is not None else x.to(DEV)
    captioner = net.forward(xs)
    assert captioner is net.captioner
    assert net.captioner.gru.registered_parameters_name == ["weight_ih", "weight_hh", "bias_ih", "bias_hh"]
    logits, alphas = captioner(feats, caps.long(), 0.0)
    assert logits.shape == g["logits"].shape and alphas.shape == g["alphas"].shape
    assert maxdiff(logits.detach().cpu(), g["logits"]) < 2e-6
    assert maxdiff(alphas.detach().cpu(), g["alphas"]) < 1e-6
    loss = F.cross_entropy(logits.view(-1, dims.V), caps.view(-1).long(), ignore_index=0)
    loss.backward()
    sd = dict(net.named_parameters())
    for k, v in g.items():
        if k.startswith("gint/"):
            assert maxdiff(sd[k[5:]].grad.cpu(), v) < 2e-6, k
        elif k.startswith("glit/") and not (tok is not None and k.endswith("embed.weight")):
            assert maxdiff(sd[k[5:]].grad.cpu(), v) < 2e-6, k
    # second forward: injection is idempotent (flip finds the attached tensors)
    net.forward(xs.detach())
    l2, _ = net.captioner(feats, caps.long(), 0.0)
    assert maxdiff(l2.detach().cpu(), g["logits"]) < 2e-6


@pytest.mark.parametrize("name", CASES[:2])
def test_module_api_literal_detached(name):
    """DETACH_THETA=True reproduces utils.py:57: generated weights are fresh leaf Parameters, the
    hypernet gets no gradient, captioner.gru.weight_ih.grad holds dL/dtheta."""
    from caphn import config
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    net = build_net(dims, p, cc=tok is None)
    config.DETACH_THETA = True
    try:
        xs = net.captioner.embed(torch.tensor([tok], device=DEV)) if tok is not None else x.to(DEV)
        net.forward(xs)
        gru = net.captioner.gru
        assert all(isinstance(getattr(gru, n), torch.nn.Parameter) for n in gru.registered_parameters_name)
        logits, _ = net.captioner(g["features"].to(DEV), g["captions"].to(DEV), 0.0)
        F.cross_entropy(logits.view(-1, dims.V), g["captions"].to(DEV).view(-1), ignore_index=0).backward()
        assert all(q.grad is None for q in list(net.hn_base.parameters()) + list(net.hn_heads.parameters()))
        dth = torch.cat([getattr(gru, n).grad.flatten() for n in gru.registered_parameters_name])
        assert maxdiff(dth.cpu(), g["dtheta"]) < 2e-6
        assert maxdiff(net.captioner.embed.weight.grad.cpu(), g["glit/captioner.embed.weight"]) < 2e-6
        net.forward(xs)          # re-registration keeps working step after step
        assert "weight_ih" in dict(gru.named_parameters())
    finally:
        config.DETACH_THETA = False


def test_module_errors():
    from caphn._lib import CaphnError
    from models.decoderlstm import AttentionGru
    m = AttentionGru(32, 16, 16, 16, 50)
    with pytest.raises(CaphnError):
        m(torch.zeros(2, 7, 32), torch.zeros(2, 5, dtype=torch.long))       # CPU tensors: no fallback
    m = m.to(DEV)
    out, _ = m(torch.zeros(2, 7, 32, device=DEV), torch.zeros(2, 5, dtype=torch.long, device=DEV), 1.0)
    assert out.requires_grad                     # the sampling path is a differentiable node (train_gru.py:84 trains through it)
    with pytest.raises(CaphnError):
        m(torch.zeros(2, 7, 31, device=DEV), torch.zeros(2, 5, dtype=torch.long, device=DEV))
    with pytest.raises(IndexError):
        m(torch.zeros(2, 7, 32, device=DEV), torch.full((2, 5), 50, dtype=torch.long, device=DEV))


@pytest.mark.parametrize("name", CASES)
def test_fused_engine_step(name):
    from caphn.engine import FusedTrainer
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    net = build_net(dims, p, cc=tok is None)
    max_norm = float(g["clip_max_norm"])
    tr = FusedTrainer(net, lr=1e-3, max_norm=max_norm)
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV)
    loss = tr.forward_backward(feats, caps, x_style=None if tok is not None else x.to(DEV), style_token=tok,
                               validate=True)
    loss0 = float(loss[0])                      # the engine reuses its loss buffer
    assert abs(loss0 - float(g["loss"])) < 2e-6
    assert maxdiff(tr.flat_g[:tr.theta_size].cpu(), g["dtheta"]) < 2e-6
    mine = {}
    for k, v in g.items():
        if k.startswith(("gint/", "glit/")):
            nm = k[5:]
            if k.startswith("glit/") and tok is not None and nm == "captioner.embed.weight":
                continue
            if nm.endswith(".2.weight") and nm.startswith("hn_heads."):
                got = tr.w2_grad_dense(int(nm.split(".")[1]))
            else:
                got = tr.grad(nm)
            assert maxdiff(got.cpu(), v) < 2e-6, k
            mine[nm] = got.cpu().clone()
    # clip + Adam: the oracle's optimiser fed with the engine's own gradients must land on the same
    # parameters (Adam's first step amplifies rounding noise on ~0 gradients, so identical inputs)
    p2 = {k: v.clone() for k, v in p.items()}
    _, tot, _, _ = O.train_step(dims, p2, {}, 1, x, g["features"], g["captions"], lr=1e-3, max_norm=max_norm,
                                style_token=tok, grads_override=mine)
    coef = tr.optimizer_step()
    assert abs(float(coef[1]) - tot) < 1e-5 * max(tot, 1.0)
    sd = dict(net.named_parameters())
    for n in O.trainable_names(p):
        assert maxdiff(sd[n].detach().cpu(), p2[n]) < 2e-7, n
    # and close to the reference's own optimiser step (looser: see above)
    for k, v in g.items():
        if k.startswith("padam/"):
            tol = 2.1e-3 if k.endswith("v_a.bias") else 1e-5
            assert maxdiff(sd[k[6:]].detach().cpu(), v) < tol, k
    # a second step runs and changes the loss
    l2 = tr.step(feats, caps, x_style=None if tok is not None else x.to(DEV), style_token=tok)
    assert float(l2[0]) != loss0


def test_engine_tracks_swapped_submodule():
    """captioner.embed = captioner.embed.from_pretrained(...) (hypernet_attention.py:108) must be honoured."""
    from caphn.engine import FusedTrainer
    name = "gru_tiny_cc"
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, _ = style_args(g)
    net = build_net(dims, p, cc=True)
    tr = FusedTrainer(net)
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV)
    l0 = float(tr.forward_backward(feats, caps, x_style=x.to(DEV))[0])
    new = torch.randn(dims.V, dims.E, generator=torch.Generator().manual_seed(0))
    net.captioner.embed = net.captioner.embed.from_pretrained(new.to(DEV), freeze=False)
    l1 = float(tr.forward_backward(feats, caps, x_style=x.to(DEV))[0])
    p2 = dict(p); p2["captioner.embed.weight"] = new
    ref, *_ = O.forward_backward(dims, p2, x, g["features"], g["captions"])
    assert abs(l1 - float(ref)) < 2e-6 and abs(l0 - float(g["loss"])) < 2e-6


def test_graph_replayed_step_matches_eager():
    """step_graphed (hipGraph replay, device-side Adam scalars) follows the same trajectory as
    eagerly launched steps."""
    from caphn.engine import FusedTrainer
    name = "gru_tiny_cc"
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, _ = style_args(g)
    feats, caps, xs = g["features"].to(DEV), g["captions"].to(DEV), x.to(DEV)
    ta = FusedTrainer(build_net(dims, p, cc=True), lr=1e-3)
    tb = FusedTrainer(build_net(dims, p, cc=True), lr=1e-3)
    la = [float(ta.step(feats, caps, x_style=xs)[0]) for _ in range(5)]
    lb = []
    for _ in range(5):
        lb.append(float(tb.step_graphed(feats, caps, x_style=xs)[0]))
    assert len(tb._graphs) == 1 and tb.step_count == 5
    assert la[0] == lb[0]
    assert max(abs(a - b) for a, b in zip(la, lb)) < 2e-5, (la, lb)
    assert la[-1] < la[0]
    # (two runs with fp32-atomic gradient sums: Adam turns last-bit noise on near-zero gradients into visible steps of up to lr
    #  each; tests/test_gpu_determinism.py makes the same comparison bit-exact with the atomics switched off)
    assert maxdiff(ta.flat_p.cpu(), tb.flat_p.cpu()) < 5e-4


def test_graph_capture_after_eager_steps_that_left_side_work_pending():
    """Regression for the capture hazards ADVICE named: an eager step that announced the next minibatch leaves a side-stream
    precompute (an event the next forward would wait on) and a prefetched theta pending; step_graphed must drain both before it
    captures (a capturing stream may not wait on uncaptured work), and the composites' forked branches must all be joined when
    the capture ends.  The replayed trajectory equals eager stepping."""
    from caphn.engine import FusedTrainer
    name = "gru_tiny_flickr"
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    _, tok = style_args(g)
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV)
    ta = FusedTrainer(build_net(dims, p, cc=False), lr=1e-3)
    tb = FusedTrainer(build_net(dims, p, cc=False), lr=1e-3)
    la = [float(ta.step(feats, caps, style_token=tok)[0]) for _ in range(5)]
    lb = [float(tb.step_graphed(feats, caps, style_token=tok)[0])]          # first sight of these buffers: runs eagerly
    lb.append(float(tb.step(feats, caps, style_token=tok, next_style_token=tok, next_features=feats, next_captions=caps)[0]))
    assert tb._pre_key is not None and tb._next_key is not None            # side work IS pending when the capture starts
    for _ in range(3):
        lb.append(float(tb.step_graphed(feats, caps, style_token=tok)[0]))  # capture + replay, then two replays
    assert len(tb._graphs) == 1 and tb.step_count == 5 and tb._pre_key is None
    assert max(abs(a - b) for a, b in zip(la, lb)) < 2e-5, (la, lb)
    assert maxdiff(ta.flat_p.cpu(), tb.flat_p.cpu()) < 1e-4


def test_hypernet_lstm_module_and_engine():
    """HyperNet(cell='lstm'): hypernet generates the LSTMCell weights of an AttentionLstm behind a
    feature_fc.  Module API gradients and one fused engine step against the oracle."""
    from hypernet_attention import HyperNet
    from models.decoderlstm import AttentionLstm
    from caphn.engine import FusedTrainer
    dims = O.Dims(D=24, F=12, E=10, H=12, V=50, he=6, cell="lstm")
    p = O.init_params(dims, seed=15)
    batch = O.synth_batch(dims, B=4, T=7, P=6, seed=16)
    x = torch.zeros(dims.he); x[1] = 1.0
    loss_ref, logits_ref, _, _, gref = O.forward_backward(dims, p, x, batch["features"], batch["captions"])

    def build():
        net = HyperNet(dims.F, dims.E, dims.H, dims.V, _Vocab(), cc=True, hyper_emb=dims.he, cell="lstm")
        net.captioner = AttentionLstm(dims.D, dims.E, dims.H, dims.V, p=0.0, feature_out=dims.F)
        sd = {k.replace("captioner.embed.", "captioner.embeddings."): v for k, v in p.items()}
        res = net.load_state_dict(sd, strict=False)
        assert all(k.startswith("captioner.lstm.") for k in res.missing_keys) and not res.unexpected_keys, res
        return net.to(DEV)
    net = build()
    feats, caps = batch["features"].to(DEV), batch["captions"].to(DEV)
    cap = net.forward(x.to(DEV))
    logits, _ = cap(caps, feats, 0.0)                       # reference argument order (captions, features)
    assert maxdiff(logits.detach().cpu(), logits_ref) < 3e-6
    F.cross_entropy(logits.view(-1, dims.V), caps.view(-1), ignore_index=0).backward()
    sd = dict(net.named_parameters())
    for k, v in gref.items():
        if k == "dtheta" or v is None:
            continue
        assert maxdiff(sd[k.replace("captioner.embed.", "captioner.embeddings.")].grad.cpu(), v) < 3e-6, k
    lf, _ = cap(caps, feats)                                 # the reference default sample_prob=1.0 (free running)
    assert lf.shape == logits.shape and lf.requires_grad
    tr = FusedTrainer(build(), lr=1e-3)
    l = tr.forward_backward(feats, caps, x_style=x.to(DEV), validate=True)
    assert abs(float(l[0]) - float(loss_ref)) < 3e-6
    assert maxdiff(tr.flat_g[:tr.theta_size].cpu(), gref["dtheta"]) < 3e-6
    assert maxdiff(tr.grad("captioner.init_c.weight").cpu(), gref["captioner.init_c.weight"]) < 3e-6
    l0 = float(l[0])
    for _ in range(5):
        l = tr.step(feats, caps, x_style=x.to(DEV))
    assert float(l[0]) < l0
    # the announced next minibatch (split front of the next forward beside the rank-1 passes, overlap_level 4) with the LSTM cell's
    # four heads: same trajectory as the plain step
    ta, tb = FusedTrainer(build(), lr=1e-3), FusedTrainer(build(), lr=1e-3)
    xs = x.to(DEV)
    la = [float(ta.step(feats, caps, x_style=xs)[0]) for _ in range(3)]
    lb = []
    for _ in range(3):
        lb.append(float(tb.step(feats, caps, x_style=xs, next_x_style=xs, next_features=feats, next_captions=caps)[0]))
        assert tb._pre_key is not None and tb._pre_key[-1] == 3
    assert max(abs(a - b) for a, b in zip(la, lb)) < 2e-5, (la, lb)
    assert maxdiff(ta.flat_p.cpu(), tb.flat_p.cpu()) < 2e-5


@pytest.mark.parametrize("name", CASES)
def test_module_validation_path_free_running(name):
    """validation_step runs the captioner twice: teacher forced and sample_prob = 1.0
    (cc_train_hypernet.py:187-188); scheduled sampling consumes numpy's global RNG like the reference."""
    import numpy as np
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    net = build_net(dims, p, cc=tok is None)
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV)
    with torch.no_grad():
        xs = net.captioner.embed(torch.tensor([tok], device=DEV)) if tok is not None else x.to(DEV)
        cap = net.forward(xs)
        l_tf, _ = cap(feats, caps.long(), 0.0)
        l_free, a_free = cap(feats, caps.long(), 1.0)
        np.random.seed(4321)
        l_mixed, _ = cap(feats, caps.long(), 0.5)
    assert maxdiff(l_tf.cpu(), g["logits"]) < 2e-6
    assert maxdiff(l_free.cpu(), g["logits_free"]) < 2e-6 and maxdiff(a_free.cpu(), g["alphas_free"]) < 1e-6
    assert torch.equal(l_free.argmax(-1).cpu(), g["tokens_free"])
    assert maxdiff(l_mixed.cpu(), g["logits_mixed"]) < 2e-6


@pytest.mark.parametrize("mode", ["one hot", "embedding", "histograme", "JSD"])
def test_hypernet_cc_front_ends(mode):
    """HyperNetCC's domain-embedding front-ends (cc_train_hypernet.py:63-106, :136-149): the gradient
    reaches the front-end parameters through the hypernet's input row."""
    import copy
    from cc_train_hypernet import HyperNetCC
    from models.decoderlstm import AttentionGru
    torch.manual_seed(0)
    domains = ["news\n", "sport\n", "travel\n"]
    dims0 = O.Dims(D=24, F=12, E=12, H=12, V=40, he=3 if mode == "one hot" else 10)

    class _V(_Vocab):
        def __len__(self):
            return len(self.w2i)
    vocab = _V()
    feats_dom = None
    if mode == "histograme":
        feats_dom = {d: torch.rand(len(vocab) + 1).tolist() for d in domains}
    if mode == "JSD":
        feats_dom = {d: torch.randn(2).tolist() for d in domains}
    net = HyperNetCC(dims0.F, dims0.E, dims0.H, dims0.V, vocab, domains, lr=1e-3, hyper_emb=10, embedding=mode,
                     domain_features=feats_dom)
    assert net.hyper_emb == dims0.he
    net.hypernet.captioner = AttentionGru(dims0.D, dims0.F, dims0.E, dims0.H, dims0.V, p=0.0)
    net = net.to(DEV)
    batch = O.synth_batch(dims0, B=3, T=6, P=5, seed=9)
    tb = (batch["features"].to(DEV), batch["captions"].to(DEV).float(), None, ("sport", "sport", "sport"))
    loss = net.training_step(tb, 0)
    loss.backward()
    # oracle: same parameters, torch autograd through the attached theta
    p = {k[len("hypernet."):]: v.detach().cpu() for k, v in net.state_dict().items() if k.startswith("hypernet.")}
    p = {k: v for k, v in p.items() if not k.startswith("captioner.gru.")}
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    fe = fe_mod = None
    if mode == "one hot":
        x = torch.nn.functional.one_hot(torch.tensor(1), 3).float()
    elif mode == "embedding":
        fe = net.embed.weight.detach().cpu().clone().requires_grad_(True)
        x = fe[1]
    else:
        fe_mod = copy.deepcopy(net.embed).cpu()
        for prm in fe_mod.parameters():
            prm.grad = None
        x = fe_mod(torch.tensor(feats_dom["sport\n"], dtype=torch.float32))
    theta = O.hyper_forward(q, x)
    logits, _ = O.decoder_forward(dims0, q, O.split_theta(dims0, theta), batch["features"], batch["captions"])
    ref = O.caption_loss(logits, batch["captions"])
    ref.backward()
    assert abs(float(loss) - float(ref)) < 3e-6
    assert maxdiff(net.hypernet.hn_base[0].weight.grad.cpu(), q["hn_base.0.weight"].grad) < 3e-6
    if mode == "embedding":
        assert maxdiff(net.embed.weight.grad.cpu(), fe.grad) < 3e-6
        assert float(net.embed.weight.grad[0].abs().sum()) == 0 and float(net.embed.weight.grad[1].abs().sum()) > 0
    elif mode != "one hot":
        for a, b in zip(net.embed.parameters(), fe_mod.parameters()):
            assert maxdiff(a.grad.cpu(), b.grad) < 3e-6
    out = net.validation_step(tb, 0)
    assert set(out) == {"val_loss", "val_loss with TF"} and abs(float(out["val_loss with TF"]) - float(ref)) < 3e-6
    assert len(net.configure_optimizers()[0][0].param_groups[0]["params"]) > 10


@pytest.mark.parametrize("name", ["gru_tiny_cc", "gru_tiny_flickr", "gru_odd_cc"])
def test_next_theta_fused_into_adam_pass(name):
    """optimizer_step(next_...) produces the next step's theta inside the rank-1 Adam pass
    (caphn_adam_rank_gemv_f32): same trajectory as recomputing it with caphn_hyper_forward."""
    from caphn.engine import FusedTrainer
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV)
    xs = None if tok is not None else x.to(DEV)
    ta = FusedTrainer(build_net(dims, p, cc=tok is None), lr=1e-3)
    tb = FusedTrainer(build_net(dims, p, cc=tok is None), lr=1e-3)
    la, lb = [], []
    for i in range(5):
        la.append(float(ta.step(feats, caps, x_style=xs, style_token=tok)[0]))
        lb.append(float(tb.step(feats, caps, x_style=xs, style_token=tok, next_x_style=xs, next_style_token=tok)[0]))
        if i < 4:
            assert tb._next_key is not None
    assert max(abs(a - b) for a, b in zip(la, lb)) < 2e-5, (la, lb)
    # the prefetched theta equals a fresh hypernet forward on the updated parameters
    hp = {n: tb._owned[n].data for n in tb.shape.param_names() if n in tb._owned}
    for i in range(tb._nh):
        hp[f"hn_heads.{i}.2.weight"] = tb.W2[i].data
    from caphn import ops
    xin = tb._view(tb.flat_p, "captioner.embed.weight")[tok] if tok is not None else xs
    fresh, _ = ops.hyper_forward(tb.shape, hp, xin)
    assert maxdiff(fresh.cpu(), tb._theta_next.cpu()) < 2e-6
    # a different input invalidates the prefetch
    other = 5 if tok is not None else None
    xo = None if tok is not None else torch.roll(xs, 1)
    l_new = tb.forward_backward(feats, caps, x_style=xo, style_token=other)
    l_ref = ta.forward_backward(feats, caps, x_style=xo, style_token=other)
    assert abs(float(l_new[0]) - float(l_ref[0])) < 1e-4


@pytest.mark.parametrize("name", ["gru_tiny_cc", "gru_tiny_flickr"])
def test_next_precompute_overlaps_optimizer(name):
    """step(next_features=...) runs the next minibatch's feature_fc / init_hidden / W_a f on a side stream beside
    the Adam passes (caphn_decoder_precompute + dims.precomputed): same trajectory as the plain step, also when
    the announced features do not arrive."""
    from caphn.engine import FusedTrainer
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    caps = g["captions"].to(DEV)
    f0 = g["features"].to(DEV)
    f1 = (f0 * 0.5 + 0.1).contiguous()
    seq = [f0, f1, f0, f1, f1]                  # the last announcement (f0 after step 3) is wrong on purpose
    ann = [f1, f0, f1, f0, None]
    xs = None if tok is not None else x.to(DEV)
    ta = FusedTrainer(build_net(dims, p, cc=tok is None), lr=1e-3)
    tb = FusedTrainer(build_net(dims, p, cc=tok is None), lr=1e-3)
    tc = FusedTrainer(build_net(dims, p, cc=tok is None), lr=1e-3)
    tc.overlap_level = 2
    te = FusedTrainer(build_net(dims, p, cc=tok is None), lr=1e-3)
    te.overlap_level = 3         # split front: G behind the W_ih pass, x-side gates behind the b_ih pass
    tf = FusedTrainer(build_net(dims, p, cc=tok is None), lr=1e-3)
    tf.overlap_level = 4         # ... and the theta-independent part issued before the rank-1 passes
    caps2 = caps.clone(); caps2[:, 2] = (caps2[:, 2] + 3) % dims.V
    cseq = [caps, caps2, caps, caps2, caps2]         # level 2 also announces the captions (last one wrong again)
    cann = [caps2, caps, caps2, caps, None]
    la, lb, lc, ld, le, lf = [], [], [], [], [], []
    td = FusedTrainer(build_net(dims, p, cc=tok is None), lr=1e-3)
    for i, f in enumerate(seq):
        la.append(float(ta.step(f, caps, x_style=xs, style_token=tok)[0]))
        lb.append(float(tb.step(f, caps, x_style=xs, style_token=tok, next_features=ann[i])[0]))
        if ann[i] is not None:
            assert tb._pre_key is not None and tb._pre_key[-1] == 1
        # level 2: next style + next features + next captions -> the next forward starts at the recurrent kernel
        ld.append(float(td.step(f, cseq[i], x_style=xs, style_token=tok)[0]))
        lc.append(float(tc.step(f, cseq[i], x_style=xs, style_token=tok, next_x_style=xs, next_style_token=tok,
                                next_features=ann[i], next_captions=cann[i])[0]))
        le.append(float(te.step(f, cseq[i], x_style=xs, style_token=tok, next_x_style=xs, next_style_token=tok,
                                next_features=ann[i], next_captions=cann[i])[0]))
        if ann[i] is not None:
            assert tc._pre_key is not None and tc._pre_key[-1] == 2
            assert te._pre_key is not None and te._pre_key[-1] == 3
        lf.append(float(tf.step(f, cseq[i], x_style=xs, style_token=tok, next_x_style=xs, next_style_token=tok,
                                next_features=ann[i], next_captions=cann[i])[0]))
        if ann[i] is not None:
            assert tf._pre_key is not None and tf._pre_key[-1] == 3 and tf._theta_pre is tf._theta_next
    assert max(abs(a - b) for a, b in zip(la, lb)) < 2e-5, (la, lb)
    assert maxdiff(ta.flat_p.cpu(), tb.flat_p.cpu()) < 2e-5
    assert max(abs(a - b) for a, b in zip(ld, lc)) < 2e-5, (ld, lc)
    assert maxdiff(td.flat_p.cpu(), tc.flat_p.cpu()) < 2e-5
    assert max(abs(a - b) for a, b in zip(ld, le)) < 2e-5, (ld, le)
    assert maxdiff(td.flat_p.cpu(), te.flat_p.cpu()) < 2e-5
    for a, b in zip(td.W2, te.W2):
        assert maxdiff(a.data.cpu(), b.data.cpu()) < 2e-5
    assert max(abs(a - b) for a, b in zip(ld, lf)) < 2e-5, (ld, lf)
    assert maxdiff(td.flat_p.cpu(), tf.flat_p.cpu()) < 2e-5


@pytest.mark.parametrize("B,T,P", [(1, 1, 1), (1, 2, 1), (2, 3, 2), (3, 2, 64), (1, 5, 65)])
def test_degenerate_shapes_vs_oracle(B, T, P):
    """Smallest and awkward extents: one caption, T = 1 / 2 (every input is the zeroed view, decoderlstm.py:82-88),
    one attention position, P just above a wavefront.  Forward + all gradients through the fused engine vs the oracle."""
    from caphn.engine import FusedTrainer
    dims = O.Dims(D=20, F=8, E=8, H=8, V=30, he=4)
    p = O.init_params(dims, seed=B * 100 + T * 10 + P)
    batch = O.synth_batch(dims, B, max(T, 5), P, seed=5)
    feats, caps = batch["features"], batch["captions"][:, :T].contiguous()
    caps[:, 0] = 1
    x = torch.zeros(dims.he); x[1] = 1.0
    loss_ref, _, _, _, grads_ref = O.forward_backward(dims, p, x, feats, caps)
    for rows in (False, True):          # with and without the live-row map
        tr = FusedTrainer(build_net(dims, p, cc=True), lr=1e-3)
        tr.skip_ignored_rows = rows
        out = tr.forward_backward(feats.to(DEV), caps.to(DEV), x_style=x.to(DEV))
        assert abs(float(out[0]) - float(loss_ref)) < 2e-6
        n_checked = 0
        for n, gref in grads_ref.items():
            if n in tr.offs and gref is not None:
                assert maxdiff(tr.grad(n).cpu(), gref) < 2e-6, (n, rows)
                n_checked += 1
        assert n_checked >= 20
        assert maxdiff(tr.flat_g[:tr.theta_size].cpu(), grads_ref["dtheta"]) < 2e-6


def test_all_targets_ignored_matches_torch():
    """Every target is <pad>: F.cross_entropy's mean over zero targets is NaN (0/0); the fused loss must say the same
    rather than report a finite number, and the live-row map must cope with an empty set."""
    from caphn.engine import FusedTrainer
    dims = TINY_DIMS["gru_tiny_cc"]
    g, p = load_case("gru_tiny_cc")
    caps = torch.zeros_like(g["captions"])
    tr = FusedTrainer(build_net(dims, p, cc=True), lr=1e-3)
    out = tr.forward_backward(g["features"].to(DEV), caps.to(DEV), x_style=g["x_style"].to(DEV))
    ref = F.cross_entropy(torch.zeros(4, dims.V), torch.zeros(4, dtype=torch.long), ignore_index=0)
    assert bool(torch.isnan(ref)) and bool(torch.isnan(out[0].cpu())) and float(out[1]) == 0.0


def test_host_running_ahead_of_the_gpu_changes_nothing():
    """Steps issued back to back (the host several steps ahead of the device, as in bench.py) must give the trajectory of
    steps that are synchronised one by one: nothing the host rewrites per step (Adam scalars, cached structs, pinned
    buffers) may be read later by the device.  Two synchronised runs already differ in the last bits (fp32 atomics in
    the split-K weight gradients, amplified by Adam's normalisation: ~1e-3 on single parameters after 6 steps, 1e-7
    on the loss), so the check is on the loss trajectory, with the parameters only bounded."""
    from caphn.engine import FusedTrainer
    from hypernet_attention import HyperNet
    dims = O.Dims()
    B, T, P = 32, 12, 49
    batch = O.synth_batch(dims, B, T, P, seed=2)
    feats, caps = batch["features"].to(DEV), batch["captions"].to(DEV)

    def run(sync):
        torch.manual_seed(11)
        net = HyperNet(dims.F, dims.E, dims.H, dims.V, _Vocab()).to(DEV)
        tr = FusedTrainer(net, lr=1e-3)
        losses = []
        for _ in range(6):
            losses.append(tr.step(feats, caps, style_token=4, next_style_token=4, next_features=feats,
                                  next_captions=caps).clone())      # the returned loss is a reused device buffer
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return tr.flat_p.clone(), [float(l.flatten()[0]) for l in losses]

    pa, la = run(True)
    pb, lb = run(False)
    assert la[0] > la[-1] + 2.0                       # it trains
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-5 * abs(x), (la, lb)
    assert maxdiff(pa.cpu(), pb.cpu()) < 5e-3


@pytest.mark.parametrize("B,P,Fd,H", [(5, 49, 200, 200), (3, 7, 13, 19), (2, 130, 24, 70)])
def test_standalone_bahdanau_attention_forward_backward(B, P, Fd, H):
    """models.attention.BahdanauAttention.forward stepped by hand (models/attention.py:21-46): context, weights and every
    gradient (parameters, features, hidden state; through context AND through the returned weights) against the reference's
    formula evaluated by torch in fp64."""
    from models.attention import BahdanauAttention
    torch.manual_seed(B * 100 + P)
    att = BahdanauAttention(Fd, H).to(DEV)
    with torch.no_grad():
        att.v_a.weight.mul_(3.0)
    feats = torch.randn(B, P, Fd, device=DEV, requires_grad=True)
    hid = torch.randn(B, H, device=DEV, requires_grad=True)
    wc, wa = torch.randn(B, Fd, device=DEV), torch.randn(B, P, device=DEV)
    ctx, alpha = att(feats, hid)
    assert ctx.shape == (B, Fd) and alpha.shape == (B, P)
    ((ctx * wc).sum() + (alpha * wa).sum()).backward()
    # the reference's arithmetic in fp64
    p64 = {n: q.detach().double().cpu().requires_grad_(True) for n, q in att.named_parameters()}
    f64, h64 = feats.detach().double().cpu().requires_grad_(True), hid.detach().double().cpu().requires_grad_(True)
    a1 = F.linear(f64, p64["W_a.weight"], p64["W_a.bias"])
    a2 = F.linear(h64.unsqueeze(1), p64["U_a.weight"], p64["U_a.bias"])
    sc = F.linear(torch.tanh(a1 + a2), p64["v_a.weight"], p64["v_a.bias"])
    w = F.softmax(sc, dim=1)
    c = torch.sum(w * f64, dim=1)
    ((c * wc.double().cpu()).sum() + (w.squeeze(2) * wa.double().cpu()).sum()).backward()
    assert maxdiff(ctx.detach().cpu(), c.detach()) < 5e-6 and maxdiff(alpha.detach().cpu(), w.squeeze(2).detach()) < 2e-6
    assert abs(float(alpha.sum()) - B) < 1e-4
    tol = lambda ref: 2e-5 * max(1.0, float(ref.abs().max()))
    assert maxdiff(feats.grad.cpu(), f64.grad) < tol(f64.grad)
    assert maxdiff(hid.grad.cpu(), h64.grad) < tol(h64.grad)
    for n, q in att.named_parameters():
        assert maxdiff(q.grad.cpu(), p64[n].grad) < tol(p64[n].grad), n
    with torch.no_grad():
        c2, a2_ = att(feats, hid)
    assert torch.equal(c2, ctx.detach()) and torch.equal(a2_, alpha.detach())


def test_announcement_state_machine_under_a_random_schedule():
    """Sixty steps in which every announcement is drawn at random -- honoured, wrong features, wrong captions, wrong style, only the
    features, nothing at all, a forward_backward without its optimiser step in between -- against a trainer that never announces
    anything.  The split front (overlap_level 4), the packed W_hh from the Adam pass, the rank-1 passes clearing d theta and the
    fall-backs for a batch that did not come as announced must all leave the trajectory alone."""
    import random
    from caphn.engine import FusedTrainer
    from caphn import ops
    dims = TINY_DIMS["gru_tiny_flickr"]
    g, p = load_case("gru_tiny_flickr")
    x, tok = style_args(g)
    assert tok is not None
    caps0 = g["captions"].to(DEV)
    f0 = g["features"].to(DEV)
    feats = [f0, (f0 * 0.5 + 0.1).contiguous(), (f0 * 1.5 - 0.2).contiguous()]
    caps = [caps0]
    for s in (2, 4):
        c = caps0.clone(); c[:, 2] = (c[:, 2] + s) % dims.V; c[:, 0] = caps0[:, 0]
        caps.append(c)
    toks = [tok, (tok + 1) % dims.V or 1, (tok + 2) % dims.V or 2]
    rng = random.Random(7)
    sched = [(rng.randrange(3), rng.randrange(3), rng.randrange(3)) for _ in range(61)]
    ta = FusedTrainer(build_net(dims, p, cc=False), lr=1e-3)
    tb = FusedTrainer(build_net(dims, p, cc=False), lr=1e-3)
    assert tb.overlap_level == 4
    la, lb, kinds = [], [], []
    for i in range(60):
        fi, ci, ti = sched[i]
        nfi, nci, nti = sched[i + 1]
        kind = rng.choice(["ok", "ok", "ok", "wrong_f", "wrong_c", "wrong_s", "features_only", "none", "extra_fb"])
        kinds.append(kind)
        la.append(float(ta.step(feats[fi], caps[ci], style_token=toks[ti])[0]))
        kw = {}
        if kind in ("ok", "extra_fb"):
            kw = dict(next_style_token=toks[nti], next_features=feats[nfi], next_captions=caps[nci])
        elif kind == "wrong_f":
            kw = dict(next_style_token=toks[nti], next_features=feats[(nfi + 1) % 3], next_captions=caps[nci])
        elif kind == "wrong_c":
            kw = dict(next_style_token=toks[nti], next_features=feats[nfi], next_captions=caps[(nci + 1) % 3])
        elif kind == "wrong_s":
            kw = dict(next_style_token=toks[(nti + 1) % 3], next_features=feats[nfi], next_captions=caps[nci])
        elif kind == "features_only":
            kw = dict(next_features=feats[nfi])
        lb.append(float(tb.step(feats[fi], caps[ci], style_token=toks[ti], **kw)[0]))
        if kind == "extra_fb":       # a forward_backward whose optimiser step never comes (validation-style), on both trainers
            ta.forward_backward(feats[nfi], caps[nci], style_token=toks[nti])
            tb.forward_backward(feats[nfi], caps[nci], style_token=toks[nti])
    worst = max(abs(a - b) for a, b in zip(la, lb))
    assert worst < 5e-5, (worst, [(k, a, b) for k, a, b in zip(kinds, la, lb) if abs(a - b) > 5e-5][:5])
    assert maxdiff(ta.flat_p.cpu(), tb.flat_p.cpu()) < 5e-4
    assert ops.device_error() == 0
