"""GPU parity of the device-resident greedy / beam search (caphn_decoder_search_*) against vectors produced by the
reference's own AttentionGru.greedy_search and by HyperNet.test_step's beam loop run around the reference's
sub-modules (tests/golden/gru_search.npz), plus batch-independence at the canonical sizes.

Discrete outcomes (tokens, beam order) are compared exactly; the fixture's smallest score gap between a kept and
a rejected candidate is 3e-3, three orders of magnitude above fp32 rounding, and the test asserts that margin.
Scores / attention maps: 2e-6 absolute (fp32, O(1) values; only summation order differs)."""
import numpy as np
import pytest
import torch

from oracle import caphn_oracle as O
from helpers import TINY_DIMS, dec_dims, dec_params_from_oracle, load_case, maxdiff

pytestmark = pytest.mark.gpu
DEV = "cuda"
ATOL = 2e-6


def _case():
    dims = TINY_DIMS["gru_search"]
    g, p = load_case("gru_search")
    params = dec_params_from_oracle(p, g["theta"], dims, DEV)
    return dims, g, p, params


def test_beam_search_matches_reference_loop():
    from caphn import ops
    dims, g, p, params = _case()
    feats = g["features"].to(DEV)
    n, P = feats.shape[0], feats.shape[1]
    k, end = int(g["beam"]), int(g["end_token"])
    assert float(g["beam_margin"].min()) > 1e-3
    seqs, lengths, scores, finished, _ = ops.decoder_search(dec_dims(dims, n, 1, P), params, feats, k, 51, end_token=end)
    assert finished.cpu().tolist() == [bool(x) for x in g["beam_finished"].tolist()]
    for i in range(n):
        if int(g["beam_finished"][i]):
            L = int(g["beam_len"][i])
            assert int(lengths[i]) == L
            assert seqs[i, :L].cpu().tolist() == g["beam_seq"][i, :L].tolist()
            assert abs(float(scores[i]) - float(g["beam_score"][i])) < 1e-5
        else:                      # the reference ran past step 50 (compute = False): 51 steps taken, beam not empty
            assert int(lengths[i]) == 52
    # polling granularity must not change anything
    s2, l2, sc2, f2, _ = ops.decoder_search(dec_dims(dims, n, 1, P), params, feats, k, 51, end_token=end, poll_every=1)
    assert torch.equal(seqs, s2) and torch.equal(lengths, l2) and torch.equal(finished, f2) and torch.equal(scores, sc2)
    # batch independence: each image alone gives the same result as inside the batch
    for i in (1, 2):
        s1, l1, sc1, f1, _ = ops.decoder_search(dec_dims(dims, 1, 1, P), params, feats[i:i + 1].contiguous(), k, 51, end_token=end)
        assert torch.equal(s1[0], seqs[i]) and int(l1[0]) == int(lengths[i]) and bool(f1[0]) == bool(finished[i])


def test_beam_search_all_completed_candidates_vs_oracle():
    """Every beam width 1..4 against the oracle's restatement, image by image (completed lists included via the
    best-of selection; margins checked so that the comparison is meaningful)."""
    from caphn import ops
    dims, g, p, params = _case()
    feats = g["features"]
    cellw = O.split_theta(dims, g["theta"])
    end = int(g["end_token"])
    for k in (1, 2, 4):
        seqs, lengths, scores, finished, _ = ops.decoder_search(dec_dims(dims, feats.shape[0], 1, feats.shape[1]), params,
                                                                feats.to(DEV), k, 51, end_token=end)
        for i in range(feats.shape[0]):
            best, score, _, _, margin = O.beam_search(p, cellw, feats[i:i + 1], k, end)
            if margin < 1e-4:
                continue
            assert bool(finished[i]) == (best is not None)
            if best is not None:
                assert seqs[i, :int(lengths[i])].cpu().tolist() == best
                assert abs(float(scores[i]) - score) < 1e-5


def test_greedy_search_matches_reference():
    from caphn import ops
    dims, g, p, params = _case()
    feats = g["features"]
    n, P = feats.shape[0], feats.shape[1]
    end, max_sentence = int(g["end_token"]), int(g["max_sentence"])
    f_post = O._feature_fc(p, feats).to(DEV).contiguous()          # greedy_search takes feature_fc outputs (:181)
    d = ops.DecDims(n, 1, P, dims.F, dims.F, dims.E, dims.H, dims.V, raw=True)
    seqs, lengths, _, finished, alphas = ops.decoder_search(d, params, f_post, 1, max_sentence, end_token=end, greedy=True,
                                                           want_alphas=True)
    for i in range(n):
        L = int(g["greedy_len"][i])
        assert int(lengths[i]) == L + 1                               # + the start token
        assert seqs[i, 1:L + 1].cpu().tolist() == g["greedy_tokens"][i, :L].tolist()
        assert bool(finished[i]) == (int(g["greedy_tokens"][i, L - 1]) == end)
        assert maxdiff(alphas[i, :L].cpu(), g["greedy_alphas"][i, :L]) < ATOL


class _Vocab:
    w2i = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3, "factual": 4}
    i2w = {**{i: "w%d" % i for i in range(50)}, **{v: k for k, v in w2i.items()}}

    def __call__(self, w):
        return self.w2i.get(w, 3)


def test_module_api_greedy_infer_and_beam():
    from hypernet_attention import HyperNet
    from models.decoderlstm import AttentionGru
    dims, g, p, _ = _case()
    net = HyperNet(dims.F, dims.E, dims.H, dims.V, _Vocab(), cc=True, hyper_emb=dims.he)
    net.captioner = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, p=0.0)
    res = net.load_state_dict(p, strict=False)
    assert not res.unexpected_keys
    net = net.to(DEV)
    feats = g["features"].to(DEV)
    with torch.no_grad():
        cap = net(g["x_style"].to(DEV))
        # reference call pattern: one image, feature_fc outputs (models/decoderlstm.py:181-182)
        sent, weights = cap.greedy_search(cap.feature_fc(feats[0:1]), end_sentence=2, max_sentence=int(g["max_sentence"]))
        L = int(g["greedy_len"][0])
        assert sent == g["greedy_tokens"][0, :L].tolist()
        assert len(weights) == L and tuple(weights[0].shape) == (1, feats.shape[1])
        assert maxdiff(torch.cat(weights, 0).cpu(), g["greedy_alphas"][0, :L]) < ATOL
        sents, _ = cap.greedy_search(cap.feature_fc(feats), 2, int(g["max_sentence"]))      # batched
        for i in range(feats.shape[0]):
            assert sents[i] == g["greedy_tokens"][i, :int(g["greedy_len"][i])].tolist()
        text = cap.infer(feats[0:1], 2, int(g["max_sentence"]), vocab=_Vocab())
        L0 = int(g["greedy_len"][0])
        assert text == " ".join(_Vocab.i2w[t] for t in g["greedy_tokens"][0, :L0].tolist() if t not in (0, 1, 2))
        out, scores = net.beam_search(feats)
        for i in range(feats.shape[0]):
            if int(g["beam_finished"][i]):
                assert out[i] == g["beam_seq"][i, :int(g["beam_len"][i])].tolist()
            else:
                assert out[i] is None
        # test_step (Flickr protocol: theta from the style token's embedding row, hypernet_attention.py:243-248)
        got = net.test_step((feats[1:2], ("factual", (torch.zeros(1, 4), None))), 0)
        net(net.captioner.embed(torch.tensor([4], device=DEV)))
        exp, _ = net.beam_search(feats[1:2])
        assert got == exp[0]


def test_full_size_beam_properties():
    """Canonical dims (D=2048, F=E=H=200, P=49, V=9684), 32 images x beam 3: size-independent properties --
    batch independence (a sub-batch decodes identically), well-formed sequences, scores are log-probabilities,
    and beam = 1 beam search equals an argmax decode under the same input rule."""
    from caphn import ops
    dims = O.Dims()
    p = O.init_params(dims, seed=5)
    p["captioner.fc.bias"] = p["captioner.fc.bias"].clone()
    p["captioner.fc.bias"][2] += 3.0
    theta = O.hyper_forward(p, torch.nn.functional.one_hot(torch.tensor(3), dims.he).float())
    params = dec_params_from_oracle(p, theta, dims, DEV)
    n, P = 32, 49
    feats = O.synth_batch(dims, n, 4, P, seed=9)["features"].to(DEV)
    seqs, lengths, scores, finished, _ = ops.decoder_search(dec_dims(dims, n, 1, P), params, feats, 3, 51, end_token=2)
    assert bool((scores <= 0).all())
    for i in range(n):
        L = int(lengths[i])
        assert int(seqs[i, 0]) == 0 and bool((seqs[i, L:] == 0).all())
        if bool(finished[i]):
            assert int(seqs[i, L - 1]) == 2 and not bool((seqs[i, 1:L - 1] == 2).any())
    sub = slice(5, 9)
    s2, l2, sc2, f2, _ = ops.decoder_search(dec_dims(dims, 4, 1, P), params, feats[sub].contiguous(), 3, 51, end_token=2)
    assert torch.equal(s2, seqs[sub]) and torch.equal(l2, lengths[sub]) and torch.equal(f2, finished[sub])
    assert maxdiff(sc2.cpu(), scores[sub].cpu()) < 1e-5
