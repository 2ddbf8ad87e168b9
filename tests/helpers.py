"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

from oracle import caphn_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY_DIMS = {
    "gru_tiny_flickr": O.Dims(D=32, F=16, E=16, H=16, V=50, he=16),
    "gru_tiny_cc": O.Dims(D=32, F=16, E=16, H=16, V=50, he=2),
    "gru_odd_cc": O.Dims(D=37, F=13, E=11, H=19, V=83, he=5),
    "lstm_tiny": O.Dims(D=12, F=12, E=8, H=8, V=40, he=8, cell="lstm"),
    "gru_search": O.Dims(D=32, F=16, E=16, H=16, V=50, he=16),
}


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    p = {k[2:]: v for k, v in g.items() if k.startswith("p/")}
    return g, p


def style_args(g):
    tok = int(g["style_token"]) if "style_token" in g else -1
    if tok >= 0:
        return None, tok
    return g["x_style"], None


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


# ---- GPU-side helpers ------------------------------------------------------------------
def dec_params_from_oracle(p, theta, dims, device):
    """oracle names ('captioner.x') + theta -> the decoder's parameter dict on `device`."""
    out = {k[len("captioner."):]: v.to(device).contiguous() for k, v in p.items() if k.startswith("captioner.")}
    th = theta.to(device).contiguous()
    off = 0
    for name, shape in dims.cell_param_shapes():
        n = int(np.prod(shape))
        out[dims.cell + "." + name] = th[off:off + n].view(shape)
        off += n
    return out


def hyper_params_from_oracle(p, device):
    return {k: v.to(device).contiguous() for k, v in p.items() if k.startswith("hn_")}


def hyper_shape(dims):
    from caphn import ops
    return ops.HyperShape(dims.he, [(k, w) for _, k, w in O.head_layout(dims)])


def dec_dims(dims, B, T, P, raw=False):
    from caphn import ops
    return ops.DecDims(B, T, P, dims.D, dims.F, dims.E, dims.H, dims.V, cell=dims.cell, raw=raw)


def rel_err(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
