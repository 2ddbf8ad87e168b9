"""Weight-injection protocol of the reference (utils.py:24-69), same names and semantics.

Only the two functions on the hot path are provided; the text-metric / GloVe helpers of the
reference's utils.py are out of scope.
"""
import torch
from torch import nn

from caphn import config


def flip_parameters_to_tensors(module):
    """utils.py:24-42: pop every registered parameter, re-set each as a plain zero tensor
    attribute, record the names (original registration order) in
    ``module.registered_parameters_name``; recurse into children except 'embed' / 'fc_out'."""
    attr = []
    while bool(module._parameters):
        attr.append(module._parameters.popitem())
    prev = list(getattr(module, 'registered_parameters_name', []))
    if not attr and prev:
        # already flipped and holding attached (non-Parameter) tensors from a detach=False
        # set_all_parameters: nothing to pop, keep the recorded names (the reference re-registers
        # Parameters in set_all_parameters, so its second flip finds them again)
        pass
    else:
        setattr(module, 'registered_parameters_name', [])
        for name, value in reversed(attr):
            setattr(module, name, torch.zeros(value.shape, requires_grad=True, device=value.device))
            module.registered_parameters_name.append(name)
    for name in [k for k, _ in module._modules.items()]:
        if name in ('embed', 'fc_out'):
            continue
        flip_parameters_to_tensors(module._modules[name])


def set_all_parameters(module, theta, detach=None):
    """utils.py:44-69: consecutive slices of theta[0] become the module's recorded parameters
    (names exactly 'weight' / 'bias' are skipped); returns the number of elements consumed.
    Child modules restart at offset 0, as in the reference (:68).

    detach=True reproduces the reference literally (each slice wrapped in a fresh nn.Parameter,
    utils.py:57, cutting the graph to the hypernet).  detach=False keeps the slices attached so
    the hypernet receives gradients.  Default: caphn.config.DETACH_THETA.
    """
    if detach is None:
        detach = config.DETACH_THETA
    count = 0
    for name in module.registered_parameters_name:
        if name in ('weight', 'bias'):
            continue
        cur = getattr(module, name)
        a, b = count, count + cur.numel()
        t = torch.reshape(theta[0, a:b], cur.shape)
        if detach:
            t = nn.Parameter(t)
        else:
            # a previous detach=True call may have re-registered the name as a Parameter
            module._parameters.pop(name, None)
        setattr(module, name, t)
        count += cur.numel()
    for name in [k for k, _ in module._modules.items()]:
        if name in ('embed', 'fc_out'):
            continue
        count += set_all_parameters(module._modules[name], theta, detach)
    return count


# ---- domain-embedding helpers used by HyperNetCC's front-ends (reference utils.py:354-393) -----------
def get_domain_list(cap_dir1, cap_dir2):
    """utils.py:354-368: unique domains (3rd field, separated by five spaces) in file order."""
    domains = []
    for path in (cap_dir1, cap_dir2):
        if path is None or len(path) == 0:
            continue
        with open(path, 'r') as f:
            for line in f.readlines():
                domains.append(line.split("     ")[2])
    return list(dict.fromkeys(domains))


def get_hist_embedding(cap_dir1, vocab, list_domain, do_log=True):
    """utils.py:371-393: per-domain word histogram over the vocabulary (+1 bucket for unknown words),
    optionally log10(count + 1e-4)."""
    from math import log
    eps = 0.0001
    with open(cap_dir1, 'r') as f:
        lines = [ln.split("     ") for ln in f.readlines()]
    out = {}
    for cur_domain in list_domain:
        counter_word = [0] * (len(vocab) + 1)
        for x in lines:
            if cur_domain == x[2]:
                for word in x[1].split(" "):
                    try:
                        counter_word[vocab.w2i[word]] += 1
                    except KeyError:
                        counter_word[len(vocab)] += 1
        if do_log:
            counter_word = [log(c + eps, 10) for c in counter_word]
        out[cur_domain.replace("\n", '')] = counter_word
    return out


def clean_sentence(output, voc):
    """utils.py:347-351: ids -> words without the markers and commas."""
    words = [voc.i2w.get(idx) for idx in output]
    words = [word for word in words if word not in ('<s>', ',', '<pad>', '</s>')]
    return " ".join(words)
