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
