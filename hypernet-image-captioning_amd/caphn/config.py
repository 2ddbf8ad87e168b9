"""Process-wide switches of the module API."""

# The reference wraps every generated weight slice in a fresh nn.Parameter (utils.py:57), which
# detaches it from the hypernet graph: its hypernet never receives a gradient.
# False (default): generated weights stay attached, hypernet gradients = VJP of heads/base with
# dL/dtheta (the "intended" gradients, pinned by golden vectors).  True: literal reference behaviour.
DETACH_THETA = False

# configure_optimizers() of the three LightningModule mirrors returns caphn.optim.FusedAdam (torch.optim.Optimizer subclass on
# libcaphn, hypernet second-layer weights updated from their rank-1 factors).  True: the reference's literal
# torch.optim.Adam over dense gradients (the hypernet backward then materialises dW2, 4 B per hypernet parameter).
TORCH_OPTIMIZER = False

# The module API's decoder forward range-checks the caption token ids on the host before it launches (an out-of-range id must raise
# IndexError as torch's nn.Embedding does on the CPU, not fault on the device).  The check is one device-to-host synchronisation per
# forward: the host cannot run ahead of the GPU across it.  False: skip it (a loader that guarantees 0 <= id < vocab_size).
VALIDATE_TOKENS = True


def make_adam(params, lr, hypernet=None, max_norm=None):
    """The optimiser the reference builds at hypernet_attention.py:131 / cc_train_hypernet.py:120 / hypernet.py:121."""
    import torch
    if TORCH_OPTIMIZER:
        if hypernet is not None:
            hypernet.rank1_factors = None
        return torch.optim.Adam(params, lr=lr)
    from .optim import FusedAdam
    opt = FusedAdam(params, lr=lr, max_norm=max_norm)
    if hypernet is not None:
        opt.adopt_rank1(hypernet)
    return opt


def lightning_gradient_clipping(optimizer, gradient_clip_val=None, gradient_clip_algorithm=None):
    """Body of LightningModule.configure_gradient_clipping for the mirrors: the Trainer's gradient_clip_val (5.0 in
    cc_train_hypernet.py:405, norm clipping) becomes the fused optimiser's max_norm -- clipping then happens inside step(), over
    the dense gradients AND the rank-1 members torch's clip_grad_norm_ cannot see.  Returns False when the caller must fall back
    to Lightning's own clipping (a torch optimiser, or value clipping)."""
    from .optim import FusedAdam
    opt = getattr(optimizer, "optimizer", optimizer)        # LightningOptimizer wraps the real one
    if not isinstance(opt, FusedAdam) or (gradient_clip_algorithm not in (None, "norm")):
        return False
    opt.max_norm = None if not gradient_clip_val else float(gradient_clip_val)
    return True
