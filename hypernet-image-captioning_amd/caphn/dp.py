"""Data-parallel gradient exchange over RCCL (torch.distributed backend "nccl" on ROCm).

The reference has no collective call at all.  One process drives one GPU;
every rank takes its own minibatch AND its own domain row, all parameters are replicated.

What crosses xGMI per step (canonical sizes):
  * the hypernet's big second-layer gradients are rank-1 PER RANK (the hypernet sees one input
    row): dW2_i = dtheta_i (x) a_i.  Ranks all-gather the factors -- dtheta (361 200 floats) and the
    activation buffer (1.7 k floats), 1.45 MB per rank -- instead of all-reducing 576 MB; every rank
    then forms sum_r dtheta_r (x) a_r inside the fused Adam pass (caphn_adam_rank_f32, R terms).
  * everything else (decoder parameters, small hypernet layers, second-layer biases: 5.2 M floats,
    21 MB) lives in one flat arena and is all-reduced in BUCKETS, each issued as soon as the backward
    has finished that range (caphn_decoder_backward_milestone): the vocabulary projection's gradients
    (7.8 MB) right after BPTT, the embedding table's (7.7 MB) after its scatter-add, the small
    hypernet layers after the hypernet VJP, the rest at the end -- so the exchange runs on RCCL's
    stream beside the attention / feature_fc chain and the 576 MB transposed GEMV.
Gradients are averaged over ranks (DistributedDataParallel semantics): the 1/R factor is folded
into the clip coefficient.

CAPHN_FORCE_COLLECTIVES=1 (or dp.FORCE = True) sends a world-size-1 group through the same collective
calls (all_gather_into_tensor, asynchronous all_reduce, Work.wait): the RCCL code path can then be
exercised on a one-GPU box.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist

FORCE = os.environ.get("CAPHN_FORCE_COLLECTIVES") == "1"


def world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def active(group=None) -> bool:
    """True when the step has to go through the collectives: more than one rank, or forced."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE


def all_gather_factors(pack_local: torch.Tensor, out: Optional[torch.Tensor] = None, group=None, async_op: bool = False):
    """pack_local [L] -> out [R, L] (row r = rank r's factors).  Returns out, or (out, work) when async_op."""
    R = world(group)
    if out is None:
        out = torch.empty(R, pack_local.numel(), dtype=pack_local.dtype, device=pack_local.device)
    assert out.shape == (R, pack_local.numel()) and out.is_contiguous() and pack_local.is_contiguous()
    work = None
    if not active(group):
        out[0].copy_(pack_local)
    elif dist.get_backend(group) == "gloo":          # gloo has no all_gather_into_tensor
        work = dist.all_gather([out[r] for r in range(R)], pack_local, group=group, async_op=async_op)
    else:
        work = dist.all_gather_into_tensor(out.view(-1), pack_local, group=group, async_op=async_op)
    return (out, work) if async_op else out


def all_reduce_dense(flat: torch.Tensor, group=None, async_op: bool = False):
    """Sum-reduce a contiguous slice of the flat dense-gradient arena in place; returns a work handle when async."""
    if not active(group):
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def wait_all(works: List) -> None:
    """Make the CURRENT stream wait for every collective in `works` (Work.wait of the nccl backend is a stream wait,
    not a host block; gloo blocks the host)."""
    for w in works:
        if w is not None:
            w.wait()
