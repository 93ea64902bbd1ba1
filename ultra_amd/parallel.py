"""Data parallelism of the training step: shard query lists, one all-reduce, divide once.

Reference: ULTRA trains single-process (dla.py:237-255 builds one graph on one device); BASELINE.json's
north_star asks for "data parallelism splits query lists across the 8 GPUs of one node, with gradient
all-reduce on RCCL".  One process per GPU (``torch.distributed``, backend "nccl" = RCCL on ROCm; "gloo" in the
CPU tests).  Every rank samples its own B lists -- slot s of rank r is global slot ``r * B + s`` of the Philox
stream, so N ranks x B lists draw exactly what one rank x N*B lists would -- and runs forward, loss and backward
on them.  The loss kernels emit UN-normalised gradients plus the additive statistics the normalisers are made of
(sum of weighted labels, ...); gradients, statistics and in-step metrics travel in ONE flat bucket, are summed
once, and only then divided.  That is what makes N x B equal 1 x N*B exactly (up to fp32 summation order)
rather than "average of per-rank normalised gradients", which is a different estimator for the listwise
softmax loss (its denominator is batch-global, base_algorithm.py:242-251).
"""
import torch


def world_and_rank():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_world_size(), torch.distributed.get_rank()
    return 1, 0


def slot_base(rank, batch_size):
    """First global sampling slot of this rank's shard (ClickSimulationFeed device path)."""
    return rank * batch_size


def all_reduce_bucket(bucket, n_reduce, world=None, group=None):
    """Sum the first ``n_reduce`` floats of the flat step bucket over the ranks, in place: one collective per step."""
    if world is None:
        world, _ = world_and_rank()
    if world > 1:
        torch.distributed.all_reduce(bucket[:n_reduce], op=torch.distributed.ReduceOp.SUM, group=group)
    return bucket
