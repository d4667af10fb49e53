"""Data-parallel evaluation over the GPUs of one node (one process per GPU, torch.distributed;
backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The flow is embarrassingly parallel over the sample batch (every reduction inside the path is over
the feature axis), parameters are replicated, so the ONLY data-path collective is one
``all_reduce(SUM)`` of ``[sum_i loss_i, n_local]`` per KL / NLL evaluation.
For training steps the flat gradient bucket (4.3 MB for the 16-layer ala2 flow) is all-reduced once.
"""
import os

import torch
import torch.distributed as dist

__all__ = ["init_from_env", "is_distributed", "shard_size", "global_mean", "global_mean_from_sums", "global_kl_mean",
           "allreduce_gradients_", "rank_seed",
           "global_logsumexp", "global_normalized_log_weights", "global_effective_sample_size"]


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_size(n_total, rank=None, world=None):
    """Number of samples of a global batch ``n_total`` owned by ``rank`` (even split, remainder to
    the low ranks)."""
    if world is None:
        world = dist.get_world_size() if is_distributed() else 1
    if rank is None:
        rank = dist.get_rank() if is_distributed() else 0
    return n_total // world + (1 if rank < n_total % world else 0)


def rank_seed(base_seed, rank=None):
    if rank is None:
        rank = dist.get_rank() if is_distributed() else 0
    return int(base_seed) + int(rank)


def global_mean(per_sample_loss, drop_nonfinite=False):
    """Mean of a per-sample loss [n_local, 1] over ALL ranks with a single all-reduce of the
    2-vector [sum, count].  Differentiable w.r.t. the local losses (each rank's gradient is its own
    share d/d loss_i = 1 / n_global).  ``drop_nonfinite``: samples with a non-finite loss (degenerate
    geometries give log|det J| = -inf) get weight zero instead of poisoning the mean -- the reference's
    KLTrainer skips the whole optimizer step in that case (nn/training/trainers.py:198-201)."""
    if drop_nonfinite:
        ok = torch.isfinite(per_sample_loss)
        per_sample_loss = torch.where(ok, per_sample_loss, torch.zeros_like(per_sample_loss))
        n_local = ok.sum()
    else:
        n_local = torch.tensor(float(per_sample_loss.numel()), device=per_sample_loss.device)
    local_sum = per_sample_loss.sum()
    if not is_distributed():
        return local_sum / n_local.to(local_sum.dtype)
    stats = torch.stack([local_sum.detach().to(torch.float64), n_local.to(torch.float64)])
    dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    n_global = stats[1]                        # stays on the device: no host sync in the loss path
    mean_value = (stats[0] / n_global).to(local_sum.dtype)
    # value = global mean; gradient flows through the local sum only
    return mean_value + (local_sum - local_sum.detach()) / n_global.to(local_sum.dtype)


def global_mean_from_sums(sums):
    """Mean over ALL ranks from a local f64 pair [sum of the per-sample losses, number of samples kept] (what the target-energy
    kernel writes, distributions.kl_loss_sums): ONE all-reduce of the pair itself; differentiable through the local sum (each rank's
    gradient is its own share 1 / n_global)."""
    local_sum, n_local = sums[0], sums[1]
    if not is_distributed():
        return (local_sum / n_local.detach()).to(torch.float32)
    stats = sums.detach().clone()
    dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    n_global = stats[1]
    return ((stats[0] + (local_sum - local_sum.detach())) / n_global).to(torch.float32)


def global_kl_mean(target, xs, dlogp, temperature=1.0, drop_nonfinite=False):
    """mean over all ranks of the KL integrand u_target(x) - dlogp (bg.py:13-17): loss sums inside the target-energy kernel when the
    target has kernel fields, else ``global_mean`` of the per-sample tensor"""
    from .distributions import kl_loss_sums
    res = kl_loss_sums(target, tuple(xs), dlogp, temperature=temperature, drop_nonfinite=drop_nonfinite)
    if res is None:
        return global_mean(target.energy(*xs, temperature=temperature) - dlogp, drop_nonfinite=drop_nonfinite)
    return global_mean_from_sums(res[0])


def allreduce_gradients_(parameters):
    """Sum the gradients of ``parameters`` over ranks in ONE flat bucket (in place)."""
    if not is_distributed():
        return
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    offset = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[offset:offset + n].view_as(g))
        offset += n


def global_logsumexp(x):
    """log sum_i exp(x_i) over the samples of ALL ranks (x: local [n] or [n, 1]).  Two tiny all-reduces (MAX, then SUM of
    the shifted exponentials) -- the importance-weight normalisation of bg.py:62-63 for a batch sharded over the GPUs."""
    x = x.reshape(-1)
    m = x.detach().max() if x.numel() else torch.tensor(float("-inf"), dtype=x.dtype, device=x.device)
    if is_distributed():
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
    m = torch.where(torch.isfinite(m), m, torch.zeros_like(m))
    s = torch.exp(x - m).sum().to(torch.float64)
    if is_distributed():
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return m + torch.log(s).to(x.dtype)


def global_normalized_log_weights(log_w):
    """log w_i - logsumexp over all ranks (log_weights_given_latent(normalize=True), bg.py:54-64, for sharded batches)"""
    return log_w.reshape(-1) - global_logsumexp(log_w)


def global_effective_sample_size(log_w):
    """Kish ESS (bg.py:67-69) of a weight set sharded over the ranks: exp(2 LSE(log w) - LSE(2 log w))"""
    log_w = log_w.reshape(-1)
    return torch.exp(2 * global_logsumexp(log_w) - global_logsumexp(2 * log_w))

