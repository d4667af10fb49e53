"""BoltzmannGenerator: prior -> flow -> target bookkeeping (bgflow/bg.py:13-165)."""
import torch

from .distributions import Energy, Sampler
from .utils import pack_tensor_in_tuple

__all__ = [
    "BoltzmannGenerator", "unnormalized_kl_div", "unormalized_nll", "sampling_efficiency",
    "effective_sample_size", "log_weights", "log_weights_given_latent", "log_weights_from_samples",
]


def unnormalized_kl_div(prior, flow, target, n_samples, temperature=1.0):
    """u_target(F(z)) - log|det J_F(z)| for z ~ prior  (bg.py:13-17)"""
    z = pack_tensor_in_tuple(prior.sample(n_samples, temperature=temperature))
    *x, dlogp = flow(*z, temperature=temperature)
    return target.energy(*x, temperature=temperature) - dlogp


def unormalized_nll(prior, flow, *x, temperature=1.0):
    """u_prior(F^-1(x)) - log|det J_F^-1(x)|  (bg.py:20-22)"""
    *z, neg_dlogp = flow(*x, inverse=True, temperature=temperature)
    return prior.energy(*z, temperature=temperature) - neg_dlogp


def log_weights_given_latent(x, z, dlogp, prior, target, temperature=1.0, normalize=True):
    x = pack_tensor_in_tuple(x)
    z = pack_tensor_in_tuple(z)
    logw = prior.energy(*z, temperature=temperature) + dlogp - target.energy(*x, temperature=temperature)
    if normalize:
        logw = logw - torch.logsumexp(logw, dim=0)
    return logw.view(-1)


def log_weights(*x, prior, flow, target, temperature=1.0, normalize=True):
    *z, neg_dlogp = flow(*x, inverse=True, temperature=temperature)
    return log_weights_given_latent(x, z, -neg_dlogp, prior, target, temperature=temperature, normalize=normalize)


def log_weights_from_samples(prior, flow, target, num_samples, batch_size, temperature=1.0, normalize=True):
    """Importance weights of ``num_samples // batch_size`` freshly sampled batches (bg.py:31-52): the batches run through the
    flow one after the other (bounded memory), weights are normalised over all of them.  Call pattern of the reference: the
    prior is sampled and the flow is run WITHOUT the temperature (``prior.sample(batch_size)``, ``flow(*z)``); the temperature
    only enters the energies of ``log_weights_given_latent``.  Unlike the reference, flows with several output tensors per
    sample are handled too (its ``x_batch, dlogp_batch = flow(*z_batch)`` assumes one)."""
    zs, xs, dls = [], [], []
    with torch.no_grad():
        for _ in range(num_samples // batch_size):
            z = pack_tensor_in_tuple(prior.sample(batch_size))
            *x, dlogp = flow(*z)
            zs.append(z); xs.append(tuple(x)); dls.append(dlogp)
        z_cat = tuple(torch.cat([z[i] for z in zs], dim=0) for i in range(len(zs[0])))
        x_cat = tuple(torch.cat([x[i] for x in xs], dim=0) for i in range(len(xs[0])))
        dlogp = torch.cat(dls, dim=0)
    return log_weights_given_latent(x_cat, z_cat, dlogp, prior, target, temperature=temperature, normalize=normalize)


def effective_sample_size(log_weights):
    """Kish effective sample size (bg.py:67-69)"""
    return torch.exp(2 * torch.logsumexp(log_weights, dim=0) - torch.logsumexp(2 * log_weights, dim=0))


def sampling_efficiency(log_weights):
    return effective_sample_size(log_weights) / len(log_weights)


class BoltzmannGenerator(Energy, Sampler):
    """``sample`` draws z ~ prior and pushes it through the flow; ``energy`` is the NLL of data
    under the generator; ``kldiv`` the per-sample reverse-KL integrand (bg.py:77-165)."""

    def __init__(self, prior, flow, target):
        super().__init__(target.event_shapes if target is not None else prior.event_shapes)
        self._prior = prior
        self._flow = flow
        self._target = target

    flow = property(lambda self: self._flow)
    prior = property(lambda self: self._prior)

    def sample(self, n_samples, temperature=1.0, with_latent=False, with_dlogp=False, with_energy=False,
               with_log_weights=False, with_weights=False):
        z = pack_tensor_in_tuple(self._prior.sample(n_samples, temperature=temperature))
        *x, dlogp = self._flow(*z, temperature=temperature)
        results = list(x)
        if with_latent:
            results.append(*z)
        if with_dlogp:
            results.append(dlogp)
        if with_energy or with_log_weights or with_weights:
            bg_energy = self._prior.energy(*z, temperature=temperature) + dlogp
            if with_energy:
                results.append(bg_energy)
            if with_log_weights or with_weights:
                logw = bg_energy - self._target.energy(*x, temperature=temperature)
                if with_log_weights:
                    results.append(logw)
                if with_weights:
                    results.append(torch.softmax(logw, dim=0).view(-1))
        return (*results,) if len(results) > 1 else results[0]

    def energy(self, *x, temperature=1.0):
        return unormalized_nll(self._prior, self._flow, *x, temperature=temperature)

    def kldiv(self, n_samples, temperature=1.0):
        return unnormalized_kl_div(self._prior, self._flow, self._target, n_samples, temperature=temperature)

    def kldiv_mean(self, n_samples, temperature=1.0, drop_nonfinite=False):
        """mean of ``kldiv`` over the batch (and over all data-parallel ranks): the scalar KLTrainer minimises (trainers.py:158-163),
        with the per-sample loss and its sum formed inside the target-energy kernel (no [B, 1] loss tensor, one all-reduce of a
        ready 2-vector).  Not in the reference API: ``kldiv(n).mean()`` is the same number."""
        from . import dp
        z = pack_tensor_in_tuple(self._prior.sample(n_samples, temperature=temperature))
        fused = getattr(self._flow, "kl_sums", None)
        if fused is not None:          # the target energy inside the generation tail's launch where the flow / target allow it
            sums = fused(z, self._target, temperature=temperature, drop_nonfinite=drop_nonfinite)
            if sums is not None:
                return dp.global_mean_from_sums(sums)
        *x, dlogp = self._flow(*z, temperature=temperature)
        return dp.global_kl_mean(self._target, x, dlogp, temperature=temperature, drop_nonfinite=drop_nonfinite)

    def log_weights(self, *x, temperature=1.0, normalize=True):
        return log_weights(*x, prior=self._prior, flow=self._flow, target=self._target,
                           temperature=temperature, normalize=normalize)

    def log_weights_given_latent(self, x, z, dlogp, temperature=1.0, normalize=True):
        return log_weights_given_latent(x, z, dlogp, self._prior, self._target, temperature=temperature,
                                        normalize=normalize)

    def trigger(self, function_name):
        return self.flow.trigger(function_name)
