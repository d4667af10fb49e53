"""Slow GPU tests: marked ``gpu_slow`` and NOT ``gpu`` -- `pytest -m gpu` leaves them out, `pytest -m gpu_slow` (or
`-m "gpu or gpu_slow"`) runs them on a GPU box; without a HIP device they skip."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu_slow


def test_kl_gradient_at_the_bench_batch_against_f64_over_all_samples(hip_lib, dev):
    """the direct form of tests/test_gpu_round4.py::test_kl_gradient_at_the_bench_batch: the one-pass flat KL gradient of cfg 3 at
    B = 2^18 against f64 autograd of the reference's op chain over ALL 2^18 samples (32 chunks on the host, ~5 minutes of CPU)"""
    from test_gpu_round4 import KL_GRAD_REL_L2, KL_GRAD_WORST, _grad_errors, _kl_gradient_f64, _kl_gradient_gpu, _kl_gradient_setup
    B, gen, gen_cpu, mean, z = _kl_gradient_setup(dev)
    full, loss_full = _kl_gradient_gpu(gen, z)
    ref, loss_ref = _kl_gradient_f64(gen_cpu, mean, [v.cpu().double() for v in z], 32)
    rel, worst = _grad_errors(full, ref)
    print(f"flat KL gradient at B = 2^18 vs f64 over all samples: relative L2 {rel:.2e}, worst tensor {worst[1]} {worst[0]:.2e} of its norm")
    assert rel <= KL_GRAD_REL_L2 and worst[0] <= KL_GRAD_WORST, f"rel L2 {rel:.2e}, {worst[1]} {worst[0]:.2e}"
    const = 0.5 * 66 * np.log(2 * np.pi)
    assert abs(loss_full - loss_ref) <= 1e-4 * abs(loss_ref) or abs(loss_full - loss_ref - const) <= 1e-4 * abs(loss_ref)
