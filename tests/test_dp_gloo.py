"""CPU, world_size 2 (gloo): the data-parallel helpers of bgflow_amd.dp -- the single all-reduce of
[sum loss, count] for the KL / NLL mean, the flat gradient bucket, sharding and per-rank seeds."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from bgflow_amd import dp
    r, w, l = dp.init_from_env("gloo")
    assert (r, w) == (rank, world) and dp.is_distributed()
    # uneven shards of a global batch of 11 samples
    n_local = dp.shard_size(11)
    assert n_local == (6 if rank == 0 else 5)
    g = torch.Generator().manual_seed(dp.rank_seed(1234))
    theta = torch.nn.Parameter(torch.tensor([0.5, -1.0]))
    x = torch.randn(n_local, 2, generator=g)
    per_sample = ((x * theta).sum(-1, keepdim=True)) ** 2          # stand-in for u(x_i) - dlogp_i
    loss = dp.global_mean(per_sample)
    loss.backward()
    dp.allreduce_gradients_([theta])
    # the [sum, n] pair the target-energy kernel writes (distributions.kl_loss_sums), all-reduced as it is: same global mean,
    # gradient = each rank's own share 1 / n_global
    theta2 = torch.nn.Parameter(torch.tensor([0.5, -1.0]))
    ps2 = ((x * theta2).sum(-1, keepdim=True)) ** 2
    sums = torch.stack([ps2.sum().double(), torch.tensor(float(n_local), dtype=torch.float64)])
    loss2 = dp.global_mean_from_sums(sums)
    loss2.backward()
    dp.allreduce_gradients_([theta2])
    assert abs(float(loss2.detach()) - float(loss.detach())) <= 1e-5 * abs(float(loss.detach()))
    assert torch.allclose(theta2.grad, theta.grad, rtol=1e-5, atol=1e-6)
    # and the KL mean through a target without kernel fields (CPU tensors): the per-sample path with one all-reduce
    import bgflow_amd as bg
    tgt = bg.NormalDistribution(2)
    klm = dp.global_kl_mean(tgt, (x,), torch.zeros(n_local, 1))
    assert abs(float(klm) - float(dp.global_mean(tgt.energy(x)))) <= 1e-6
    logw = x[:, 0] * 3.0
    # plain lists, not tensors: tensor storages travel through the queue as file descriptors served by THIS process,
    # which may already have exited when the parent reads them
    out.put((rank, float(loss.detach()), theta.grad.tolist(), x.tolist(),
             float(dp.global_logsumexp(logw)), dp.global_normalized_log_weights(logw).tolist(),
             float(dp.global_effective_sample_size(logw))))
    dist.barrier()
    dist.destroy_process_group()


def test_global_mean_and_gradient_bucket_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, g0, x0, lse0, nw0, ess0), (_, l1, g1, x1, lse1, nw1, ess1) = res
    g0, g1, x0, x1, nw0, nw1 = (torch.tensor(v) for v in (g0, g1, x0, x1, nw0, nw1))
    assert l0 == pytest.approx(l1, rel=1e-6)                      # every rank holds the GLOBAL mean
    assert not torch.equal(x0[:5], x1)                            # per-rank RNG streams differ
    theta = torch.nn.Parameter(torch.tensor([0.5, -1.0]))
    x = torch.cat([x0, x1])
    ref = (((x * theta).sum(-1, keepdim=True)) ** 2).mean()
    ref.backward()
    assert l0 == pytest.approx(float(ref.detach()), rel=1e-5)
    assert torch.allclose(g0, theta.grad, rtol=1e-5, atol=1e-6) and torch.allclose(g1, theta.grad, rtol=1e-5, atol=1e-6)
    # importance-weight normalisation / ESS over the sharded batch == the single-process formulas of bg.py on the union
    from bgflow_amd.bg import effective_sample_size
    logw = torch.cat([x0[:, 0], x1[:, 0]]) * 3.0
    assert lse0 == pytest.approx(float(torch.logsumexp(logw, 0)), rel=1e-6) and lse1 == pytest.approx(lse0, rel=1e-7)
    assert torch.allclose(torch.cat([nw0, nw1]), logw - torch.logsumexp(logw, 0), atol=1e-6)
    assert ess0 == pytest.approx(float(effective_sample_size(logw)), rel=1e-5) and ess1 == pytest.approx(ess0, rel=1e-7)


def test_single_process_is_a_noop():
    from bgflow_amd import dp
    assert not dp.is_distributed()
    assert dp.shard_size(10) == 10
    v = torch.arange(4.0).reshape(4, 1)
    assert float(dp.global_mean(v)) == pytest.approx(1.5)
