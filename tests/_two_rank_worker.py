"""Worker of tests/test_gpu_round3.py::test_two_rank_rccl_kl_step (launched by torch.distributed.run, one rank per GPU, backend nccl =
RCCL): a sharded KL evaluation + training step -- ONE all-reduce of the [sum loss, n] pair and one of the flat gradient bucket -- must
give every rank the same loss, the same gradients and the same parameters as the other rank."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bgflow_amd import configs, dp                                   # noqa: E402
from bgflow_amd.training import FlatAdam                             # noqa: E402


def main():
    rank, world, local = dp.init_from_env("nccl")
    assert world == 2 and torch.distributed.get_backend() == "nccl"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    gen = configs.make_ala2_spline_generator(dev)
    opt = FlatAdam([p for p in gen.flow.parameters()], lr=1e-4)
    g = torch.Generator(device=dev).manual_seed(dp.rank_seed(1234, rank))
    z = [torch.rand(4096, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
    opt.zero_grad()
    *x, dlogp = gen.flow(*z)
    loss = dp.global_kl_mean(gen._target, x, dlogp, drop_nonfinite=True)
    opt.backward(loss)
    opt.allreduce_gradients()
    opt.step()
    torch.cuda.synchronize(dev)
    # every rank must hold the same loss, gradient bucket and parameters
    probe = torch.stack([loss.detach().double(), opt.grad.double().norm(), opt.flat.double().norm()])
    both = [torch.zeros_like(probe) for _ in range(world)]
    torch.distributed.all_gather(both, probe)
    assert torch.equal(both[0], both[1]), f"ranks disagree: {both}"
    assert torch.isfinite(probe).all() and float(probe[1]) > 0
    # the local shards differ (per-rank seed), so the global mean is not the local one
    local_mean = (gen._target.energy(*x) - dlogp).detach()
    local_mean = local_mean[torch.isfinite(local_mean)].mean()
    gathered = [torch.zeros_like(local_mean) for _ in range(world)]
    torch.distributed.all_gather(gathered, local_mean)
    assert abs(float(sum(gathered) / world) - float(loss)) <= 1e-3 * abs(float(loss))
    if rank == 0:
        print("TWO_RANK_OK", float(loss), flush=True)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
