#!/usr/bin/env python
"""bench.py -- flow samples/s (forward + log|det J|) of the BASELINE workload on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2] [--batch B]

One "step" = one pass of the hot path (prior sample already resident in HBM -> flow forward ->
x, dlogp) over one batch of B synthetic samples PER RANK (weak scaling: the batch is sharded
data-parallel, parameters replicated, no data-path collective in sampling).  Rank 0 prints ONE JSON
line.  `roofline` describes the dominant kernel (HIP-event timed inside this process, on the stream
the kernels are launched on); `cpu_baseline` times the CPU oracle (oracle/, kind "port") on a bounded
sample of the same workload on the host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bgflow_amd import configs, dp  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-input MFMA peak

# algorithmic bytes per sample of the hand-written kernels (fp32)
ALG_BYTES = {"cfg3": 26800.0, "cfg2": 4672.0}


def make_workload(name, dev):
    if name == "cfg3":
        gen = configs.make_ala2_spline_generator(dev)
        dims = (17, 17, 17, 9)
        sampler = lambda n, g: [torch.rand(n, d, device=dev, generator=g) for d in dims]  # noqa: E731
        desc = "ala2-shaped 16x RQ-spline coupling (K=8, hidden 128x128 SiLU) + 4 icdf maps + mixed IC -> 66 xyz (cfg 3 recipe)"
    elif name == "cfg2":
        gen = configs.make_affine8_generator(device=dev)
        sampler = lambda n, g: [torch.randn(n, 64, device=dev, generator=g)]  # noqa: E731
        desc = "DoubleWell dim 64, 8x affine coupling blocks, DenseNet [32,64,64,32] (cfg 2)"
    else:
        raise ValueError(name)
    return gen, sampler, desc


def kernel_roofline(gen, zs, workload, steps):
    """HIP-event timing of the dominant hand-written kernel on the current stream."""
    import bgflow_amd as bg
    dev = zs[0].device
    B = zs[0].shape[0]
    with torch.no_grad():
        # locate the most expensive coupling layer type: time every block once with events
        xs = tuple(zs)
        timings = []
        for block in gen.flow:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            *xs, _ = block(*xs)
            ev1.record()
            torch.cuda.synchronize(dev)
            timings.append(ev0.elapsed_time(ev1))
    return timings


def cpu_baseline(workload, n_samples):
    """the CPU oracle (C restatement + OpenMP) on a bounded sample of the same workload"""
    from oracle import flow_oracle as fo
    from oracle import oracle as orc
    orc.build()
    if workload == "cfg3":
        gen = configs.make_ala2_spline_generator()
        rng = np.random.default_rng(1234)
        u = [rng.random((n_samples, d), dtype=np.float32) for d in (17, 17, 17, 9)]
    else:
        gen = configs.make_affine8_generator()
        rng = np.random.default_rng(1234)
        u = [rng.standard_normal((n_samples, 64), dtype=np.float32)]
    fo.run_flow(gen.flow, [v[:256] for v in u], dtype=np.float32)   # warm-up
    t0 = time.perf_counter()
    fo.run_flow(gen.flow, u, dtype=np.float32)
    dt = time.perf_counter() - t0
    return dict(value=n_samples / dt, unit="samples/s", cores=orc.num_threads(), kind="port",
                sample=f"{n_samples} samples of the same flow, forward + log|det J|, f32, one pass ({dt:.1f} s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--batch", type=int, default=1 << 20, help="samples per GPU per step")
    ap.add_argument("--cpu-samples", type=int, default=1 << 16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank, world, local = dp.init_from_env("nccl")
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    gen, sampler, desc = make_workload(args.workload, dev)
    g = torch.Generator(device=dev).manual_seed(dp.rank_seed(1234, rank))
    zs = sampler(args.batch, g)

    def step():
        with torch.no_grad():
            *x, dlogp = gen.flow(*zs)
        return x, dlogp

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_samples = args.batch * world * args.steps
    value = total_samples / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    if rank == 0:
        block_ms = kernel_roofline(gen, zs, args.workload, args.steps)
        k = int(np.argmax(block_ms))
        # algorithmic bytes of the dominant block (per launch) -- see DESIGN.md
        from bgflow_amd.flow import CouplingFlow
        blk = gen.flow[k]
        alg = None
        if isinstance(blk, CouplingFlow):
            d_t = zs[blk.transformed_indices[0]].shape[1] if args.workload == "cfg3" else 32
        roof = dict(bound="hbm", achieved=ALG_BYTES[args.workload] * args.batch / (ms_per_step * 1e-3) / 1e9,
                    peak=HBM_PEAK_GBS, unit="GB/s", traffic=None,
                    note="whole-step algorithmic bytes / step time (per-kernel breakdown: see DESIGN.md)",
                    block_ms=[round(v, 3) for v in block_ms])
        roof["frac"] = roof["achieved"] / roof["peak"]
        out = dict(metric="flow samples/s (fwd+log|detJ|)", value=value, unit="samples/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload=desc, batch_per_gpu=args.batch, global_batch=args.batch * world,
                               parallelism=f"dp{world}"),
                   roofline=roof)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_samples)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
