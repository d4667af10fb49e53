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
ALG_BYTES = {"cfg3": 26800.0, "cfg2": 4672.0, "cfg5": 24384.0}


def make_workload(name, dev):
    if name == "cfg3":
        gen = configs.make_ala2_spline_generator(dev)
        dims = (17, 17, 17, 9)
        sampler = lambda n, g: [torch.rand(n, d, device=dev, generator=g) for d in dims]  # noqa: E731
        desc = "ala2-shaped 16x RQ-spline coupling (K=8, hidden 128x128 SiLU) + 4 icdf maps + mixed IC -> 66 xyz (cfg 3 recipe)"
    elif name == "cfg5":
        gen = configs.make_ala2_augmented_generator(dev)
        dims = (17, 17, 17, 9, 66)
        sampler = lambda n, g: [torch.rand(n, d, device=dev, generator=g) for d in dims]  # noqa: E731
        desc = "augmented ala2 flow: 10 RQ-spline + 6 affine couplings (hidden 128x128 SiLU) + 5 icdf maps + mixed IC (cfg 5, f32)"
    elif name == "cfg2":
        gen = configs.make_affine8_generator(device=dev)
        sampler = lambda n, g: [torch.randn(n, 64, device=dev, generator=g)]  # noqa: E731
        desc = "DoubleWell dim 64, 8x affine coupling blocks, DenseNet [32,64,64,32] (cfg 2)"
    else:
        raise ValueError(name)
    return gen, sampler, desc


def layer_macs(block):
    """multiply-accumulates per sample of a coupling block's conditioner"""
    macs = 0
    for m in block.modules():
        if isinstance(m, torch.nn.Linear):
            macs += m.in_features * m.out_features
    return macs


def timed_steps(gen, zs, steps):
    """K passes of the flow with HIP events around every hand-written-kernel block (events are recorded
    on the current stream = the stream the kernels are launched on).  Returns {block index: [ms,...]}."""
    from bgflow_amd.flow import CouplingFlow, WrapFlow
    evs = []
    with torch.no_grad():
        for _ in range(steps):
            xs = tuple(zs)
            total = 0.0
            for i, block in enumerate(gen.flow):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                *xs, dd = block(*xs)
                e1.record()
                total = total + dd
                evs.append((i, e0, e1))
    return evs


def measured_traffic(kernel, batch=1 << 20):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/<round>_traffic.json,
    written by tools/profile_round.sh: FETCH_SIZE x 2 [gfx950 rule] + WRITE_SIZE, KiB units), or None"""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_traffic.json")))
    if not files:
        return None
    try:
        v = json.load(open(files[-1])).get(kernel, {}).get("hbm_bytes_per_launch")
        return None if v is None else v * (batch / float(1 << 20))     # the PMC passes ran at 2^20 samples per launch
    except Exception:
        return None


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
    # passes over the same n_samples batch until >= 12 s of CPU work (bounded at 8 passes)
    passes, dt = 0, 0.0
    while dt < 12.0 and passes < 8:
        t0 = time.perf_counter()
        fo.run_flow(gen.flow, u, dtype=np.float32)
        dt += time.perf_counter() - t0
        passes += 1
    cpu = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), cpu)
    except OSError:
        pass
    return dict(value=passes * n_samples / dt, unit="samples/s", cores=orc.num_threads(), kind="port", cpu=cpu,
                sample=f"{passes} pass(es) over {n_samples} samples of the same flow, forward + log|det J|, f32 ({dt:.1f} s); "
                       f"C restatement of the reference's op chain (oracle/), OpenMP over samples")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--gemm", default=None, choices=["f32", "f16x2", "bf16"],
                    help="conditioner GEMM mode of the fused coupling kernel (default: bgflow_amd.dense.GEMM_MODE)")
    ap.add_argument("--batch", type=int, default=1 << 20, help="samples per GPU per step")
    ap.add_argument("--cpu-samples", type=int, default=1 << 17)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the exact-f32 and cfg-2 side measurements")
    ap.add_argument("--kl-steps", type=int, default=3, help="extra: time this many KL-loss training steps (0 = skip)")
    ap.add_argument("--kl-batch", type=int, default=1 << 18, help="samples per GPU per KL step")
    args = ap.parse_args()

    # BGK_BENCH_TEST_SHARED_GPU=1: self-test of the multi-rank code path on a ONE-GPU box (all ranks on cuda:0, gloo for
    # the collectives -- RCCL refuses two ranks on one device).  Never set by the driver; numbers of such a run mean nothing.
    shared_gpu_test = os.environ.get("BGK_BENCH_TEST_SHARED_GPU") == "1"
    rank, world, local = dp.init_from_env("gloo" if shared_gpu_test else "nccl")
    if shared_gpu_test:
        local = 0
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from bgflow_amd import dense as _dense
    if args.gemm:
        _dense.GEMM_MODE = args.gemm
    gen, sampler, desc = make_workload(args.workload, dev)
    g = torch.Generator(device=dev).manual_seed(dp.rank_seed(1234, rank))
    zs = sampler(args.batch, g)

    for _ in range(args.warmup):
        timed_steps(gen, zs, 1)
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    evs = timed_steps(gen, zs, args.steps)
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- extra: the same workload with the conditioner GEMMs in exact-f32 MFMA mode (bit-identical to the CPU oracle)
    exact = None
    gemm_mode = _dense.GEMM_MODE
    if args.workload == "cfg3" and gemm_mode != "f32" and rank == 0 and world == 1 and not args.no_extras:
        _dense.GEMM_MODE = "f32"
        timed_steps(gen, zs, 1)
        torch.cuda.synchronize(dev)
        te = time.perf_counter()
        timed_steps(gen, zs, 3)
        torch.cuda.synchronize(dev)
        te = (time.perf_counter() - te) / 3
        exact = dict(gemm="f32", value=args.batch / te, unit="samples/s", ms_per_step=1e3 * te, steps=3,
                     note="same flow, conditioner GEMMs on the f32-input MFMA (exact fma chain, bit-identical to the oracle)")
        _dense.GEMM_MODE = gemm_mode
        timed_steps(gen, zs, 1)   # re-pack for the headline mode (KL bench below uses the generic path)
        torch.cuda.synchronize(dev)

    # ---- extra (cfg 5 is specified "fp32 vs bf16"): the same flow with bf16 weights / GEMM inputs in the spline layers
    bf16_leg = None
    if args.workload == "cfg5" and gemm_mode != "bf16" and rank == 0 and world == 1 and not args.no_extras:
        _dense.GEMM_MODE = "bf16"
        timed_steps(gen, zs, 2)
        torch.cuda.synchronize(dev)
        tb = time.perf_counter()
        timed_steps(gen, zs, 5)
        torch.cuda.synchronize(dev)
        tb = (time.perf_counter() - tb) / 5
        bf16_leg = dict(gemm="bf16", value=args.batch / tb, unit="samples/s", ms_per_step=1e3 * tb, steps=5,
                        note="REDUCED PRECISION leg: bf16 weights + GEMM inputs in the 10 spline layers (f32 accumulate; knots, bin "
                             "search, log-det f32); the 6 affine layers stay split-f16")
        _dense.GEMM_MODE = gemm_mode
        timed_steps(gen, zs, 1)
        torch.cuda.synchronize(dev)

    # ---- extra: BASELINE.json configs[1] (8 affine coupling blocks, dim 64, batch 2^20) on the same GPU
    cfg2 = None
    if args.workload == "cfg3" and not args.no_extras and rank == 0 and world == 1:
        gen2, sampler2, desc2 = make_workload("cfg2", dev)
        z2 = sampler2(1 << 20, torch.Generator(device=dev).manual_seed(1234))
        timed_steps(gen2, z2, 2)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        timed_steps(gen2, z2, 5)
        torch.cuda.synchronize(dev)
        t2 = (time.perf_counter() - t2) / 5
        cfg2 = dict(workload=desc2, value=(1 << 20) / t2, unit="samples/s", ms_per_step=1e3 * t2, steps=5, batch=1 << 20,
                    hbm_view=dict(algorithmic_bytes_per_sample=ALG_BYTES["cfg2"],
                                  achieved_GBs=ALG_BYTES["cfg2"] * (1 << 20) / t2 / 1e9, peak_GBs=HBM_PEAK_GBS),
                    note="fused affine coupling kernel (both conditioner MLPs on the f16 matrix cores + affine tail), 8 launches")
        del gen2, z2

    # ---- extra (second half of BASELINE.json's metric): KL-loss training steps/s -------------------------
    # one step = kldiv(B).mean() -> backward through the hand-written backward kernels -> one all-reduce of
    # [sum, n] (+ one flat gradient bucket) -> Adam.  Reported next to the headline, not instead of it.
    kl = None
    if args.kl_steps > 0:
        params = [p for p in gen.flow.parameters()]
        opt = torch.optim.Adam(params, lr=1e-5)
        zk = sampler(args.kl_batch, g)

        def kl_step():
            opt.zero_grad(set_to_none=True)
            *x, dlogp = gen.flow(*zk)
            loss = dp.global_mean(gen._target.energy(*x) - dlogp, drop_nonfinite=True)
            loss.backward()
            dp.allreduce_gradients_(params)
            opt.step()
            return loss
        kl_step()
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        tk = time.perf_counter()
        for _ in range(args.kl_steps):
            last = kl_step()
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        dtk = time.perf_counter() - tk
        kl = dict(steps_per_s=args.kl_steps / dtk, samples_per_s=args.kl_steps * args.kl_batch * world / dtk,
                  batch_per_gpu=args.kl_batch, steps=args.kl_steps, loss=float(last.detach()),
                  note="fwd: one-launch coupling layers (training variant, saves pre-activations + spline parameters) + IC / CDF kernels; "
                       "bwd: bgk_rqs_backward / bgk_ic_ic2xyz_backward, conditioner input-gradient chain on bgk_dense_backward_dx, "
                       "split-K weight-gradient GEMMs, bias gradients on bgk_column_sum; one all-reduce of [sum, n] + one "
                       "gradient bucket; Adam")

    total_samples = args.batch * world * args.steps
    value = total_samples / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    if rank == 0:
        from bgflow_amd.flow import CouplingFlow
        n_blocks = len(gen.flow)
        block_ms = [0.0] * n_blocks
        for i, e0, e1 in evs:
            block_ms[i] += e0.elapsed_time(e1) / args.steps
        coupling = [i for i, b in enumerate(gen.flow) if isinstance(b, CouplingFlow)]
        # dominant kernel = the coupling-layer kernel (one launch per coupling block)
        t_coupling_ms = sum(block_ms[i] for i in coupling)
        n_launch = len(coupling)
        avg_launch_s = 1e-3 * t_coupling_ms / n_launch
        flops_per_launch = 2.0 * sum(layer_macs(gen.flow[i]) for i in coupling) / n_launch * args.batch
        alg_bytes_step = ALG_BYTES[args.workload] * args.batch
        if args.workload in ("cfg3", "cfg5"):
            split = gemm_mode in ("f16x2", "bf16")
            roof = dict(bound="mfma", achieved=flops_per_launch / avg_launch_s / 1e12, peak=MFMA_F32_PEAK_TFLOPS,
                        unit="TFLOP/s", traffic=measured_traffic("coupling_rqs_dense_h2_kernel" if split else "coupling_rqs_dense_kernel", args.batch),
                        kernel=("coupling_rqs_dense_h2_kernel (fused DenseNet on the f16 matrix cores in split-f16 form + RQ-spline "
                                "coupling layer)" if split else
                                "coupling_rqs_dense_kernel (fused DenseNet on the f32-input MFMA + RQ-spline coupling layer)"),
                        note=("achieved = algorithmic f32 flops of the conditioner (2*MACs) / launch time, peak = dense f32-input MFMA "
                              "peak: the split-f16 form executes 3 f16 MFMAs per product at 16x the f32 rate, so the fraction can "
                              "exceed 1; the kernel is then VALU-issue bound (spline + SiLU), see profiles/README.md" if split else
                              "f32-input MFMA and f32 VALU share the issue port on gfx950: time = MFMA + VALU, see profiles/README.md"),
                        launches_per_step=n_launch, avg_launch_ms=1e3 * avg_launch_s,
                        flops_per_launch=flops_per_launch,
                        hbm_view=dict(note="SURVEY 8(d) algorithmic bytes (unfused kernel boundaries: 26 800 B per sample for cfg 3) "
                                           "per GPU and step / step time, against the 8 TB/s HBM3E peak",
                                      algorithmic_bytes_per_step=alg_bytes_step,
                                      achieved_GBs=alg_bytes_step / (1e-3 * ms_per_step) / 1e9, peak_GBs=HBM_PEAK_GBS,
                                      frac=alg_bytes_step / (1e-3 * ms_per_step) / 1e9 / HBM_PEAK_GBS))
        else:
            roof = dict(bound="hbm", achieved=alg_bytes_step / (1e-3 * ms_per_step) / 1e9, peak=HBM_PEAK_GBS,
                        unit="GB/s", traffic=measured_traffic("coupling_affine_dense_kernel", args.batch),
                        kernel="coupling_affine_dense_kernel (fused: 2 DenseNets on the f16 matrix cores + affine tail)")
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["block_ms"] = [round(v, 3) for v in block_ms]
        out = dict(metric="flow samples/s (fwd+log|detJ|) at batch 2^20", value=value, unit="samples/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="bf16" if gemm_mode == "bf16" else "f32", data="synthetic",
                   config=dict(workload=desc, batch_per_gpu=args.batch, global_batch=args.batch * world,
                               parallelism=f"dp{world}",
                               conditioner_gemm={"f16x2": "split-f16: f32 operands as hi+lo f16 pairs, 3 MFMAs per product, f32 accumulate "
                                                          "(f32-class accuracy)",
                                                 "f32": "f32-input MFMA (exact)",
                                                 "bf16": "REDUCED PRECISION: bf16 weights and GEMM inputs (spline layers), f32 accumulate; "
                                                         "spline / log-det arithmetic f32"}[gemm_mode]),
                   roofline=roof)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_samples)
        if exact is not None:
            out["exact_f32_mode"] = exact
        if cfg2 is not None:
            out["cfg2"] = cfg2
        if bf16_leg is not None:
            out["bf16_mode"] = bf16_leg
        if kl is not None:
            out["kl"] = kl
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
