#!/usr/bin/env python
"""bench.py -- flow samples/s (forward + log|det J|) of the BASELINE workload on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg5] [--batch B]

One "step" = one pass of the hot path (prior sample already resident in HBM -> flow forward -> x, dlogp) over one batch of
B synthetic samples PER RANK (weak scaling: the batch is sharded data-parallel, parameters replicated, no data-path collective
in sampling).  With --gpus N > 1 and no torchrun environment the script launches its N ranks itself
(``python -m torch.distributed.run --nproc-per-node N``, rendezvous on 127.0.0.1); under torchrun it is one of the ranks.
Rank 0 prints ONE JSON line:

  value / ms_per_step   whole-job samples/s over the timed K steps (barrier + synchronize on both sides, max over ranks)
  roofline              the dominant kernel (the fused coupling layer, one launch per coupling block): SURVEY.md 8(d)'s algorithmic
                        bytes per launch / its average launch duration (HIP events on the launch stream, this process) against the
                        8 TB/s HBM peak; `mfma_util` = matrix-core flops the kernel executes / duration against the 2.5 PFLOP/s
                        dense f16 peak; `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc passes (a profile
                        constant, labelled as such); `step_view` = the whole step in the same accounting
  cpu_baseline          the reference's op chain on the host cores (rank 0, N = 1): `value` = stock torch-CPU ops, all physical
                        cores, processes pinned (oracle/torch_flow.py, kind "port"); `inverse` and `kl` = the NLL direction and KL
                        training steps on the same cores; the scalar C oracle is timed beside it; `parity_sample` = GPU / f32 C
                        oracle / torch f32 chain against the f64 oracle on 2^14 random samples (log-det error statistics, the share
                        of samples beyond 1e-5, bin-index ties)
  inverse               the NLL direction of the same flow on the GPU (samples/s)
  rccl                  (N > 1) backend, world size, the device behind every rank, a same-run one-rank pass and the weak-scaling
                        efficiency against it, the [sum loss, n] all-reduce alone
  exact_f32_mode, cfg2, cfg5, kl   side measurements (HIP events, >= 10 steps each)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16 / bf16 MFMA peak
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-input MFMA peak (gemm_mode "f32" only)

# algorithmic bytes per sample of the hand-written kernels (fp32), whole step
ALG_BYTES = {"cfg3": 26800.0, "cfg2": 4672.0, "cfg5": 24384.0}


def default_batch(gpus, workload):
    """(samples per GPU per step, is BASELINE cfg 4) when --batch is not given: 2^20 per GPU (the batch the headline metric is quoted on);
    the cfg-3 flow on 8 GPUs is BASELINE cfg 4 -- 2^22 samples sharded data-parallel, 2^19 per rank"""
    if gpus == 8 and workload == "cfg3":
        return 1 << 19, True
    return 1 << 20, False


def self_launch(args_list, n):
    """Re-exec under torch.distributed.run with one rank per GPU and relay rank 0's JSON line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + args_list
    return subprocess.call(cmd, env=env)


def make_workload(name, dev):
    from bgflow_amd import configs
    if name == "cfg3":
        gen = configs.make_ala2_spline_generator(dev)
        dims = (17, 17, 17, 9)
        sampler = lambda n, g: [torch.rand(n, d, device=dev, generator=g) for d in dims]  # noqa: E731
        desc = "ala2-shaped 16x RQ-spline coupling (K=8, hidden 128x128 SiLU) + 4 icdf maps + mixed IC -> 66 xyz (cfg 3 recipe)"
    elif name == "cfg5":
        gen = configs.make_ala2_augmented_generator(dev)
        dims = (17, 17, 17, 9, 66)
        sampler = lambda n, g: [torch.rand(n, d, device=dev, generator=g) for d in dims]  # noqa: E731
        desc = "augmented ala2 flow: 10 RQ-spline + 6 affine couplings (hidden 128x128 SiLU) + 5 icdf maps + mixed IC (cfg 5)"
    elif name == "cfg2":
        gen = configs.make_affine8_generator(device=dev)
        sampler = lambda n, g: [torch.randn(n, 64, device=dev, generator=g)]  # noqa: E731
        desc = "DoubleWell dim 64, 8x affine coupling blocks, DenseNet [32,64,64,32] (cfg 2)"
    else:
        raise ValueError(name)
    return gen, sampler, desc


def coupling_stats(block, gemm_mode):
    """(SURVEY 8(d) algorithmic bytes per sample, conditioner MACs per sample, matrix-core flops per 32-sample tile the fused
    kernel executes) of one coupling block"""
    tr = block.transformer
    lin = [m for m in block.modules() if isinstance(m, torch.nn.Linear)]
    macs = sum(m.in_features * m.out_features for m in lin)
    if type(tr).__name__ == "ConditionalSplineTransformer":
        P, n_in = lin[-1].out_features, lin[0].in_features
        nm = int(torch.as_tensor(tr._is_circular).numel())
        d = nm if nm > 1 else (P // 24 if bool(torch.as_tensor(tr._is_circular).any()) else P // 25)   # P = 3 K d + #non-circular, K = 8
        alg = 4.0 * (P + 2 * d + 2)
        S0 = (n_in + 1 + 15) // 16
        chunks = (d + 4) // 5
        last_tiles = ((d - 5 * (chunks - 1)) * 25 + 31) // 32
        if gemm_mode == "f32":      # 32x32x2 f32 MFMAs: k2-steps incl. bias, 4 tiles each (bgk_fused.hip)
            T0 = (((n_in + 1) // 2 + 3) & ~3) + 1
            n_mfma = 4 * T0 + 4 * 65 + (chunks - 1) * 4 * 65 + (2 if last_tiles <= 2 else 4) * 65
            return alg, macs, n_mfma * 2 * 32 * 32 * 2
        per = 1 if gemm_mode == "bf16" else 3
        n_mfma = 4 * per * S0 + (4 * 8 * per + 4) + (chunks - 1) * (4 * 8 * per + 4)
        n_mfma += (2 * 8 * per + 2) if (last_tiles <= 2 and gemm_mode == "f16x2") else (4 * 8 * per + 4)
        return alg, macs, n_mfma * 2 * 32 * 32 * 16
    d = lin[-1].out_features            # affine: mu, s, y in, y out, dlogp
    return 4.0 * (4 * d + 2), macs, None


class _SegmentTimer:
    """HIP events (recorded on the current stream = the stream the kernels are launched on) around every segment of
    SequentialFlow.run -- the pass itself is the product code path (one running log-det buffer, written by the kernels)."""

    def __init__(self):
        self.events = []

    def __call__(self, i, label):
        timer = self

        class _Ctx:
            def __enter__(self_inner):
                self_inner.e0, self_inner.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self_inner.e0.record()

            def __exit__(self_inner, *exc):
                self_inner.e1.record()
                timer.events.append((i, self_inner.e0, self_inner.e1))
                return False
        return _Ctx()


N_INPUT_SETS = 2      # the timed loops rotate this many synthetic input sets (2^20 x 60 floats = 252 MB each: a set does not survive in
#                       the 256 MiB Infinity Cache until it is used again, as the one reused set of the earlier rounds might have)


def timed_steps(gen, zsets, steps, inverse=False, first=0):
    """K passes of the flow with HIP events around every segment, pass k on input set (first + k) mod len(zsets).  Returns
    [(segment index, start event, end event), ...]."""
    timer = _SegmentTimer()
    with torch.no_grad():
        for k in range(steps):
            gen.flow.run(tuple(zsets[(first + k) % len(zsets)]), inverse=inverse, around=timer)
    return timer.events


def event_ms_per_call(fn, steps, warmup):
    """average duration of fn() over `steps` calls, HIP events on the current stream"""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def flow_pass(gen, zsets, inverse=False):
    """callable: one pass of the flow, successive calls rotate through the input sets ``zsets`` (a list of input tuples)"""
    if torch.is_tensor(zsets[0]):
        zsets = [zsets]
    k = [0]

    def run():
        zs = zsets[k[0] % len(zsets)]
        k[0] += 1
        with torch.no_grad():
            return gen.flow(*zs, inverse=inverse)
    return run


def kl_side_leg(name, dev, batch, steps):
    """KL-loss training steps of BASELINE cfg 2 / cfg 5 on this rank (round 6: their affine couplings train on the hand-written kernels --
    bgk_coupling_affine_dense_h2_train forward, bgk_affine_backward + bgk_dense_backward_dx + bgk_mlp_weight_grad backward).  One step =
    zero_grad -> flow forward -> target energy with the [sum, n] loss sums -> backward -> FlatAdam step; HIP-event timed; with
    BGK_BENCH_AB=1 the same step once more with the affine couplings on the layer-by-layer / library-GEMM path of rounds 1 - 5."""
    from bgflow_amd import dense, dp
    from bgflow_amd.training import FlatAdam
    gen, sampler, desc = make_workload(name, dev)
    g = torch.Generator(device=dev).manual_seed(1234)
    zks = [sampler(batch, g) for _ in range(N_INPUT_SETS)]
    opt = FlatAdam(list(gen.flow.parameters()), lr=1e-5)
    last, k = [None], [0]

    def step():
        z = zks[k[0] % len(zks)]
        k[0] += 1
        opt.zero_grad()
        *x, dlogp = gen.flow(*z)
        loss = dp.global_kl_mean(gen._target, x, dlogp, drop_nonfinite=True)
        opt.backward(loss)
        opt.allreduce_gradients()
        opt.step()
        last[0] = loss
    step()
    torch.cuda.synchronize(dev)
    ms = event_ms_per_call(step, steps, 1)
    out = dict(workload=desc, steps_per_s=1e3 / ms, samples_per_s=batch * 1e3 / ms, ms_per_step=ms, batch=batch, steps=steps, timer="HIP events",
               loss=float(last[0].detach()), skipped_steps=opt.skipped_steps(),
               note="affine couplings: one-launch training forward (both conditioner networks on the f16 matrix cores + the affine tail; saves the "
                    "hidden layers' pre-activations and the networks' outputs: bgk_coupling_affine_dense_fwd64_train for cfg 2's 64-unit networks, "
                    "bgk_coupling_affine_dense_h2_train for cfg 5's 128-unit ones); backward: cfg 2 -- ONE call per coupling, "
                    "bgk_affine_coupling_backward64 (the tail's backward inside the scale network's launch; per network the input-gradient "
                    "chain and the weight gradients in one launch, g_z on chip); cfg 5 -- bgk_affine_backward + per network "
                    "bgk_mlp_backward_dx + bgk_mlp_weight_grad; no library GEMM, no aten activation kernel in the step")
    if os.environ.get("BGK_BENCH_AB") == "1":
        try:
            dense.AFFINE_TRAIN = False
            step()
            torch.cuda.synchronize(dev)
            ms0 = event_ms_per_call(step, max(2, steps // 2), 1)
            out["layer_by_layer_path"] = dict(ms_per_step=ms0, steps_per_s=1e3 / ms0,
                                              note="AFFINE_TRAIN = False: conditioners layer by layer, backward through F.linear / bmm (hipBLASLt) + aten "
                                                   "activation kernels -- the path of rounds 1 - 5")
        finally:
            dense.AFFINE_TRAIN = True
    return out


def measured_traffic(kernel, batch):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/<round>_traffic.json, written by
    tools/profile_round.sh: FETCH_SIZE x 2 [gfx950 rule] + WRITE_SIZE), scaled to the batch, or (None, None)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    for f in reversed(files):
        try:
            v = json.load(open(f)).get(kernel, {}).get("hbm_bytes_per_launch")
        except Exception:
            v = None
        if v is not None:
            return v * (batch / float(1 << 20)), os.path.relpath(f, ROOT)
    return None, None


def pmc_traffic(kernel, args):
    """HBM bytes per launch of the dominant kernel MEASURED IN THIS RUN: two short passes of the same workload under rocprofv3 --pmc
    (FETCH_SIZE and WRITE_SIZE in separate passes, counters only -- never combined with a trace domain), collected and corrected as
    MI355X_MICROARCH.md prescribes (KiB units; FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950 -> x 2).
    Returns (bytes per launch, note) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--batch", str(args.batch), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras", "--kl-steps", "0"] + (["--gemm", args.gemm] if args.gemm else [])
    means = {}
    root = tempfile.mkdtemp(prefix="bgk_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(root, counter)
            r = subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--"] + cmd,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=root, timeout=600)
            vals = []
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter and kernel in row["Kernel_Name"]:
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, f"no {counter} rows for {kernel} (rocprofv3 exit code {r.returncode})"
            means[counter] = (sum(vals) / len(vals), len(vals))
    except Exception as e:        # a side measurement must never take the headline line down
        return None, repr(e)[:200]
    finally:
        shutil.rmtree(root, ignore_errors=True)
    hbm = 1024.0 * (2.0 * means["FETCH_SIZE"][0] + means["WRITE_SIZE"][0])
    return hbm, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --workload {args.workload} "
                 f"--batch {args.batch} --steps 2`, mean over {means['FETCH_SIZE'][1]} launches of {kernel}; bytes = 1024 (2 FETCH_SIZE + WRITE_SIZE) "
                 "[KiB units, gfx950 x2 rule for wide coalesced reads]")


def host_cpu():
    model, cores = "unknown CPU", set()
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                phys = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_phys = min(len(cores), avail) if cores else max(1, avail // 2)
    return model, n_phys, avail


def cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max / v1 cfs quota), or None if unlimited.  On the pool's
    boxes the container sees all 256 logical CPUs but is throttled to 16: more runnable threads than that only add throttling stalls
    (one 8-thread process alone: 5.9e4 samples/s; eight of them concurrently: 7.6e4 in total)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _cpu_flow(workload):
    from bgflow_amd import configs
    return {"cfg3": configs.make_ala2_spline_generator, "cfg5": configs.make_ala2_augmented_generator,
            "cfg2": configs.make_affine8_generator}[workload]()


def _cpu_worker(job):
    """one process of a multi-process torch-CPU leg, pinned to its own cores: seconds per pass (best of `reps`) over its own chunk.
    mode "fwd": sampling direction (forward + log|det J|); "inv": NLL direction; "kl": one KL training step (forward with autograd
    through the reference's op chain, backward, torch.optim.Adam step)"""
    workload, chunk, threads, seed, cores, mode, reps = job
    if cores:
        try:
            os.sched_setaffinity(0, set(cores))
        except OSError:
            pass
    torch.set_num_threads(threads)
    from oracle import torch_flow as tfl
    gen = _cpu_flow(workload)
    g = torch.Generator().manual_seed(seed)
    dims = {"cfg3": (17, 17, 17, 9), "cfg5": (17, 17, 17, 9, 66), "cfg2": (64,)}[workload]
    ut = [torch.rand(chunk, d, generator=g) if workload != "cfg2" else torch.randn(chunk, d, generator=g) for d in dims]
    if mode == "inv":
        xs, _ = tfl.run_flow(gen.flow, ut)
        run = lambda: tfl.run_flow(gen.flow, xs, inverse=True)         # noqa: E731
    elif mode == "kl":
        params = [p for p in gen.flow.parameters()]
        opt = torch.optim.Adam(params, lr=1e-5)

        def run():
            opt.zero_grad()
            xs, dl = tfl.run_flow(gen.flow, ut, grad=True)
            loss = (gen._target.energy(*xs) - dl)
            loss = loss[torch.isfinite(loss)].mean()
            loss.backward()
            opt.step()
    else:
        run = lambda: tfl.run_flow(gen.flow, ut)                       # noqa: E731
    run()
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        run()
        best = min(best, time.perf_counter() - t0)
    return best


def _cpu_leg(workload, mode, chunk, shapes, phys, reps, timeout=60):
    """Run `mode` with every (processes x threads) shape in `shapes`, all processes concurrently, each pinned (sched_setaffinity
    before its OpenMP pool exists) to its own block of physical cores (consecutive ids = one CCD / NUMA node on the EPYC hosts of
    the pool).  Returns the per-shape results and the name of the best one."""
    import multiprocessing as mp
    legs = {}
    for workers, threads in shapes:
        name = f"{workers}x{threads}"
        pool = None
        try:
            jobs = [(workload, chunk, threads, 1234 + i, phys[i * threads:(i + 1) * threads], mode, reps) for i in range(workers)]
            pool = mp.get_context("spawn").Pool(workers)
            res = pool.map_async(_cpu_worker, jobs).get(timeout=timeout)     # a stuck worker is killed below, never waited for
            tmax = max(res)
            legs[name] = dict(value=workers * chunk / tmax, cores=workers * threads, slowest_pass_s=tmax,
                              sample=f"{workers} process(es) x {threads} intra-op threads, each pinned to its own cores, best of {reps} passes "
                                     f"over its own chunk of {chunk} samples, all concurrently")
        except Exception as e:   # the baseline must never take the bench line down
            legs[name] = dict(value=None, error=repr(e)[:200])
        finally:
            if pool is not None:
                pool.terminate()
                pool.join()
    ok = [k for k in legs if legs[k].get("value")]
    return legs, (max(ok, key=lambda k: legs[k]["value"]) if ok else None)


def _err_stats(a, ref):
    r = np.abs(np.asarray(a, np.float64).reshape(-1) - ref.reshape(-1)) / np.maximum(np.abs(ref.reshape(-1)), 1.0)
    return dict(median=float(np.median(r)), p99=float(np.quantile(r, 0.99)), max=float(r.max()), frac_gt_1e5=float((r > 1e-5).mean()))


def cpu_baseline(workload, gen_gpu, dev, n_c_samples, kl_batch):
    """The reference's op chain on the host cores (the reference itself cannot travel to the GPU box):
       value / forward: stock torch-CPU ops (oracle/torch_flow.py), vectorised aten, all physical cores -- the best of
                        {1 x N, N/8 x 8, N/16 x 16} (processes x threads, every process pinned to its own cores), chunks of 2^15
                        samples; the rate of ONE 8-thread process running alone is reported beside it;
       inverse:         the NLL direction on the best shape;   kl: KL training steps (autograd through the same op chain + Adam);
       c_oracle:        the scalar C oracle (OpenMP over samples; the parity checker, latency-bound by design);
       parity_sample:   the GPU path, the f32 C oracle and the torch f32 chain against the f64 oracle on 2^14 random samples."""
    from bgflow_amd import configs
    from oracle import flow_oracle as fo
    from oracle import oracle as orc
    from oracle import torch_flow as tfl
    orc.build()
    model, n_phys, n_avail = host_cpu()
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(n_avail))
    phys = avail[:n_phys]               # logical ids 0 .. n_phys - 1 are distinct physical cores on the pool's hosts (siblings: + n_phys)
    if workload == "cfg3":
        gen = configs.make_ala2_spline_generator()
        dims = (17, 17, 17, 9)
        rng = np.random.default_rng(1234)
        u = [rng.random((n_c_samples, d), dtype=np.float32) for d in dims]
    elif workload == "cfg5":
        gen = configs.make_ala2_augmented_generator()
        dims = (17, 17, 17, 9, 66)
        rng = np.random.default_rng(1234)
        u = [rng.random((n_c_samples, d), dtype=np.float32) for d in dims]
    else:
        gen = configs.make_affine8_generator()
        rng = np.random.default_rng(1234)
        u = [rng.standard_normal((n_c_samples, 64), dtype=np.float32)]
    chunk = 1 << 15
    quota = cpu_quota()
    usable = n_phys if quota is None else max(1, min(n_phys, int(round(quota))))     # threads worth running at once
    shapes = [(1, usable)]
    for t in (8, 4):
        if usable // t > 1:
            shapes.append((usable // t, t))
    if quota is not None and 2 * usable <= n_phys:
        shapes.append((2 * usable // 8, 8))        # mild oversubscription of the quota (throttling vs idle time: measured, not assumed)
    legs, pick = _cpu_leg(workload, "fwd", chunk, shapes, phys, reps=2)
    alone, _ = _cpu_leg(workload, "fwd", chunk, [(1, min(8, n_phys))], phys, reps=2)
    legs["1x8_alone"] = alone.get(f"1x{min(8, n_phys)}")
    if pick is None:
        torch_leg = dict(value=None, unit="samples/s", cores=n_phys, kind="port", sample="torch-CPU leg failed", torch_cpu_configurations=legs)
        best_shape = (1, n_phys)
    else:
        best_shape = tuple(int(v) for v in pick.split("x"))
        torch_leg = dict(value=legs[pick]["value"], unit="samples/s", cores=legs[pick]["cores"], kind="port",
                         sample=f"forward + log|det J|, f32, torch.no_grad; oracle/torch_flow.py = the reference's op chain on stock aten ops "
                                f"(coordinate transform: C oracle); host: {n_phys} physical cores, container CPU quota "
                                f"{'none' if quota is None else quota} CPUs; best of the shapes tried ({pick}): " + legs[pick]["sample"],
                         torch_cpu_configurations=legs)
    # ---- the other half of the metric on the host cores: NLL direction and KL training steps
    extra = {}
    if workload != "cfg2":
        inv, p2 = _cpu_leg(workload, "inv", chunk, [best_shape], phys, reps=2)
        if p2:
            extra["inverse"] = dict(value=inv[p2]["value"], unit="samples/s", cores=inv[p2]["cores"], kind="port",
                                    sample="NLL direction (xyz -> IC -> cdf maps -> 16 inverse couplings), " + inv[p2]["sample"])
        klc = 1 << 13
        kl, p3 = _cpu_leg(workload, "kl", klc, [best_shape], phys, reps=1)
        if p3:
            sps = kl[p3]["value"]
            extra["kl"] = dict(samples_per_s=sps, steps_per_s_at_gpu_batch=sps / kl_batch, gpu_batch=kl_batch, unit="steps/s", cores=kl[p3]["cores"],
                               kind="port",
                               sample=f"KL step = forward with autograd through the reference's op chain (torch ops incl. IC -> xyz) + backward + "
                                      f"torch.optim.Adam; {best_shape[0]} data-parallel process(es) x {best_shape[1]} threads on chunks of {klc} samples "
                                      f"({kl[p3]['slowest_pass_s']:.2f} s per step); steps/s at the GPU's batch = aggregate samples/s / {kl_batch}")
    # ---- C oracle
    fo.run_flow(gen.flow, [v[:256] for v in u], dtype=np.float32)
    passes, dt = 0, 0.0
    while dt < 6.0 and passes < 4:
        t0 = time.perf_counter()
        fo.run_flow(gen.flow, u, dtype=np.float32)
        dt += time.perf_counter() - t0
        passes += 1
    c_leg = dict(value=passes * n_c_samples / dt, unit="samples/s", cores=orc.num_threads(), kind="port",
                 sample=f"{passes} pass(es) over {n_c_samples} samples ({dt:.1f} s); scalar C restatement (oracle/bgo_oracle.c, k-ordered fmaf "
                        f"chains, OpenMP over samples): the bit-exact parity checker, not a throughput reference")
    # ---- parity sample: GPU path vs the oracles, layer by layer on the oracle's inputs, and whole flow vs the f64 oracle
    parity = None
    if workload == "cfg3" and gen_gpu is not None:
        import bgflow_amd as bg
        gen_gpu = configs.make_ala2_spline_generator(dev)     # fresh weights (the KL extra has stepped the bench's generator)
        n = min(1 << 14, n_c_samples)
        v = [w[:n] for w in u]
        pb, trace = [], []
        _, dl32 = fo.run_flow(gen.flow, v, dtype=np.float32, per_block=pb, trace=trace)
        _, dl64 = fo.run_flow(gen.flow, [w.astype(np.float64) for w in v], dtype=np.float64)
        _, dlt = tfl.run_flow(gen.flow, [torch.as_tensor(w) for w in v])
        n_mis = n_el = 0
        far = 0.0
        k = 0
        with torch.no_grad():
            for i, block in enumerate(gen_gpu.flow):
                if not isinstance(block, bg.CouplingFlow):
                    continue
                block.transformer.return_bin_indices = True
                ins = v if i == 0 else pb[i - 1][0]
                block(*[torch.as_tensor(w).to(dev) for w in ins])
                idx = block.transformer.last_bin_indices.cpu().numpy()
                block.transformer.return_bin_indices = False
                det = trace[k]; k += 1
                mis = idx != det["bin_idx"]
                n_mis += int(mis.sum()); n_el += mis.size
                if mis.any():
                    y = np.asarray(ins[block.transformed_indices[0]])
                    far = max(far, float(np.abs(det["knots"] - y[..., None]).min(-1)[mis].max()))
            *_, dl = gen_gpu.flow(*[torch.as_tensor(w).to(dev) for w in v])
        s_gpu, s_c32, s_t32 = _err_stats(dl.cpu().numpy(), dl64), _err_stats(dl32, dl64), _err_stats(dlt.numpy(), dl64)
        parity = dict(samples=n, elements=n_el, bin_index_differences=n_mis, tie_rate=n_mis / max(n_el, 1),
                      bin_index_policy="tie-level in the shipped gemm_mode f16x2 (a difference only where the input lies within rounding distance of a knot); "
                                       "bit-exact in gemm_mode f32 (exact_f32_mode leg) and in the generic kernels (bgk_rqs_transform)",
                      max_distance_to_knot_of_differences=far,
                      dlogp_rel_vs_f64_oracle=dict(gpu=s_gpu, c_oracle_f32=s_c32, torch_cpu_f32_chain=s_t32,
                                                   gpu_frac_over_reference_chain=s_gpu["frac_gt_1e5"] / max(s_t32["frac_gt_1e5"], 1e-12),
                                                   gpu_frac_over_c_oracle=s_gpu["frac_gt_1e5"] / max(s_c32["frac_gt_1e5"], 1e-12)),
                      note="per-sample |dlogp - dlogp64| / max(|dlogp64|, 1) on uniform prior samples (icdf tails included): GPU (shipped mode), "
                           "the f32 C oracle (icdf evaluated through f64) and the reference's f32 op chain on torch-CPU, all against the f64 oracle; "
                           "frac_gt_1e5 = share of samples beyond the north-star tolerance (no f32 evaluation holds it on every sample); bin "
                           "indices: every coupling fed the oracle's inputs, a difference is legal only within rounding distance (<= 2.4e-7) of a knot")
    return dict(torch_leg, cpu=model, logical_cpus=n_avail, physical_cores=n_phys, cgroup_cpu_quota=quota, c_oracle=c_leg, parity_sample=parity,
                **extra)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--gemm", default=None, choices=["f32", "f16x2", "bf16"],
                    help="conditioner GEMM mode of the fused coupling kernel (default: bgflow_amd.dense.GEMM_MODE)")
    ap.add_argument("--batch", type=int, default=None,
                    help="samples per GPU per step (default: 2^20; with --gpus 8: 2^19 = BASELINE cfg 4, 2^22 samples sharded over 8 GPUs)")
    ap.add_argument("--pmc", action="store_true",
                    help="measure roofline.traffic in this run: two extra short passes of this workload under rocprofv3 --pmc (FETCH_SIZE, "
                         "WRITE_SIZE; separate passes, no trace domain)")
    ap.add_argument("--cpu-samples", type=int, default=1 << 16, help="samples of the C-oracle leg of cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the exact-f32, cfg-2 and cfg-5 side measurements")
    ap.add_argument("--extra-steps", type=int, default=10, help="timed steps of every side measurement")
    ap.add_argument("--kl-steps", type=int, default=10, help="extra: time this many KL-loss training steps (0 = skip)")
    ap.add_argument("--kl-batch", type=int, default=1 << 18, help="samples per GPU per KL step")
    args = ap.parse_args()
    cfg4 = args.batch is None and default_batch(args.gpus, args.workload)[1]
    if args.batch is None:
        args.batch = default_batch(args.gpus, args.workload)[0]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    from bgflow_amd import dp
    from bgflow_amd import dense as _dense
    from bgflow_amd.flow import CouplingFlow
    # BGK_BENCH_TEST_SHARED_GPU=1: self-test of the multi-rank code path on a ONE-GPU box (all ranks on cuda:0, gloo for
    # the collectives -- RCCL refuses two ranks on one device).  Never set by the driver; numbers of such a run mean nothing.
    shared_gpu_test = os.environ.get("BGK_BENCH_TEST_SHARED_GPU") == "1"
    rank, world, local = dp.init_from_env("gloo" if shared_gpu_test else "nccl")
    if shared_gpu_test:
        local = 0
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    rccl = None
    if world > 1:
        # what the collectives really run on: backend, world size and the device behind every rank (gathered over the group)
        props = torch.cuda.get_device_properties(dev)
        mine = dict(rank=rank, local_rank=local, device=torch.cuda.get_device_name(dev), index=dev.index,
                    uuid=str(getattr(props, "uuid", "")), pci_bus_id=getattr(props, "pci_bus_id", None), host=socket.gethostname())
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, mine)
        # distinct devices = distinct (host, device index) pairs -- what LOCAL_RANK selects.  The UUIDs / PCI bus ids are recorded too, but
        # only reported: a runtime that hands out the same (or an empty) UUID for every device must not stop a correct run.
        rccl = dict(backend=torch.distributed.get_backend(), world_size=torch.distributed.get_world_size(), devices=gathered,
                    distinct_devices=len({(g["host"], g["index"]) for g in gathered}),
                    distinct_uuids=len({(g["host"], g["uuid"]) for g in gathered if g["uuid"]}),
                    distinct_pci_bus_ids=len({(g["host"], g["pci_bus_id"]) for g in gathered if g["pci_bus_id"] is not None}))
        if rccl["distinct_devices"] < world and not shared_gpu_test:
            raise SystemExit(f"bench.py: {world} ranks on {rccl['distinct_devices']} distinct GPUs ({gathered}): a multi-GPU number needs one device "
                             "per rank (LOCAL_RANK / visible devices are wrong)")
    if args.gemm:
        _dense.GEMM_MODE = args.gemm
    gemm_mode = _dense.GEMM_MODE
    gen, sampler, desc = make_workload(args.workload, dev)
    g = torch.Generator(device=dev).manual_seed(dp.rank_seed(1234, rank))
    zsets = [sampler(args.batch, g) for _ in range(N_INPUT_SETS)]
    zs = zsets[0]

    # ---- headline: W warm-up steps, then exactly K timed steps bracketed by barrier + synchronize, max over ranks
    for w in range(args.warmup):
        timed_steps(gen, zsets, 1, first=w)
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    evs = timed_steps(gen, zsets, args.steps, first=args.warmup)
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed = elapsed_local
    per_rank = [args.batch * args.steps / elapsed_local]
    if world > 1:
        tl = torch.tensor([elapsed_local], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(tl) for _ in range(world)]
        torch.distributed.all_gather(allt, tl)
        elapsed = max(float(v.item()) for v in allt)
        per_rank = [args.batch * args.steps / float(v.item()) for v in allt]
    solo = rank == 0 and world == 1
    E, W = args.extra_steps, 2
    if world > 1:
        # same-run single-rank pass (the other ranks idle at the barrier): weak-scaling efficiency = (value_N / N) / value_1
        torch.distributed.barrier()
        if rank == 0:
            ms1 = event_ms_per_call(flow_pass(gen, zsets), max(3, args.steps // 2), 1)
            rccl["one_rank_alone_samples_per_s"] = args.batch / (1e-3 * ms1)
        torch.distributed.barrier()
        # the only data-path collective of a KL / NLL evaluation, alone: all-reduce of the [sum loss, n] pair
        pair = torch.zeros(2, dtype=torch.float64, device=dev)
        for _ in range(5):
            torch.distributed.all_reduce(pair)
        torch.cuda.synchronize(dev)
        t_ar = time.perf_counter()
        for _ in range(50):
            torch.distributed.all_reduce(pair)
        torch.cuda.synchronize(dev)
        rccl["loss_pair_allreduce_us"] = 1e6 * (time.perf_counter() - t_ar) / 50

    # ---- the other direction of the path: x -> (xyz -> IC, cdf maps, 16 inverse couplings) -> z, log|det J| (the NLL direction)
    inverse_leg = None
    if args.workload != "cfg2" and rank == 0:
        try:
            with torch.no_grad():
                xsets = [gen.flow(*z)[:-1] for z in zsets]
            ms = event_ms_per_call(flow_pass(gen, xsets, inverse=True), E, W)
            inverse_leg = dict(value=args.batch / (1e-3 * ms), unit="samples/s", ms_per_step=ms, steps=E, batch=args.batch, timer="HIP events",
                               segments=[lbl for lbl, _ in gen.flow.segments(inverse=True)],
                               note="inverse (NLL) direction of the same flow on this rank: Flow.forward(x, inverse=True) -> (z, dlogp)")
            del xsets
        except Exception as e:      # a side measurement must never take the headline line down
            inverse_leg = dict(error=repr(e)[:300])

    # ---- extra: the same workload with the conditioner GEMMs in exact-f32 MFMA mode (bit-identical to the CPU oracle)
    exact = None
    if args.workload == "cfg3" and gemm_mode != "f32" and solo and not args.no_extras:
        try:
            _dense.GEMM_MODE = "f32"
            ms = event_ms_per_call(flow_pass(gen, zsets), E, W)
            exact = dict(gemm="f32", value=args.batch / (1e-3 * ms), unit="samples/s", ms_per_step=ms, steps=E, timer="HIP events",
                         note="same flow, conditioner GEMMs on the f32-input MFMA (exact fma chain): bit-identical to the CPU oracle")
            _dense.GEMM_MODE = gemm_mode
            flow_pass(gen, zs)()      # re-pack for the headline mode
            torch.cuda.synchronize(dev)
        except Exception as e:      # a side measurement must never take the headline line down
            exact = dict(error=repr(e)[:300])
            _dense.GEMM_MODE = gemm_mode

    # ---- extra: BASELINE.json configs[1] (8 affine coupling blocks, dim 64, batch 2^20)
    cfg2 = None
    if args.workload == "cfg3" and solo and not args.no_extras:
        try:
            gen2, sampler2, desc2 = make_workload("cfg2", dev)
            g2 = torch.Generator(device=dev).manual_seed(1234)
            z2 = [sampler2(1 << 20, g2) for _ in range(N_INPUT_SETS)]
            ms = event_ms_per_call(flow_pass(gen2, z2), E, W)
            cfg2 = dict(workload=desc2, value=(1 << 20) / (1e-3 * ms), unit="samples/s", ms_per_step=ms, steps=E, batch=1 << 20,
                        timer="HIP events",
                        hbm_view=dict(algorithmic_bytes_per_sample=ALG_BYTES["cfg2"], achieved_GBs=ALG_BYTES["cfg2"] * (1 << 20) / (1e-3 * ms) / 1e9,
                                      peak_GBs=HBM_PEAK_GBS, frac=ALG_BYTES["cfg2"] * (1 << 20) / (1e-3 * ms) / 1e9 / HBM_PEAK_GBS),
                        note="fused affine coupling kernel (both conditioner MLPs on the f16 matrix cores + affine tail), 8 launches")
            del gen2, z2
        except Exception as e:      # a side measurement must never take the headline line down
            cfg2 = dict(error=repr(e)[:300])

    # ---- extra: BASELINE.json configs[4] (augmented flow, fp32 vs bf16, batch 2^20)
    cfg5 = None
    if args.workload == "cfg3" and solo and not args.no_extras:
        try:
            gen5, sampler5, desc5 = make_workload("cfg5", dev)
            g5 = torch.Generator(device=dev).manual_seed(1234)
            z5 = [sampler5(1 << 20, g5) for _ in range(N_INPUT_SETS)]
            legs = {}
            for mode in (gemm_mode if gemm_mode != "bf16" else "f16x2", "bf16"):
                _dense.GEMM_MODE = mode
                ms = event_ms_per_call(flow_pass(gen5, z5), E, W)
                legs["bf16" if mode == "bf16" else "f32"] = dict(
                    gemm=mode, value=(1 << 20) / (1e-3 * ms), unit="samples/s", ms_per_step=ms, steps=E,
                    hbm_view_frac=ALG_BYTES["cfg5"] * (1 << 20) / (1e-3 * ms) / 1e9 / HBM_PEAK_GBS)
            _dense.GEMM_MODE = gemm_mode
            legs["bf16"]["note"] = ("REDUCED PRECISION leg: bf16 weights + GEMM inputs in the 10 spline layers (f32 accumulate; knots, bin search, "
                                    "log-det f32); the 6 affine layers stay split-f16")
            cfg5 = dict(workload=desc5, batch=1 << 20, timer="HIP events", **legs)
            del gen5, z5
            flow_pass(gen, zs)()
            torch.cuda.synchronize(dev)
        except Exception as e:      # a side measurement must never take the headline line down
            cfg5 = dict(error=repr(e)[:300])
            _dense.GEMM_MODE = gemm_mode

    # ---- extra (second half of BASELINE.json's metric): KL-loss training steps/s.  One step = kldiv(B).mean() -> backward through
    # the hand-written backward kernels -> ONE all-reduce of [sum, n] (+ one flat gradient bucket) -> optimizer step.
    kl = None
    if args.kl_steps > 0 and args.workload != "cfg2":
        try:
            from bgflow_amd.training import FlatAdam
            params = [p for p in gen.flow.parameters()]
            opt = FlatAdam(params, lr=1e-5)            # flat parameter / gradient bucket, bgk_adam_step (what KLTrainer uses)
            zks = [sampler(args.kl_batch, g) for _ in range(N_INPUT_SETS)]
            last, kcount = [None], [0]

            def kl_step():
                zk = zks[kcount[0] % len(zks)]
                kcount[0] += 1
                opt.zero_grad()
                *x, dlogp = gen.flow(*zk)
                loss = dp.global_kl_mean(gen._target, x, dlogp, drop_nonfinite=True)      # loss sums inside the target-energy kernel
                opt.backward(loss)                     # loss.backward() with the weight gradients accumulated straight into the bucket
                opt.allreduce_gradients()              # ONE collective on the bucket
                opt.step()                             # skips itself on the device if a gradient is NaN
                last[0] = loss
            kl_step()
            torch.cuda.synchronize(dev)
            if world > 1:
                torch.distributed.barrier()
                # the step's two collectives, each alone: the [sum loss, n] pair (timed above) and the flat gradient bucket
                for _ in range(3):
                    opt.allreduce_gradients()
                torch.cuda.synchronize(dev)
                t_ar = time.perf_counter()
                for _ in range(20):
                    opt.allreduce_gradients()
                torch.cuda.synchronize(dev)
                rccl["grad_bucket_allreduce_us"] = 1e6 * (time.perf_counter() - t_ar) / 20
                rccl["grad_bucket_bytes"] = int(opt.grad.numel() * 4)
                torch.distributed.barrier()
            ms = event_ms_per_call(kl_step, args.kl_steps, 1)
            if world > 1:
                tm = torch.tensor([ms], dtype=torch.float64, device=dev)
                torch.distributed.all_reduce(tm, op=torch.distributed.ReduceOp.MAX)
                ms = float(tm.item())
            kl = dict(steps_per_s=1e3 / ms, samples_per_s=args.kl_batch * world * 1e3 / ms, ms_per_step=ms, batch_per_gpu=args.kl_batch,
                      collectives_per_step=(None if world == 1 else dict(loss_pair_allreduce_us=rccl.get("loss_pair_allreduce_us"),
                                                                         grad_bucket_allreduce_us=rccl.get("grad_bucket_allreduce_us"),
                                                                         grad_bucket_bytes=rccl.get("grad_bucket_bytes"))),
                      steps=args.kl_steps, timer="HIP events", loss=float(last[0].detach()),
                      arithmetic="forward as the headline (f32; conditioner GEMMs on f16 hi + lo operand pairs, 22 - 24 significant bits, f32 "
                                 "accumulate).  Backward GEMMs (input-gradient chain bgk_dense_backward_dx, weight gradients "
                                 "bgk_dense_weight_grad): the same f16 hi + lo split, 3 MFMAs per product, with every gradient operand under a "
                                 "power-of-two scale taken from its largest magnitude (per tensor, published by the kernel that wrote it; per "
                                 "32-sample tile inside the chain) -- 22 significant bits per product whatever the loss scale; rounds 1 - 4 "
                                 "multiplied bf16 hi + lo pairs there (16 bits per product; flat-gradient error vs f64 1.3e-4, now 2.7e-5 over all 2^18 samples, "
                                 "tests/test_gpu_slow.py; tests/test_gpu_round4.py::test_kl_gradient_at_the_bench_batch pins chunks at <= 5e-5 -- the distance was the "
                                 "IC backward's treatment of clamped placements, not these GEMMs); spline VJP, activation derivatives, "
                                 "coordinate-transform backward: f32 with hardware exp2 / log2 / rcp / sin / cos forms (1 ulp)",
                      note="fwd: one-launch coupling layers (training variant: saves the pre-activations z0, z1; the spline parameters are "
                           "not written) + IC / CDF kernels; bwd: bgk_coupling_rqs_dense_h2_backward (the parameters recomputed from z1 on the "
                           "matrix cores, element VJP from LDS) / bgk_ic_ic2xyz_backward, conditioner input-gradient chain on bgk_dense_backward_dx, "
                           "weight / bias gradients on bgk_dense_weight_grad; one all-reduce of [sum, n] + one all-reduce of the flat gradient bucket; "
                           "bgk_adam_step (device-side NaN skip, trainers.py:198-201)")
            # the same step as ONE call of the generator: z drawn inside the step by the counter-based prior (one launch of
            # bgk_philox_fields; key = torch seed of this rank), flow, target energy with the loss sums formed in its kernel
            prior = gen._prior
            if world == 1 and hasattr(prior, "sample_fused"):
                had = prior.sample_fused
                try:
                    prior.sample_fused = True
                    torch.manual_seed(dp.rank_seed(1234, rank))

                    def kl_call():
                        opt.zero_grad()
                        loss = gen.kldiv_mean(args.kl_batch, drop_nonfinite=True)
                        opt.backward(loss)
                        opt.allreduce_gradients()
                        opt.step()
                        last[0] = loss
                    # two forms of that call: the target energy + loss sums as their own launch behind the flow (bgk_energy_fields), and
                    # formed INSIDE the generation tail's launch (bgk_icdf_ic2xyz_uni_train_kl: SURVEY f-3's single-pass kldiv, the default)
                    for leg, epilogue in (("single_call", False), ("single_pass", True)):
                        gen.flow.FUSE_KL_EPILOGUE = epilogue
                        kl_call()
                        torch.cuda.synchronize(dev)
                        ms1 = event_ms_per_call(kl_call, max(2, args.kl_steps // 2), 1)
                        kl[leg] = dict(steps_per_s=1e3 / ms1, ms_per_step=ms1, steps=max(2, args.kl_steps // 2), loss=float(last[0].detach()),
                                       note="gen.kldiv_mean(B) per step: Philox prior sample inside the step (sample_fused=True), flow, "
                                            + ("target energy + [sum, n] loss sums inside the generation tail's launch (bgk_icdf_ic2xyz_uni_train_kl)"
                                               if epilogue else "bgk_energy_fields with the [sum, n] loss sums")
                                            + ", backward, Adam")
                except Exception as e:
                    kl["single_call"] = dict(error=repr(e)[:300])
                finally:
                    prior.sample_fused = had
                    gen.flow.FUSE_KL_EPILOGUE = True
        except Exception as e:      # single-process runs only: with several ranks a failing rank cannot be papered over
            if world > 1:
                raise
            kl = dict(error=repr(e)[:300])

    # ---- extra: the KL step of the configs with affine couplings (BASELINE configs[1] at 2^20, configs[4] at 2^18)
    kl_cfg2 = kl_cfg5 = None
    if args.kl_steps > 0 and args.workload == "cfg3" and solo and not args.no_extras:
        for nm, bt in (("cfg2", 1 << 20), ("cfg5", 1 << 18)):
            try:
                leg = kl_side_leg(nm, dev, bt, max(2, args.kl_steps // 2))
            except Exception as e:      # a side measurement must never take the headline line down
                leg = dict(error=repr(e)[:300])
            if nm == "cfg2":
                kl_cfg2 = leg
            else:
                kl_cfg5 = leg
            torch.cuda.empty_cache()

    total_samples = args.batch * world * args.steps
    value = total_samples / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    if rank == 0:
        segs = gen.flow.segments()          # the blocks; a tail of icdf maps + IC -> xyz counts as one (fused) segment
        n_blocks = len(segs)
        block_ms = [0.0] * n_blocks
        for i, e0, e1 in evs:
            block_ms[i] += e0.elapsed_time(e1) / args.steps
        # coupling layers per segment: a CouplingFlow is one launch; a fused coupling stack (Split -> couplings / swaps -> Merge on
        # one buffer) holds several launches and nothing else, so its segment time / its layer count is the per-launch time
        layers = {i: ([b] if isinstance(b, CouplingFlow) else [c for c in getattr(b, "_blocks", []) if isinstance(c, CouplingFlow)])
                  for i, (_, b) in enumerate(segs)}
        coupling = [i for i in range(n_blocks) if layers[i]]
        seg_stats = {i: [coupling_stats(c, gemm_mode) for c in layers[i]] for i in coupling}
        fused = [i for i in coupling if all(st[2] is not None for st in seg_stats[i])]   # spline couplings = the dominant kernel's launches
        idxs = fused if fused else coupling
        sel = [st for i in idxs for st in seg_stats[i]]
        n_launch = len(sel)
        avg_launch_s = 1e-3 * sum(block_ms[i] for i in idxs) / n_launch
        alg_bytes_launch = sum(st[0] for st in sel) / n_launch * args.batch
        alg_bytes_step = ALG_BYTES[args.workload] * args.batch
        split = gemm_mode in ("f16x2", "bf16")
        if args.workload == "cfg2":
            kname, klabel = "coupling_affine_resident_kernel", ("coupling_affine_resident_kernel (fused: 2 DenseNets on the f16 matrix cores with both "
                                                                "networks' operands resident in LDS + affine tail; layers chained on one buffer)")
        elif gemm_mode == "f16x2":
            kname, klabel = "coupling_rqs_dense_h2v2_kernel", ("coupling_rqs_dense_h2v2_kernel (bgk_fused2.hip: DenseNet conditioner in split-f16 form on the "
                                                                "f16 matrix cores threaded through the RQ-spline / activation VALU work, one launch per coupling)")
        elif gemm_mode == "bf16":
            kname, klabel = "coupling_rqs_dense_h2_kernel", "coupling_rqs_dense_h2_kernel<bf16> (REDUCED PRECISION conditioner GEMMs)"
        else:
            kname, klabel = "coupling_rqs_dense_kernel", "coupling_rqs_dense_kernel (fused DenseNet on the f32-input MFMA + RQ-spline coupling layer)"
        traffic = traffic_note = None
        if args.pmc and world == 1:
            traffic, traffic_note = pmc_traffic(kname, args)
        if traffic is None:
            traffic, traffic_src = measured_traffic(kname, args.batch)
            if traffic is not None:
                traffic_note = (f"{traffic_src}: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE passes of this command, per launch, scaled to "
                                f"the batch -- a committed profile constant, not measured in this run (python bench.py --pmc measures it)"
                                + (f"; the --pmc passes of this run failed: {traffic_note}" if traffic_note else ""))
        roof = dict(bound="hbm", achieved=alg_bytes_launch / avg_launch_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                    traffic=traffic,
                    traffic_source=traffic_note,
                    kernel=klabel, launches_per_step=n_launch, avg_launch_ms=1e3 * avg_launch_s,
                    algorithmic_bytes_per_launch=alg_bytes_launch,
                    note="achieved = SURVEY 8(d) algorithmic bytes of a coupling layer (4 (P + 2 d + 2) B per sample: conditioner output "
                         "materialised once -- the fused kernel keeps it on chip, so `traffic` is ~8x smaller) x samples per launch / average "
                         "launch duration (HIP events around each coupling block on the launch stream)")
        roof["frac"] = roof["achieved"] / roof["peak"]
        flops = [st[2] for st in sel if st[2] is not None]
        if flops:
            tiles = (args.batch + 31) // 32
            ex = sum(flops) / len(flops) * tiles
            peak = MFMA_F16_PEAK_TFLOPS if split else MFMA_F32_PEAK_TFLOPS
            roof["mfma_util"] = dict(executed_flops_per_launch=ex, achieved=ex / avg_launch_s / 1e12, peak=peak, unit="TFLOP/s",
                                     frac=ex / avg_launch_s / 1e12 / peak,
                                     algorithmic_flops_per_launch=2.0 * sum(st[1] for st in sel) / n_launch * args.batch,
                                     note="executed = matrix-core instructions the kernel issues per 32-sample tile x their flops (split-f16: 3 "
                                          "MFMAs per product, padded tiles included); peak = dense " + ("f16" if split else "f32-input") + " MFMA")
        roof["step_view"] = dict(algorithmic_bytes_per_step=alg_bytes_step, achieved_GBs=alg_bytes_step / (1e-3 * ms_per_step) / 1e9,
                                 peak_GBs=HBM_PEAK_GBS, frac=alg_bytes_step / (1e-3 * ms_per_step) / 1e9 / HBM_PEAK_GBS,
                                 note="SURVEY 8(d) bytes of the WHOLE step (couplings + icdf maps + coordinate transform) / step time")
        roof["block_ms"] = [round(v, 3) for v in block_ms]
        roof["block_labels"] = [lbl for lbl, _ in segs]
        out = dict(metric="flow samples/s (fwd+log|detJ|) at batch 2^20", value=value, unit="samples/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True,
                   scaling="weak", vs_baseline=None,
                   dtype={"bf16": "bf16", "f32": "f32", "f16x2": "f32 (split-f16 conditioner GEMMs: hi+lo f16 operand pairs, f32 accumulate)"}[gemm_mode],
                   arithmetic={"bf16": "spline / log-det / coordinate arithmetic f32; conditioner GEMMs bf16 inputs, f32 accumulate (reduced precision)",
                               "f32": "f32 throughout (f32-input MFMA = exact fma chain)",
                               "f16x2": "f32 throughout, except that the conditioner GEMM operands are represented as hi + lo f16 pairs (22-24 "
                                        "significant bits, 3 MFMAs per product, f32 accumulate); hardware exp2 / log2 / rcp with Newton steps"}[gemm_mode],
                   data="synthetic",
                   inputs=f"{N_INPUT_SETS} synthetic input sets resident in HBM, rotated from step to step in every timed loop "
                          f"(a set is larger than what is left of the Infinity Cache when its turn comes again)",
                   config=dict(workload=(("cfg 4: " if cfg4 else "") + desc
                                         + ("" if world == 1 else f"; data-parallel shard: {args.batch} samples per rank x {world} ranks = "
                                            f"{args.batch * world} global"
                                            + (" = BASELINE cfg 4 (2^22 samples over 8 GPUs, RCCL all-reduce of the KL loss pair and the gradient "
                                               "bucket in the `kl` leg)" if cfg4 else
                                               " (BASELINE cfg 4 is this flow at 2^22 global = 2^19 per rank on 8 GPUs: the default of --gpus 8)"))),
                               batch_per_gpu=args.batch, global_batch=args.batch * world,
                               parallelism=f"dp{world}",
                               conditioner_gemm={"f16x2": "split-f16: f32 operands as hi+lo f16 pairs, 3 MFMAs per product, f32 accumulate "
                                                          "(f32-class accuracy: per-sample log-det within 1e-5 of the reference's f64 goldens)",
                                                 "f32": "f32-input MFMA (exact, bit-identical to the CPU oracle)",
                                                 "bf16": "REDUCED PRECISION: bf16 weights and GEMM inputs (spline layers), f32 accumulate; "
                                                         "spline / log-det arithmetic f32"}[gemm_mode]),
                   per_rank_samples_per_s=per_rank,
                   roofline=roof)
        if rccl is not None:
            if "one_rank_alone_samples_per_s" in rccl:
                rccl["scaling_efficiency"] = (value / world) / rccl["one_rank_alone_samples_per_s"]
            out["rccl"] = rccl
        if inverse_leg is not None:
            out["inverse"] = inverse_leg
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, gen, dev, args.cpu_samples, args.kl_batch)
        if exact is not None:
            out["exact_f32_mode"] = exact
        if cfg2 is not None:
            out["cfg2"] = cfg2
        if cfg5 is not None:
            out["cfg5"] = cfg5
        if kl is not None:
            out["kl"] = kl
        if kl_cfg2 is not None:
            out["kl_cfg2"] = kl_cfg2
        if kl_cfg5 is not None:
            out["kl_cfg5"] = kl_cfg5
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
