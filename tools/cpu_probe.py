"""GPU box (CPU only): cgroup CPU quota and the throughput of the torch-CPU leg for a few (processes x threads) shapes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
if __name__ == "__main__":
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
        try:
            print(f, open(f).read().strip())
        except OSError as e:
            print(f, "n/a")
    print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), bench.host_cpu())
    phys = sorted(os.sched_getaffinity(0))[:64]
    for shape in ((1, 8), (2, 8), (1, 16), (4, 8), (2, 16), (1, 32), (16, 4)):
        t0 = time.time()
        legs, pick = bench._cpu_leg("cfg3", "fwd", 1 << 15, [shape], phys, reps=1, timeout=40)
        print(shape, {k: (round(v["value"]) if v.get("value") else v.get("error"), round(v.get("slowest_pass_s", 0), 2)) for k, v in legs.items()}, f"{time.time()-t0:.1f}s", flush=True)
