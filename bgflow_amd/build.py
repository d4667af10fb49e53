"""Build libbgflow_amd.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU; the resulting .so stays IN-TREE (bgflow_amd/libbgflow_amd.so,
git-ignored) so that it travels to the GPU box with the repository snapshot.

    python -m bgflow_amd.build [--force]
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbgflow_amd.so")

# -ffp-contract=off + correctly rounded div/sqrt: the f32 arithmetic of the kernels is then the same
# sequence of IEEE ops as the CPU oracle's (bit-identical spline bin indices); see csrc/bgk_detmath.h.
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
    "-Wno-comment",
]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build_extension(force=False, verbose=False):
    if not force and not os.environ.get("BGK_EXTRA_FLAGS") and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + os.environ.get("BGK_EXTRA_FLAGS", "").split() + ["-o", LIB] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True))
