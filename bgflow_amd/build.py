"""Build libbgflow_amd.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU; the resulting .so stays IN-TREE (bgflow_amd/libbgflow_amd.so,
git-ignored) so that it travels to the GPU box with the repository snapshot.  Every translation unit is
compiled to its own object (in parallel, rebuilt only when it or a header changed), then linked.

    python -m bgflow_amd.build [--force]
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libbgflow_amd.so")

# -ffp-contract=off + correctly rounded div/sqrt: the f32 arithmetic of the kernels is then the same
# sequence of IEEE ops as the CPU oracle's (bit-identical spline bin indices); see csrc/bgk_detmath.h.
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
    "-Wno-comment",
]
# per-TU additions.  bgk_fused2.hip threads VALU work between MFMAs: packed-f32 ops (which the SLP vectoriser
# would form from adjacent scalar ops) do not overlap with the matrix pipe on gfx950 (tools/ubench/issue_bench).
TU_FLAGS = {"bgk_fused2.hip": ["-fno-slp-vectorize"], "bgk_fused2_train.hip": ["-fno-slp-vectorize"],
            "bgk_fused2_bf16.hip": ["-fno-slp-vectorize"], "bgk_fused2_afftrain.hip": ["-fno-slp-vectorize"],
            # bgk_tail.hip: the SLP vectoriser packs the 3-vector arithmetic of a placement into v_pk_* pairs and pays for it with
            # register shuffles around the uniformly indexed position arrays (39 instead of 15 v_mov per placement)
            "bgk_tail.hip": ["-fno-slp-vectorize"]}


INCLUDES_SOURCE = {"bgk_fused2_train.hip": ["bgk_fused2.hip"], "bgk_fused2_bf16.hip": ["bgk_fused2.hip"], "bgk_fused2_afftrain.hip": ["bgk_fused2.hip"]}     # translation units that #include another .hip


HEADER = os.path.join(HERE, "..", "include", "bgflow_amd.h")


def abi_symbols(header=HEADER):
    """names of the C-ABI prototypes declared in include/bgflow_amd.h -- the library's export list"""
    import re
    text = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
    return sorted(set(re.findall(r"^[A-Za-z_][\w \t\*]*?\b(bgk_\w+)\s*\(", text, flags=re.M)))


def _export_map():
    """linker version script: exactly the header's prototypes are dynamic symbols; everything else (launchers shared between
    translation units, option variables, hipcc's per-unit __hip_cuid_* markers) stays local"""
    path = os.path.join(OBJ, "export.map")
    with open(path, "w") as f:
        f.write("{\n  global:\n" + "".join(f"    {n};\n" for n in abi_symbols()) + "  local:\n    *;\n};\n")
    return path


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))


def _obj(src, tag):
    return os.path.join(OBJ, os.path.basename(src) + tag + ".o")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in sources() + _headers())


def build_extension(force=False, verbose=False, out=None):
    """BGK_EXTRA_FLAGS: extra hipcc flags (experiments); BGK_EXTRA_TU: comma list of sources they apply to (default: all);
    ``out``: path of the linked library (default: the in-tree libbgflow_amd.so)."""
    extra = os.environ.get("BGK_EXTRA_FLAGS", "").split()
    only = [t for t in os.environ.get("BGK_EXTRA_TU", "").split(",") if t]
    if not force and not extra and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    tag = ("." + "".join(c if c.isalnum() else "_" for c in " ".join(extra))) if extra else ""
    hdr_t = max(os.path.getmtime(p) for p in _headers())

    def compile_one(src):
        mine = extra if (not only or os.path.basename(src) in only) else []
        obj = _obj(src, tag if mine else "")
        dep_t = max([os.path.getmtime(src), hdr_t] + [os.path.getmtime(os.path.join(CSRC, d)) for d in INCLUDES_SOURCE.get(os.path.basename(src), [])])
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > dep_t:
            return obj
        cmd = [hipcc] + HIPCC_FLAGS + TU_FLAGS.get(os.path.basename(src), []) + mine + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    target = out or LIB
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + _export_map(), "-o", target] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return target


if __name__ == "__main__":
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    print(build_extension(force="--force" in sys.argv, verbose="--quiet" not in sys.argv, out=out))
