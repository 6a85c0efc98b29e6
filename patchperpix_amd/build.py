"""Ahead-of-time build of libppp_mi355x.so (hand-written HIP kernels + C ABI) for gfx950.

hipcc cross-compiles without a GPU; the built library stays in-tree
(patchperpix_amd/csrc/libppp_mi355x.so, git-ignored) so that it travels to the GPU box.
"""
import glob
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# PPP_LIB: load / build an alternative library file (kernel experiments side by side)
LIB = os.environ.get("PPP_LIB") or os.path.join(CSRC, "libppp_mi355x.so")
ARCH = "gfx950"
# -ffp-contract=off: the float arithmetic must round exactly like the reference's
# (no fused multiply-add across the statements being restated).
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=" + ARCH,
         "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


# Kernels no dispatch reaches by default -- measured alternatives kept under test -- are compiled only
# with PPP_BUILD_EXPERIMENTS=1 (which also defines the macro the dispatch sites test):
#   ppp_consensus_v4.hip   S1 with a run's accumulators split over two waves (PPP_S1_V4=1); measured
#                          slower than the one-wave kernel
EXPERIMENTS = ("ppp_consensus_v4.hip",)


def experiments():
    return os.environ.get("PPP_BUILD_EXPERIMENTS", "0") == "1"


def sources():
    src = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(CSRC, "*.cpp")))
    if not experiments():
        src = [f for f in src if os.path.basename(f) not in EXPERIMENTS]
    return src


def _toolchain(hipcc):
    """what the object stamp records of the compiler: its path and version banner"""
    try:
        v = subprocess.run([hipcc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60).stdout
        return hipcc + " " + " ".join(v.decode(errors="replace").split())
    except (OSError, subprocess.SubprocessError):
        return hipcc


def _flags_stamp():
    return os.path.join(CSRC, "_obj", os.path.basename(LIB), "lib.flags")


def is_stale():
    if not os.path.exists(LIB):
        return True
    # a library built with other PPP_EXTRA_FLAGS is stale too
    want = " ".join((os.environ.get("PPP_EXTRA_FLAGS", "") + (" -DPPP_BUILD_EXPERIMENTS" if experiments() else "")).split())
    have = open(_flags_stamp()).read() if os.path.exists(_flags_stamp()) else ""
    if want != have:
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + \
        [os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "ppp_mi355x.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, relink=False):
    """Every translation unit is compiled on its own (in parallel, objects under csrc/_obj/),
    then linked: a full build takes as long as the slowest file instead of their sum.
    force: recompile every object; relink: link again, reusing objects that are up to date."""
    if not force and not relink and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("PPP_EXTRA_FLAGS", "").split() + (["-DPPP_BUILD_EXPERIMENTS"] if experiments() else [])
    objdir = os.path.join(CSRC, "_obj", os.path.basename(LIB))
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"]

    headers = glob.glob(os.path.join(CSRC, "*.hpp")) + \
        [os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "ppp_mi355x.h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    stamp = " ".join(cflags + extra) + " | " + _toolchain(hipcc)

    def compile_one(src):
        # an object is reused when it is newer than its source and every header and was built
        # with the same flags (PPP_EXTRA_FLAGS variants keep their own object directory)
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        flag_file = obj + ".flags"
        if not force and os.path.exists(obj) and os.path.exists(flag_file) and open(flag_file).read() == stamp and \
                os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            return obj
        cmd = [hipcc] + cflags + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(flag_file, "w") as f:
            f.write(stamp)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, sources()))
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(_flags_stamp(), "w") as f:
        f.write(" ".join(extra))
    return LIB


if __name__ == "__main__":
    import sys
    # default: rebuild what is out of date and link; --force recompiles everything
    print(build_library(force="--force" in sys.argv[1:], relink=True, verbose=True))
