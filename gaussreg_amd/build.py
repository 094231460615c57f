"""Build libgaussreg_hip.so (the C-ABI library of hand-written HIP kernels) for gfx950, in-tree.

    python -m gaussreg_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to gaussreg_amd/lib/obj/, the library to
gaussreg_amd/lib/libgaussreg_hip.so (git-ignored, but it travels to the GPU box with gpurun).
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libgaussreg_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the parity-critical kernels must keep separate fp32 mul/add;
# kernels that want an FMA say so with fmaf().
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(HERE, "..", "include", "gaussreg_hip.h"))
    jobs = []
    objs = []
    for src in sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return cmd, r.returncode, r.stdout

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for cmd, rc, out in ex.map(run, jobs):
                if rc != 0:
                    raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
                if verbose and out.strip():
                    print(out)
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        _, rc, out = run(cmd)
        if rc != 0:
            raise RuntimeError("link failed:\n" + out)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
