"""Builds libunivl_hip.so (hipcc, --offload-arch=gfx950) in-tree.  hipcc cross-compiles without a GPU."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libunivl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
         "-Wno-unused-result", "-fno-fast-math"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, trace=False):
    """trace=True: the measurement build (-DUNIVL_TRACE, csrc/gemm.hip) -> lib/libunivl_hip_trace.so, objects under lib/trace/; load it
    with UNIVL_LIB=<path> (scripts/mb_trace_gemm.py).  Never the product library."""
    if trace:
        return _build_variant("trace", ["-DUNIVL_TRACE"], force, verbose)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (s, r.stdout, r.stderr))
        if verbose:
            print("[build] compiled", os.path.basename(s))
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if jobs or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[build] linked", LIB)
    return LIB


def _build_variant(tag, extra, force, verbose):
    odir = os.path.join(LIBDIR, tag)
    os.makedirs(odir, exist_ok=True)
    lib = os.path.join(LIBDIR, "libunivl_hip_%s.so" % tag)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(odir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        r = subprocess.run([HIPCC] + FLAGS + extra + ["-c", job[0], "-o", job[1]], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (job[0], r.stdout, r.stderr))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if jobs or not os.path.exists(lib):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[build] linked", lib)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:        # python univl_amd/build.py --variant <tag> -DNAME=VALUE ...  ->  lib/libunivl_hip_<tag>.so (A/B builds, loaded with UNIVL_LIB=)
        i = sys.argv.index("--variant")
        _build_variant(sys.argv[i + 1], [x for x in sys.argv[i + 2:] if x.startswith("-D")], "--force" in sys.argv, True)
    else:
        build(force="--force" in sys.argv, trace="--trace" in sys.argv)
