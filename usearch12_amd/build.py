"""Builds the product's HIP extension in-tree: usearch12_amd/libugs.so (gfx950 only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libugs.so")
LIB_RCCL = os.path.join(HERE, "libugs_rccl.so")       # include/ugs_comm.h: the RCCL gather (libugs.so itself has no RCCL dependency)
CLI = os.path.join(HERE, "ugs_cli")
SOURCES = ["ugs_host.cpp", "ugs_writers.cpp", "ugs_cluster.cpp", "ugs_index.hip", "ugs_rank.hip", "ugs_align.hip", "ugs_xdrop.hip", "ugs_local.hip", "ugs_inbatch.hip"]
DEPS = SOURCES + ["ugs_dev.h", "ugs_host.h", "ugs_xdrop_dev.h", os.path.join("..", "..", "include", "ugs.h")]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip"]


def _mtime(path):
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _deps(path, seen=None):
    """the file and every project header it includes (transitively): a change rebuilds only the objects that see it"""
    import re
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path, errors="replace").read(), re.M):
        _deps(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def _newest(src):
    return max(_mtime(d) for d in _deps(src))


def build(force=False, verbose=False):
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _mtime(obj) < _newest(os.path.join(CSRC, src)):
            cmd = ["hipcc"] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    if force or _mtime(LIB) < max(_mtime(o) for o in objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    g_src, g_obj = os.path.join(CSRC, "ugs_gather.cpp"), os.path.join(CSRC, "ugs_gather.o")
    if force or _mtime(g_obj) < _newest(g_src):
        cmd = ["hipcc"] + FLAGS + ["-c", g_src, "-o", g_obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if force or _mtime(LIB_RCCL) < max(_mtime(g_obj), _mtime(LIB)):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_RCCL, g_obj, "-L" + HERE, "-lugs",
               "-L" + os.path.join(ROCM, "lib"), "-lrccl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    cli_src = os.path.join(CSRC, "ugs_cli.cpp")
    if force or _mtime(CLI) < max(_newest(cli_src), _mtime(LIB), _mtime(LIB_RCCL)):
        cmd = ["hipcc", "-O2", "-std=c++17", "-pthread", "-o", CLI, cli_src, "-L" + HERE, "-lugs_rccl", "-lugs", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
