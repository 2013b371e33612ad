"""Builds the product's HIP extension in-tree: usearch12_amd/libugs.so (gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libugs.so")
LIB_RCCL = os.path.join(HERE, "libugs_rccl.so")       # include/ugs_comm.h: the RCCL gather (libugs.so itself has no RCCL dependency)
CLI = os.path.join(HERE, "ugs_cli")
SOURCES = ["ugs_host.cpp", "ugs_alloc.cpp", "ugs_writers.cpp", "ugs_cluster.cpp", "ugs_index.hip", "ugs_rank.hip", "ugs_rank_hot.hip", "ugs_rank2.hip", "ugs_rank3.hip", "ugs_align.hip", "ugs_xdrop.hip", "ugs_local.hip", "ugs_inbatch.hip", "ugs_deep.hip"]
DEPS = [x for x in SOURCES if x != "ugs_rank_hot.hip"] + ["ugs_dev.h", "ugs_host.h", "ugs_rank2.h", "ugs_ring_dev.h", "ugs_rank_keys.h", "ugs_xdrop_dev.h", os.path.join("..", "..", "include", "ugs.h")]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip"]
# per-source compiler options.  k_rank's partition loop lives on the edge of its register budget: of the machine
# schedulers LLVM offers for AMDGPU, "iterative-maxocc" gives the fastest HOT instantiation (54.5 vs 55.7 ms on C2; max-ilp 56.6,
# iterative-minreg 128) but costs the mid-identity instantiations 10 %, so the HOT kernel is a translation unit of its own
# (ugs_rank.hip, UGS_RANK_TU); with -amdgpu-schedule-relaxed-occupancy on top 52.9 ms (that option without the scheduler: 56.2).
# ugs_align.hip gains nothing from any of them and does not compile with iterative-maxocc (tools/build_hot_variant.sh, tools/ab_variants.sh)
EXTRA = {"ugs_rank.hip": ["-DUGS_RANK_TU=2"],
         "ugs_rank_hot.hip": ["-DUGS_RANK_TU=1", "-mllvm", "-amdgpu-sched-strategy=iterative-maxocc", "-mllvm", "-amdgpu-schedule-relaxed-occupancy"]}
# objects compiled from another source file's text under other options: ugs_rank_hot.o = the HOT instantiation of k_rank alone
ALIAS = {"ugs_rank_hot.hip": "ugs_rank.hip"}


KEEP_ASM = ("ugs_rank.hip", "ugs_rank_hot.hip", "ugs_rank2.hip", "ugs_rank3.hip", "ugs_xdrop.hip", "ugs_local.hip", "ugs_align.hip")      # sources whose emitted code tests/test_isa.py pins


def asm_path(src):
    """gfx950 assembly of a KEEP_ASM source as emitted by the build's own compilation (git-ignored)"""
    return os.path.join(CSRC, os.path.splitext(os.path.basename(src))[0] + ".gfx950.s")


def real_src(src):
    """the file a SOURCES entry is compiled from (ALIAS: the same text under other options)"""
    b = os.path.basename(src)
    return os.path.join(CSRC, ALIAS.get(b, b))


def asm_is_fresh(src):
    a = asm_path(src)
    return os.path.exists(a) and _mtime(a) >= _newest(real_src(src))


def flags_for(src):
    return FLAGS + EXTRA.get(os.path.basename(src), [])


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
    return max([_mtime(d) for d in _deps(src)] + [_mtime(os.path.abspath(__file__))])          # (the flags live in this file)


def csrc_hash():
    """16 hex digits over the product's source text (csrc/*.hip|cpp|h + include/*.h, names and bytes, sorted): what ties a PMC profile
    under profiles/ to the build that produced it (bench.py attaches counter traffic only when the hashes agree)"""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.join(HERE, "..", "include")
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))] + \
            [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    for f in sorted(files, key=os.path.basename):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _compile(src, obj, path, verbose):
    if src in KEEP_ASM:
        # the same compilation also leaves the device assembly behind (-save-temps): tests/test_isa.py reads it
        import shutil
        import tempfile
        tmpd = tempfile.mkdtemp(prefix="ugs_build_")
        try:
            base = os.path.splitext(os.path.basename(path))[0]              # (the temporaries are named after the input file)
            cmd = ["hipcc"] + flags_for(src) + ["-save-temps=obj", "-c", path, "-o", os.path.join(tmpd, base + ".o")]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)      # (-save-temps repeats every warning of the unused-result kind: shown only on failure)
            if r.returncode != 0:
                sys.stderr.write(r.stderr)
                raise subprocess.CalledProcessError(r.returncode, cmd)
            shutil.move(os.path.join(tmpd, base + "-hip-amdgcn-amd-amdhsa-gfx950.s"), asm_path(src))
            shutil.move(os.path.join(tmpd, base + ".o"), obj)
        finally:
            shutil.rmtree(tmpd, ignore_errors=True)
        return
    cmd = ["hipcc"] + flags_for(src) + ["-c", path, "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)


def build(force=False, verbose=False):
    from concurrent.futures import ThreadPoolExecutor
    objs, todo = [], []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        path = real_src(src)
        if force or _mtime(obj) < _newest(path):
            todo.append((src, obj, path))
    if todo:                                   # the translation units are independent: compile them side by side (hipcc is single-threaded)
        with ThreadPoolExecutor(max_workers=min(len(todo), max(1, (os.cpu_count() or 2) - 1), 8)) as ex:
            for f in [ex.submit(_compile, s_, o_, p_, verbose) for s_, o_, p_ in todo]:
                f.result()
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
