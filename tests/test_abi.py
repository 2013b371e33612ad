"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/ugs.h
declares, struct layouts match the ctypes mirror, and compute entry points fail loudly (no CPU
fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from usearch12_amd import capi
from usearch12_amd.abi import Params, HIT_DTYPE, ClusterStats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="ugs.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ugs_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), n
    assert sorted(capi.EXPORTS) == names
    assert L.ugs_abi_version() == 6


def test_rccl_library_exports_every_symbol_of_ugs_comm_h():
    names = _declared("ugs_comm.h")
    assert sorted(capi.COMM_EXPORTS) == names
    L = capi.lib_rccl()                       # loads without a GPU: librccl is only called by the entry points
    for n in names:
        assert hasattr(L, n), n


def test_struct_layouts():
    assert C.sizeof(Params) == 192
    assert HIT_DTYPE.itemsize == 80
    assert C.sizeof(ClusterStats) == 104
    p = capi.params(is_nucleo=True, id=0.97)
    assert (p.word_len, p.max_accepts, p.max_rejects, p.big, p.band, p.hsp_word_len) == (8, 1, 32, 100000, 16, 5)
    assert p.id_accept == float(np.float32(0.97))     # options are stored as float (opts.cpp:265)
    q = capi.params(is_nucleo=False, id=0.8)
    assert (q.word_len, q.hsp_word_len) == (5, 3)


def test_writers_text():
    L = capi.lib()
    h = np.zeros(1, dtype=HIT_DTYPE)
    h["target"] = 408; h["ids"] = 244; h["mism"] = 5; h["aln_len"] = 250; h["opens"] = 1; h["ql"] = 249; h["tl"] = 250
    h["cigar_off"] = 0; h["cigar_len"] = 3
    pool = np.array([(206 << 2) | 0, (1 << 2) | 2, (43 << 2) | 0], dtype=np.uint32)
    buf = C.create_string_buffer(512)
    L.ugs_format_blast6(h.ctypes.data, b"q0;src=t408", b"t408", buf, 512)
    assert buf.value == b"q0;src=t408\tt408\t97.6\t250\t5\t1\t1\t249\t1\t250\t*\t*\n"
    L.ugs_format_uc_hit(h.ctypes.data, pool.ctypes.data, 1, b"q0;src=t408", b"t408", buf, 512)
    assert buf.value == b"H\t408\t249\t97.6\t+\t0\t0\t206MI43M\tq0;src=t408\tt408\n"
    L.ugs_format_uc_nohit(250, b"q9", buf, 512)
    assert buf.value == b"N\t*\t250\t*\t.\t*\t*\t*\tq9\t*\n"


def test_no_cpu_fallback():
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    seqs = np.frombuffer(b"ACGT" * 20, dtype=np.uint8)
    offs = np.array([0, 80], dtype=np.uint64)
    with pytest.raises(capi.UgsError) as e:
        capi.UgsDB(capi.params(), seqs, offs, device=0)
    assert e.value.code == -2          # UGS_E_NODEVICE
    with pytest.raises(capi.UgsError):
        capi.UgsDB(capi.params(), seqs, offs, device=-1)   # "-1 = CPU" does not exist here


def test_cli_refuses_what_the_device_path_does_not_implement():
    """-quicksort (udbusortedsearcher.cpp:192-198: an unstable quicksort over every touched target instead of CountSort) has no
    field in the ABI and no device path: the driver refuses it (and any other unknown option) instead of ignoring it"""
    import subprocess
    cli = os.path.join(os.path.dirname(capi.LIB_PATH), "ugs_cli")
    for opt in ("-quicksort", "-nosuchoption"):
        r = subprocess.run([cli, "-usearch_global", "/nonexistent.fa", "-db", "/nonexistent2.fa", "-id", "0.97", opt], capture_output=True, text=True)
        assert r.returncode != 0 and "unknown option" in r.stderr


def test_no_process_that_loads_libugs_imports_torch():
    """torch's wheel bundles a HIP runtime, an HSA runtime and an RCCL under the system libraries' SONAMEs: a process that holds torch AND
    libugs.so runs the product on torch's ROCm 7.0 runtime (torch first) or maps two runtimes and corrupts its heap (libugs first) - round 5's
    silent abort lived there (DESIGN.md section 4 "The crash").  So: nothing that a test process, bench.py or the package imports may import
    torch; the one file that does (tests/gloo_worker.py, a child process of the gloo tests) must never load the library."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "tests", "*.py")) + glob.glob(os.path.join(root, "usearch12_amd", "*.py")) + \
        [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    offenders = []
    for f in files:
        tree = ast.parse(open(f).read())
        for node in ast.walk(tree):
            names = [a.name for a in node.names] if isinstance(node, ast.Import) else ([node.module or ""] if isinstance(node, ast.ImportFrom) else [])
            if any(n == "torch" or n.startswith("torch.") for n in names):
                offenders.append(os.path.relpath(f, root))
    assert sorted(set(offenders)) == ["tests/gloo_worker.py"], offenders
    worker = open(os.path.join(root, "tests", "gloo_worker.py")).read()
    assert "capi" not in worker.replace('"usearch12_amd.capi" not in sys.modules', "")
