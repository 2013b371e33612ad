"""Pins a property of the emitted gfx950 code the ranking kernel's fast paths depend on (ugs_rank.hip process_batch,
scan_fast8): the ordered clear-with-return of a partition's counters (ds_and_rtn_b32) is only issued after every counter
increment of the partition (ds_add_u32) has COMPLETED - an `s_waitcnt lgkmcnt(0)` stands between the two batches in every
basic block that holds both.  Reads the assembly the build's own compilation left behind (usearch12_amd/build.py keeps it) or, when that
is missing or stale, compiles the kernel to assembly with the build's options (hipcc cross-compiles without a GPU)."""
import os
import re
import subprocess

import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usearch12_amd.build import asm_is_fresh, asm_path, flags_for, real_src  # noqa: E402  (the product's own compiler options, per source)
CSRC = os.path.join(ROOT, "usearch12_amd", "csrc")
BUILD_PY = os.path.join(ROOT, "usearch12_amd", "build.py")


def _asm(name):
    """gfx950 assembly of one of the build's translation units: the file the build's own compilation left behind (build.py,
    -save-temps) when it is current, else compiled here with the build's options for that unit"""
    if asm_is_fresh(name):
        return open(asm_path(name)).read()
    src = real_src(name)
    out = "/tmp/%s_isa_%d_%d.s" % (name, int(max(os.path.getmtime(src), os.path.getmtime(os.path.join(CSRC, "ugs_xdrop_dev.h")))), int(os.path.getmtime(BUILD_PY)))
    if not os.path.exists(out):
        subprocess.check_call(["hipcc"] + [f for f in flags_for(name) if f != "-fPIC"] + ["--cuda-device-only", "-S", src, "-o", out],
                              stderr=subprocess.DEVNULL)
    return open(out).read()


def _isa():
    # k_rank is two translation units (ugs_rank.hip, UGS_RANK_TU): the HOT instantiation and everything else
    return (_asm("ugs_rank.hip") + "\n" + _asm("ugs_rank_hot.hip")).split("\n")


def test_counter_clears_wait_for_the_adds_of_their_batch():
    pending_add = False            # an LDS add of this basic block may still be in flight
    n_rtn = n_guarded = 0
    for ln in _isa():
        t = ln.strip()
        if not t or t.startswith(";"):
            continue
        if re.match(r"^[.\w$]+:", t) or t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            pending_add = False
            continue
        if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
            if pending_add:
                n_guarded += 1
            pending_add = False
        elif t.startswith("ds_add_u32") and not t.startswith("ds_add_rtn"):
            pending_add = True
        elif t.startswith("ds_and_rtn_b32"):
            n_rtn += 1
            assert not pending_add, "ds_and_rtn_b32 issued while adds of its batch may be in flight"
    assert n_rtn >= 11 * 6 and n_guarded >= 6       # the fast paths are there (three row widths x ping-pong per instantiation)


def _isa_of(name):
    return _asm(name)


def _kernel_body(isa, name):
    m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm" % name, isa, re.S | re.M)
    assert m, name
    return m.group(2)


def test_rank2g_ring_is_counted_by_hand_and_the_table_lines_load_together():
    """k_rank2g (the gather variant for sparse indexes) uses the same ring: one asm load per stage into a[4k : 4k+3], waits of
    vmcnt(3), a drain before the slots are reused, no scratch access (an array of table quads indexed at run time went to scratch
    once, one drained load per quad: pinned), and the eight quads of a row's partition-table line are requested before the first is
    used (the waits between them count down from 7)."""
    isa = _isa_of("ugs_rank2.hip")
    body = _kernel_body(isa, "k_rank2g")
    i = isa.index(".name:           _Z8k_rank2g")
    meta = isa[isa.rindex("- .agpr_count", 0, i):i + 600]
    assert re.search(r"\.vgpr_spill_count:\s*0\b", meta) and re.search(r"\.private_segment_fixed_size:\s*0\b", meta), meta
    assert re.search(r"\.agpr_count:\s*16\b", meta)
    assert "scratch_" not in body
    loads = re.findall(r"global_load_dwordx4 (a\[\d+:\d+\])", body)
    assert sorted(set(loads)) == ["a[0:3]", "a[12:15]", "a[4:7]", "a[8:11]"] and len(loads) == 8, loads
    assert len(re.findall(r"s_waitcnt vmcnt\(3\)\n\s*v_accvgpr_read_b32", body)) == 4
    assert "v_accvgpr_write" not in body and "v_accvgpr_mov" not in body
    # the table line: eight vector-register quads in a row with no wait between the requests
    pos = [m.start() for m in re.finditer(r"global_load_dwordx4 v\[", body)]
    assert len(pos) >= 8
    assert "s_waitcnt vmcnt" not in body[pos[0]:pos[7]], "the eight table quads are not requested together"
    assert re.search(r"s_waitcnt vmcnt\(7\)", body[pos[7]:pos[7] + 1500])
    assert len(re.findall(r"ds_or_rtn_b32", body)) >= 16 and "flat_atomic" not in body


def test_rank3g_ring_is_counted_by_hand_in_both_passes():
    """k_rank3g (ugs_rank3.hip: the sparse-index kernel with two filter passes) streams through the same ring in BOTH passes: posting
    loads only into a[0:3] .. a[12:15], each slot waited for with vmcnt(3) right in front of its four v_accvgpr_read, no compiler wait
    for ALL loads between a pass's first load and its drain (that would stall the ring on every stage), no scratch, no spill, no
    flat atomic; pass 1 counts with one ds_or_rtn_b32 per posting, the grouping inserts with ds_cmpst_rtn_b32."""
    isa = _isa_of("ugs_rank3.hip")
    body = _kernel_body(isa, "k_rank3g")
    i = isa.index(".name:           _Z8k_rank3g")
    meta = isa[isa.rindex("- .agpr_count", 0, i):i + 600]
    assert re.search(r"\.vgpr_spill_count:\s*0\b", meta) and re.search(r"\.private_segment_fixed_size:\s*0\b", meta), meta
    assert int(re.search(r"\.agpr_count:\s*(\d+)", meta).group(1)) <= 24 and int(re.search(r"\.vgpr_count:\s*(\d+)", meta).group(1)) <= 128
    assert "scratch_" not in body and "flat_" not in body
    loads = re.findall(r"global_load_dwordx4 (a\[\d+:\d+\])", body)
    assert sorted(set(loads)) == ["a[0:3]", "a[12:15]", "a[4:7]", "a[8:11]"] and len(loads) % 8 == 0 and len(loads) >= 16, loads
    assert "global_load_dwordx4 v" not in body
    assert len(re.findall(r"s_waitcnt vmcnt\(3\)\n\s*v_accvgpr_read_b32", body)) == len(loads) - 8      # (the four priming loads of each pass are not waited for one by one)
    for m in re.finditer(r"v_accvgpr_(?:write_b32|mov_b32) a(\d+)", body):
        assert int(m.group(1)) >= 16, m.group(0)                   # the compiler's own accumulator registers lie behind the ring's
    # inside a pass - from its first ring load to the drain the source places behind it - the only vmcnt waits are the ring's own
    lines = body.split("\n")
    in_ring = False
    n_pass = 0
    for k, ln in enumerate(lines):
        t = ln.strip()
        if t.startswith("global_load_dwordx4 a[0:3]") and not in_ring:
            in_ring = True
            n_pass += 1
        elif in_ring and t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            assert "lgkmcnt" not in t and "expcnt" not in t, "a compiler wait for all loads inside the ring: " + t      # (the drain is a bare asm vmcnt(0))
            in_ring = False
        elif in_ring and t.startswith("s_waitcnt") and "vmcnt" in t:
            assert "vmcnt(3)" in t, t
    assert n_pass >= 2 and not in_ring
    assert len(re.findall(r"ds_or_rtn_b32", body)) >= 16 and "ds_cmpst_rtn_b32" in body


def test_xdrop_cross_lane_traffic_is_explicit():
    """k_xdrop / k_local hand traceback bytes, row windows and run lists from one lane to another through HBM scratch.
    Pinned on the emitted code (ugs_xdrop_dev.h, UGS_XD_SYNC): the readers are agent-scope loads (sc1: never served by a
    line the CU's L1 kept from an earlier job), and the path merge of k_xdrop has no read-modify-write on the arena
    (round 2 zero-filled it with plain stores and merged with atomics behind a fence that waits for no store)."""
    x = _kernel_body(_isa_of("ugs_xdrop.hip"), "k_xdrop")
    assert len(re.findall(r"global_load_ubyte [^\n]* sc1", x)) >= 3         # traceback bytes (three states)
    assert len(re.findall(r"global_load_dwordx2 [^\n]* sc1", x)) >= 3       # row windows
    assert len(re.findall(r"global_load_dword [^\n]* sc1", x)) >= 2         # run lists
    atom = re.findall(r"global_atomic_\w+[^\n]*", x)
    assert all("_x2" in a or " sc0" in a for a in atom), atom                # job counter (returning), arena offset / cell counter (64-bit): no 32-bit adds into the arena
    lo = _kernel_body(_isa_of("ugs_local.hip"), "k_local")
    assert len(re.findall(r"global_load_ubyte [^\n]* sc1", lo)) >= 3


def test_rank2_ring_loads_live_in_accumulator_registers_and_nothing_spills():
    """k_rank2 (ugs_rank2.hip) hides its posting loads from the compiler (asm) and counts their waits by hand.  That is only
    sound while (1) the loads land in accumulator registers the compiler never touches - no spill code, no compiler-made
    v_accvgpr_* - (2) every ring stage issues exactly one load and waits with vmcnt(3) = ring depth - 1, (3) the kernel makes no
    scratch access (a scratch store or load would sit in the same in-order VMEM queue), and (4) the ring is drained (vmcnt(0))
    before the slots are used again."""
    isa = _isa_of("ugs_rank2.hip")
    # <D = 4, CL, P16, HV>: the search kernel over 32-bit postings, its cluster_fast instantiation, and the heavy-unit instantiation (r6)
    for inst in ("k_rank2ILi4ELb0ELb0ELb0EE", "k_rank2ILi4ELb1ELb0ELb0EE", "k_rank2ILi4ELb1ELb0ELb1EE"):
        _check_rank2_ring(isa, inst)
    # <D = 4, CL = false, P16 = true>: the search kernel over 16-bit partition-relative postings (r6) - the same ring at half width:
    # 8-byte loads into a[2k : 2k+1], eight accumulator registers
    _check_rank2_ring(isa, "k_rank2ILi4ELb0ELb1ELb0EE", half=True)


def _kernel_meta(isa, mangled_prefix):
    """the kernel's entry of the code object's metadata (one YAML list item per kernel, starting at its .agpr_count line)"""
    blocks = isa.split("\n  - .agpr_count:")
    hit = [b for b in blocks[1:] if re.search(r"\.name:\s+" + re.escape(mangled_prefix), b)]
    assert len(hit) == 1, mangled_prefix
    return ".agpr_count:" + hit[0]


def _check_rank2_ring(isa, inst, half=False):
    body = _kernel_body(isa, inst)
    meta = _kernel_meta(isa, "_Z7" + inst)
    nreg = 8 if half else 16                                          # accumulator registers of the ring
    plain = "ILi4ELb0E" in inst                                       # (not the cluster_fast instantiation)
    assert re.search(r"\.vgpr_spill_count:\s*0\b", meta), "k_rank2 spills vector registers"
    assert re.search(r"\.agpr_count:\s*(\d+)", meta) and int(re.search(r"\.agpr_count:\s*(\d+)", meta).group(1)) >= nreg, "the ring's accumulator registers"
    if plain:
        assert re.search(r"\.agpr_count:\s*%d\b" % nreg, meta), "the search kernel holds nothing but the ring in accumulator registers"
    assert "scratch_" not in body
    if half:
        loads = re.findall(r"global_load_dwordx2 (a\[\d+:\d+\])", body)
        assert sorted(set(loads)) == ["a[0:1]", "a[2:3]", "a[4:5]", "a[6:7]"] and len(loads) == 8, loads
        assert not re.search(r"global_load_dwordx4 a\[", body)
    else:
        loads = re.findall(r"global_load_dwordx4 (a\[\d+:\d+\])", body)
        assert sorted(set(loads)) == ["a[0:3]", "a[12:15]", "a[4:7]", "a[8:11]"] and len(loads) == 8, loads      # prologue + ring, one slot each
    if half:
        # (the compiler's own 8-byte loads - partition-table pairs, row offsets - have VGPR destinations; none of them may sit inside the ring)
        first, last = body.index("global_load_dwordx2 a["), body.rindex("global_load_dwordx2 a[")
        assert not re.search(r"global_load_dword(x\d)? v", body[first:last]), "a compiler load inside the ring: the hand-counted vmcnt(3) would be off"
    else:
        assert not re.search(r"global_load_dwordx4 v\[", body), "a posting load with a VGPR destination"
    reads = re.findall(r"v_accvgpr_read_b32 v\d+, (a\d+)", body)
    ring = [a for a in reads if int(a[1:]) < nreg]
    assert len(ring) == nreg and sorted(set(ring), key=lambda a: int(a[1:])) == ["a%d" % i for i in range(nreg)], reads
    if plain:
        assert len(reads) == nreg and "v_accvgpr_write" not in body and "v_accvgpr_mov" not in body
    else:
        # the cluster_fast instantiation: the compiler parks a few spilled values in accumulator registers of its own - never in
        # the ring's a0 .. a15 (a load may still be in flight to those after the statement that names them as clobbered)
        for m in re.finditer(r"v_accvgpr_(?:write_b32 (a\d+)|mov_b32 (a\d+), (a\d+))", body):
            for a in m.groups():
                assert a is None or int(a[1:]) >= 16, m.group(0)
    # every block of reads follows its own counted wait inside one asm statement
    assert len(re.findall(r"s_waitcnt vmcnt\(3\)\n\s*v_accvgpr_read_b32", body)) == 4
    assert "flat_atomic" not in body
    if inst.endswith("Lb1ELb0ELb1EE"):
        # the heavy-unit instantiation: pass 1 counts with adds that return nothing, pass 2 exchanges the counter for zero
        assert len(re.findall(r"ds_add_u32 ", body)) >= 16 and len(re.findall(r"ds_and_rtn_b32", body)) >= 16
    else:
        # the atomics of the bitmap are LDS instructions (not flat), with return
        assert len(re.findall(r"ds_or_rtn_b32", body)) >= 16


def test_k_align_search_instantiations_use_no_scratch():
    """k_align's per-wave LDS pointers and scratch pointers derive from the wave index; read from threadIdx without readfirstlane the
    compiler held all of them in VECTOR registers and the search kernels spilled 95 (nt) / 72 (aa) VGPRs to scratch - 3.5 GB of scratch
    writes per C2 launch (r4).  With the index in an SGPR and the wave-uniform LDS values moved to the scalar file they spilled 19 / 1
    (r5: C2 18.0 -> 15.8 ms): loop-invariant values derived from the LANE index (lane masks, lane * stride addresses of the query
    set-up), stored once per kernel and loaded back once per unit - 4.9 KB of scratch reads per unit.  The set-up phases now make their
    lane index on the spot (fresh_lane) and the two search instantiations have no scratch at all."""
    isa = _isa_of("ugs_align.hip")
    for inst in ("k_alignILb0ELb1EE", "k_alignILb0ELb0EE"):            # <PAIR = false, NT = true / false>: the kernels of a plain search
        meta = _kernel_meta(isa, "_Z7" + inst)
        assert re.search(r"\.vgpr_spill_count:\s*0\b", meta) and re.search(r"\.private_segment_fixed_size:\s*0\b", meta), (inst, meta)
        body = _kernel_body(isa, inst)
        assert "scratch_" not in body, inst
        assert re.search(r"v_readfirstlane_b32 s\d+, v\d+", body[:4000]), "the wave index is not moved to an SGPR at the kernel's start"
