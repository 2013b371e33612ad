"""Host-heap guard for the GPU test session (test infrastructure).

Round 2's one red GPU test (test_xdrop_matches_oracle[False-200.0-5]) turned out not to be a wrong kernel answer: the
differing path character was 'M' (0x4D) -> 'L' (0x4C) at an 8-byte aligned offset of a malloc'ed Python string that
abi.path_text() had just built - a value no run list can produce ("MDI"[op] has no 'L'): a 64-bit word of host heap memory
was decremented after the string was created.  Such an event silently changes whatever owns
the memory.  This guard makes it visible and attributable: canaries (malloc'ed blocks of every small-bin size filled with
0x4D) are planted after every test and checked after the following tests; a changed canary fails the test during which it
changed with the block size, offset and new bytes, instead of surfacing later as a bogus parity difference."""
import ctypes

_LIBC = ctypes.CDLL(None)
_LIBC.malloc.restype = ctypes.c_void_p
_LIBC.malloc.argtypes = [ctypes.c_size_t]
_LIBC.free.argtypes = [ctypes.c_void_p]
_LIBC.memset.restype = ctypes.c_void_p
_LIBC.memset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
# every malloc bin up to 2 KiB ten times over (a thread's tcache holds 7 chunks per bin), then coarser steps
SIZES = [n for n in range(24, 2048, 16) for _ in range(10)] + [n for n in range(2048, 16384, 128) for _ in range(3)]
GENERATIONS = 4          # canary sets kept alive (each set is checked after each of the next GENERATIONS tests)


class HeapGuard:
    def __init__(self):
        self.sets = []

    def plant(self):
        cans = []
        for n in SIZES:
            p = _LIBC.malloc(n)
            _LIBC.memset(p, 0x4D, n)
            cans.append((p, n))
        self.sets.append(cans)
        while len(self.sets) > GENERATIONS:
            for p, _ in self.sets.pop(0):
                _LIBC.free(p)

    def check(self):
        """-> list of (size, offset, byte) of canary bytes that changed (and restores them)"""
        bad = []
        for cans in self.sets:
            for p, n in cans:
                c = ctypes.string_at(p, n)
                if c.count(b"M") != n:
                    bad += [(n, k, c[k]) for k in range(n) if c[k] != 0x4D]
                    _LIBC.memset(p, 0x4D, n)
        return bad
