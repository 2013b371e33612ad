"""Runs LAST (file order): every ranking kernel the library holds must have been launched by some test of this session
 - an instantiation no test reaches is an unverified code path.  The library records its launches per
process (ugs_debug_rank_instances, include/ugs.h); the tests that reach the rarer ones force them with the debug switches
(UGS_LONGROWS, UGS_WIDE_OFFSETS) next to an oracle or a cross-kernel comparison.  Only a full GPU run is judged."""
import pytest

from usearch12_amd import capi

pytestmark = pytest.mark.gpu

NAMES = {0: "Big 4-bit (HOT)", 1: "Big 4-bit long rows", 2: "Big 8/16-bit flattened (sparse)", 3: "Big 8/16-bit dense", 4: "Big 8/16-bit dense, long rows",
         5: "small 4-bit", 6: "small 4-bit long rows", 7: "small 8/16-bit flattened", 8: "small 8/16-bit dense", 9: "small 8/16-bit dense, long rows",
         12: "HOT, 64-bit offsets", 13: "long rows, 64-bit offsets", 14: "k_rank2 (bitmap)", 15: "k_rank2g (bitmap, sparse index)", 16: "k_rank2, cluster_fast instantiation"}


def test_every_compiled_ranking_kernel_was_launched_in_this_session(request):
    ran = request.session.__dict__.get("_ugs_gpu_tests", 0)
    if ran < 150:
        pytest.skip("only %d GPU tests ran in this session: instantiation coverage is a property of the full suite" % ran)
    seen, compiled = capi.rank_instances()
    assert compiled == sum(1 << i for i in NAMES)
    missing = [NAMES[i] for i in NAMES if not (seen >> i) & 1]
    assert not missing, "ranking kernels no test of this session launched: %s" % missing
