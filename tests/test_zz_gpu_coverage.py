"""Runs LAST (file order): every ranking kernel the library holds must have been launched by some test of this session
 - an instantiation no test reaches is an unverified code path.  The library records its launches per
process (ugs_debug_rank_instances, include/ugs.h); the tests that reach the rarer ones force them with the debug switches
(UGS_LONGROWS, UGS_WIDE_OFFSETS) next to an oracle or a cross-kernel comparison.  Only a full GPU run is judged."""
import pytest

from usearch12_amd import capi

pytestmark = pytest.mark.gpu

def test_every_compiled_ranking_kernel_was_launched_in_this_session(request):
    ran = request.session.__dict__.get("_ugs_gpu_tests", 0)
    if ran < 150:
        pytest.skip("only %d GPU tests ran in this session: instantiation coverage is a property of the full suite" % ran)
    seen, compiled = capi.rank_instances()
    NAMES = capi.rank_instance_names()                      # the library's own table (ugs_dev.h UGS_RANK_INST_TABLE)
    assert len(NAMES) >= 15 and compiled == sum(1 << i for i in NAMES)
    missing = [NAMES[i] for i in NAMES if not (seen >> i) & 1]
    assert not missing, "ranking kernels no test of this session launched: %s" % missing
