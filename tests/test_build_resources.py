"""Register / scratch figures of the kernels in the library the tests load (read from the code objects' metadata, no GPU):
the kernels of the default path keep everything in registers.  A spill inside a hand-pipelined loop is not only slow -- the
reload's compiler-inserted s_waitcnt vmcnt(0) drains the LDS-DMA prefetch ring the loop counts on."""
import os
import re

import pytest

from phamers_amd import _codeobj, _lib

# kernels allowed to use scratch, and why
ALLOWED = {}   # (round 5: none -- the last one, NumPy's recursive pairwise row sum, keeps its frames in LDS)


@pytest.fixture(scope="module")
def resources():
    path = os.path.join(os.path.dirname(_lib.__file__), "libphamers_hip.so")
    res = _codeobj.kernel_resources(path)
    assert len(res) > 100, "no gfx950 kernels found in %s" % path
    return res


def test_default_path_kernels_use_no_scratch(resources):
    offenders = {}
    for name, r in resources.items():
        if r["scratch_bytes"] == 0:
            continue
        if any(re.search(pat, name) for pat in ALLOWED):
            continue
        offenders[name] = r
    assert not offenders, offenders


def test_the_hot_kernels_are_in_the_library_with_the_expected_shapes(resources):
    def one(pattern):
        hits = [n for n in resources if re.search(pattern, n)]
        assert hits, pattern
        return [resources[n] for n in hits]
    # the sweeps hold two waves per SIMD: at most 256 registers each, none spilled
    for pat in (r"phk_knn_f16h_kernelILi2ELi4E", r"phk_knn_i8_general_kernelILi2ELi6ELi[012]E", r"phk_knn_i8_general_kernelILi3ELi4E",
                r"phk_knn_f16_kernelILi[01]E"):
        for r in one(pat):
            assert r["vgpr_count"] <= 256 and r["vgpr_spill_count"] == 0 and r["scratch_bytes"] == 0, (pat, r)
    for pat in (r"phk_count_pairs_kernelILi1024E", r"phk_count_direct_kernelILi5E", r"phk_decide_h_kernel", r"phk_decide_gen_kernel",
                r"phk_rerank16_kernel"):
        for r in one(pat):
            assert r["scratch_bytes"] == 0, (pat, r)
