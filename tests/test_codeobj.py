"""The register budget the hand-scheduled kernels depend on, read from the code object inside plonky2_amd/libp2hot.so
(tools/codeobj.py: NT_AMDGPU_METADATA).  A compiler bump that turns the clobber sets of gl_mul3.hpp or the limb passes'
budget into spills fails HERE, in the GPU-less tier, not as a silent slowdown on the MI355X."""
import os

import pytest

from tests.conftest import ROOT

SO = os.path.join(ROOT, "plonky2_amd", "libp2hot.so")


@pytest.fixture(scope="module")
def md():
    if not os.path.exists(SO):
        pytest.skip("plonky2_amd/libp2hot.so has not been built (python -c 'import __graft_entry__ as g; g.build()')")
    from tools import codeobj
    return codeobj.kernel_metadata(SO)


def _one(md, *subs):
    hits = [k for n, k in md.items() if all(s in n for s in subs)]
    assert len(hits) == 1, (subs, [k[".name"] for k in hits])
    return hits[0]


def test_leaf_sponge_keeps_its_register_budget(md):
    """hash_leaves: four waves per SIMD (<= 128 VGPRs: 119 since round 6, when the last MDS layer got its single-row form -- the
    branchy tail keeps a copy of the state alive; a fifth wave would need <= 96 and was measured not to pay), no spills, no scratch
    """
    k = _one(md, "hash_leaves_kernel", "ColMajorReader", "18hash")
    assert k[".vgpr_count"] <= 128 and k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, k
    assert k.get(".sgpr_spill_count", 0) <= 96  # round constants parked in VGPR lanes outside the loops (83 v_readlane per permutation)
    lvl = _one(md, "19merkle_level_kernel")
    assert lvl[".vgpr_count"] <= 128 and lvl[".vgpr_spill_count"] == 0 and lvl[".private_segment_fixed_size"] == 0


def test_limb_ntt_passes_fit_four_waves_without_spills(md):
    """every ntt_limbpass instantiation: <= 128 VGPRs (two 512-thread workgroups per CU), nothing spilled to scratch"""
    from tools import codeobj
    names = [n for n in md if "ntt_limbpass_kernel" in n]
    assert len(names) >= 40
    for n in names:
        k = md[n]
        assert k[".vgpr_count"] <= 128 and k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, n
        assert k[".max_flat_workgroup_size"] == 512
        assert codeobj.waves_per_simd(k[".vgpr_count"], 512) >= 4, n
    # the four launches of the headline commit (iNTT strided / contiguous with the 1/n fold, LDE strided -- the form that reads the
    # bit-reversed inverse transform: ..., DUAL = 1, BRIN = true -- / contiguous)
    for sub in ("ILb1ELi8ELi4ELi0ELi0ELi1ELb0E", "ILb1ELi12ELi0ELi0ELi1ELi1ELb0E", "ILb0ELi8ELi4ELi2ELi2ELi1ELb1E", "ILb0ELi12ELi0ELi0ELi0ELi1ELb0E"):
        assert md[_one(md, "ntt_limbpass_kernel", sub)[".name"]][".vgpr_count"] <= 112


def test_word_per_lane_poseidon_stays_spill_free(md):
    for n, k in md.items():
        if "row_kernel" in n or "challenger_kernel" in n:
            assert k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, n
