"""CPU: what the compiler recorded about every kernel of the built library (tools/kernel_resources.py reads the gfx950 code objects' metadata — no GPU):
none of OUR kernels spills a vector register or uses scratch memory (a spilled VGPR is an HBM round trip per use; rocprim's radix sort keeps 80 bytes of its
own), every workgroup's static LDS fits the 160 KB of a gfx950 compute unit, and the hot kernels named by DESIGN.md 4 exist
under the register budgets their occupancy is derived from."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def table():
    from contrastboundary_amd import build
    if build.is_stale():
        build.build()
    import kernel_resources
    ks = kernel_resources.kernels()
    assert len(ks) > 500
    return ks, kernel_resources.ours


def test_no_kernel_of_ours_spills_vector_registers_or_uses_scratch(table):
    ks, ours = table
    bad = [k for k in ks if ours(k) and (k.get("private_segment_fixed_size", 0) or k.get("vgpr_spill_count", 0))]
    assert not bad, [(k["file"], k["name"][:80], k.get("private_segment_fixed_size"), k.get("vgpr_spill_count")) for k in bad]
    assert all(k.get("group_segment_fixed_size", 0) <= 160 * 1024 for k in ks)
    assert all(k["vgpr_count"] + k.get("agpr_count", 0) <= 512 for k in ks)


@pytest.mark.parametrize("file,needle,max_vgpr", [
    ("pointops_gather", "query_group_lds_pipe", 64),                  # the north-star gather: 8 waves per SIMD
    ("knn_grid", "knn_grid_wave_kernel", 128),                        # the K = 36 search: 4 waves per SIMD
    ("knn_grid", "knn_grid_group_kernel", 64),
    ("fps_bucket", "fps_bucket_kernel", 128),                         # one resident workgroup of 16 waves: 4 per SIMD, 512 / 4 registers each
    ("local_aggregation", "kpconv_fwd_c64_kernel", 128),
])
def test_hot_kernels_fit_their_register_budget(table, file, needle, max_vgpr):
    ks, _ = table
    mine = [k for k in ks if k["file"] == file and needle in k["name"]]
    assert mine, (file, needle)
    for k in mine:
        assert k["vgpr_count"] + k.get("agpr_count", 0) <= max_vgpr, (k["name"][:100], k["vgpr_count"])
