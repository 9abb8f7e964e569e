"""uncertainty-nerf-gs_amd/isa_check.py: the build-time scan that refuses a library in which an MFMA reads a VGPR fewer than
two wait states behind the VALU instruction that writes it (hand-placed instructions inside inline assembly are invisible to
the compiler's hazard recogniser: round 4 found one such MFMA per split in every build)."""
import pytest

from uncertainty_nerf_gs_amd import isa_check

HEAD = "0000000000001000 <_Z6kernelv>:\n"
MFMA = "\tv_mfma_f32_32x32x16_f16 v[32:47], v[98:101], v[48:51], v[32:47]      // 000000001010: D3D40020\n"


def _lines(*ins):
    return HEAD + "".join(f"\t{i}      // 000000001000: 00000000\n" for i in ins)


def test_a_valu_write_directly_in_front_of_the_mfma_that_reads_it_is_a_hazard():
    for producer, what in (("v_fma_mixhi_f16 v51, v60, -1.0, v61 op_sel:[1,0,0] op_sel_hi:[1,0,0]", "B"),
                           ("v_mov_b32_e32 v99, v3", "A"), ("v_pk_mul_f32 v[34:35], v[2:3], v[4:5]", "C")):
        h, st = isa_check.scan_listing(_lines(producer) + MFMA)
        assert len(h) == 1 and "0 wait state" in h[0], (what, h)
        h, _ = isa_check.scan_listing(_lines(producer, "s_nop 0") + MFMA)
        assert len(h) == 1 and "1 wait state" in h[0], (what, h)
        for gap in (("s_nop 1",), ("v_and_b32_e32 v7, v8, v9", "v_and_b32_e32 v10, v8, v9"), ("s_nop 0", "ds_read_b128 v[200:203], v5")):
            h, st = isa_check.scan_listing(_lines(producer, *gap) + MFMA)
            assert h == [] and st["_Z6kernelv"]["mfma"] == 1, (what, gap, h)
    # writes to registers the MFMA does not read, scalar destinations, and windows cut by control flow are no hazards
    for ins in ("v_mov_b32_e32 v7, v3", "v_cmp_lt_f32_e32 vcc, v48, v49", "v_readfirstlane_b32 s4, v48"):
        assert isa_check.scan_listing(_lines(ins) + MFMA)[0] == []
    cut = _lines("v_mov_b32_e32 v48, v3", "s_cbranch_scc1 0x10") + MFMA
    assert isa_check.scan_listing(cut)[0] == []
    st = isa_check.scan_listing(_lines("v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[2:3]", "s_nop 4") + MFMA)[1]
    assert st["_Z6kernelv"] == {"mfma": 1, "pk_f32": 1}


def test_the_built_library_passes_and_its_matrix_kernels_are_found(lib):
    import os
    if not os.path.exists(os.path.join(isa_check.LLVM_BIN, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    stats = isa_check.check_library(lib.build_library())       # raises IsaHazard on a finding
    mk = {k: v for k, v in stats.items() if v["mfma"]}
    assert len(mk) >= 30 and any("field_kernel_mfma16" in k for k in mk) and any("laplace" in k for k in mk)
    # the rasteriser, the sorts, the proposal kernels: no matrix instruction
    assert all(v["mfma"] == 0 for k, v in stats.items() if "raster" in k or "prop_patch" in k)
