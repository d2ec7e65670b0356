"""Register / scratch budget of the hot kernels, read from the code objects inside the BUILT library (no GPU needed).

The bucket passes and the NTT kernels sit right at resource cliffs the compiler does not warn about: one register past 256 drops
a kernel from two waves per SIMD to one (round 2: 10.1 -> 12.5 ms per G1 bucket pass), and an unrolled block past LLVM's pragma-
unroll cap leaves the butterfly registers in scratch memory (NTT 6.3 -> 11.4 ms).  Both regressions were silent -- every parity
test stayed green -- so the budgets are asserted here."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def kernels():
    import kernel_occupancy

    import groth16_amd

    ks = kernel_occupancy.kernels(groth16_amd.lib().path)
    assert len(ks) > 40, "could not read the code objects of the built library"
    return ks


def pick(kernels, *subs):
    hit = {n: k for n, k in kernels.items() if all(s in n for s in subs)}
    assert hit, f"no kernel matches {subs}"
    return hit


@pytest.mark.parametrize("curve,limbs", [("Bls12_381FqP", 13), ("Bn254FqP", 9)])
def test_bucket_pass_keeps_two_waves_per_simd_and_nothing_in_scratch(kernels, curve, limbs):
    """the production walk (sorted entry words), G1 and the lane-pair G2 kernel.  Round 4: the accumulator's four coordinates live in
    LDS (AccParked: 4 * NL words per lane, 64 lanes per workgroup) -- the G2 kernel, which spilled 392 B per lane into HBM-backed
    scratch while using no LDS at all, and the G1 kernel must now have NO scratch, and keep two waves per SIMD (<= 256 registers)."""
    for field in ("Fp30<", "Fp2p30<"):
        hit = pick(kernels, "bucket_accumulate30_kernel", field + curve, "false")
        for name, k in hit.items():
            assert k["waves_per_simd"] >= 2, (name, k)
            assert k["scratch"] == 0, (name, k)
            assert k["lds"] == 4 * limbs * 64 * 4, (name, k)   # x, y, zz, zzz: NL words each for 64 lanes
    # two waves per SIMD of eight single-wave workgroups per CU: their accumulator blocks must fit the CU's 160 KB of LDS
    assert 8 * 4 * limbs * 64 * 4 <= 160 * 1024


def test_g1_bucket_pass_runs_three_waves_per_simd(kernels):
    """Round 6: product-scanning field products (generated assembly, fp30.hpp / gen_fips_asm.py) keep no 2 NL-column array, and the G1
    bucket kernel went from 190 to ~140 registers: THREE waves per SIMD (<= 168 registers; twelve 13 KB accumulator blocks are exactly
    the CU's LDS).  Held to two waves the pass is 8 % slower (profiles/r06_ab_fips_pins.txt, asm_wg8), so the budget is asserted."""
    for name, k in pick(kernels, "bucket_accumulate30_kernel", "Fp30<Bls12_381FqP").items():
        assert k["waves_per_simd"] >= 3 and k["scratch"] == 0, (name, k)
        assert 12 * k["lds"] <= 160 * 1024, (name, k)
    for name, k in pick(kernels, "bucket_accumulate30_kernel", "Fp30<Bn254FqP").items():
        assert k["waves_per_simd"] >= 4 and k["scratch"] == 0, (name, k)


def test_ntt_kernels_keep_their_butterflies_in_registers(kernels):
    for name, k in pick(kernels, "ntt30_").items():
        assert k["scratch"] == 0, (name, k)
        assert k["waves_per_simd"] >= 4, (name, k)       # four workgroups of 256 lanes per CU (38 KB of LDS each): <= 128 registers


def test_streaming_kernels_have_no_scratch(kernels):
    for sub in ("spmv3_kernel", "quotient_kernel", "bitrev_scale_kernel", "class_count_kernel", "class_partition_kernel",
                "bucket_count_merged_kernel", "bucket_scatter_merged_kernel", "bucket_wg_scan_kernel", "digits_kernel", "dwm_column_kernel"):
        for name, k in pick(kernels, sub).items():
            assert k["scratch"] == 0, (name, k)


def test_window_table_builders_have_no_per_lane_arrays(kernels):
    """round 2's builders kept W XYZZ points in per-lane arrays: 5-17 KB of scratch per lane, 1.4 s for one G2 table.  The
    round-3 builders (window_tables.hpp) park their rows in an explicit HBM buffer; what is left in scratch is the standard-form
    conversion of the input point (<= 256 B), and they keep two waves per SIMD."""
    for name, k in pick(kernels, "build_window_tables_kernel").items():
        assert k["scratch"] <= 256, (name, k)
        assert k["waves_per_simd"] >= 2, (name, k)


def test_reductions_keep_two_waves_per_simd_and_nothing_in_scratch(kernels):
    """the three reduction kernels, G1 and lane-pair G2.  Round 3: the G2 ones needed 376 + 120 registers (ONE wave per SIMD, and no
    bucket-pass wave beside them) and bucket_reduce_kernel<Fp30> spilled 144 B.  Round 4: they add with acc_add_streamed (both operands
    behind stores, coordinates fetched where they are consumed): every one of the six must fit two waves per SIMD with no scratch."""
    for sub in ("bucket_reduce_kernel<", "window_reduce_kernel<", "heavy_reduce_kernel<", "bucket_combine_kernel<"):
        for field in ("Fp30<", "Fp2p30<"):
            for name, k in pick(kernels, sub, field).items():
                assert k["waves_per_simd"] >= 2, (name, k)
                assert k["scratch"] == 0, (name, k)


def test_counting_sort_fits_on_a_compute_unit_beside_the_g2_bucket_pass(kernels):
    """Round 6: with the G1 kernel at three waves per SIMD its twelve workgroups take all of a CU's LDS, but h's sort no longer needs
    to live beside it: the lane-pair kernel (185 registers since the assembly products, 239 before) leaves room for a sort wave
    (2 x 192 + 56 <= 512 registers, 8 x 13 KB + 32 KB of LDS), so the sort runs underneath the G2 pass -- the pass that starts when
    the witness map ends -- and is done long before the G1 launch (profiles/r06_final_timeline_single_k22.txt)."""
    def gran(v):
        return (v + 7) // 8 * 8

    g2 = pick(kernels, "bucket_accumulate30_kernel", "Fp2p30<Bls12_381FqP", "false")
    g2_regs = max(gran(k["vgpr"]) for k in g2.values())
    g2_lds = max(k["lds"] for k in g2.values())
    for sub in ("class_count_kernel", "class_partition_kernel", "bucket_count_merged_kernel", "bucket_scatter_merged_kernel",
                "bucket_wg_scan_kernel", "bucket_slots_kernel", "scan_block_sums_kernel", "scan_block_offsets_kernel", "scan_write_kernel"):
        for name, k in pick(kernels, sub).items():
            assert 2 * g2_regs + gran(k["vgpr"]) <= 512, (name, k, g2_regs)
            lds = k["lds"] + ((4 << 13) if "merged" in name else 0)
            assert 8 * g2_lds + lds <= 160 * 1024, (name, k)


def test_counting_sort_fits_on_a_compute_unit_beside_the_g1_bucket_pass(kernels):
    """Round 5: h's sort cannot start before the witness map ends -- which is when the passes begin -- so its kernels must be able to
    LIVE BESIDE a pass instead of waiting for a kernel boundary (with 128 KB histograms each of its two big kernels sat out a whole
    23 ms pass and the h pass then waited 2.3 ms; same box 69.2 -> 65.6 ms per proof).  A compute unit holds eight G1-pass workgroups
    (one wave each, two per SIMD) = 8 x 13 KB of LDS and 2 x (its registers, in granules of 8) of a SIMD's 512; a sort workgroup is 256
    lanes = one wave per SIMD and a 32 KB histogram (SORT_CLASS_LOG = 13: dynamic LDS, 4 << 13 bytes).  Both budgets are asserted."""
    def gran(v):
        return (v + 7) // 8 * 8

    g1 = pick(kernels, "bucket_accumulate30_kernel", "Fp30<Bls12_381FqP", "false")
    g1_regs = max(gran(k["vgpr"]) for k in g1.values())
    g1_lds = max(k["lds"] for k in g1.values())
    for sub in ("class_count_kernel", "class_partition_kernel", "bucket_count_merged_kernel", "bucket_scatter_merged_kernel",
                "bucket_wg_scan_kernel", "bucket_slots_kernel", "scan_block_sums_kernel", "scan_block_offsets_kernel", "scan_write_kernel"):
        for name, k in pick(kernels, sub).items():
            if sub in ("class_count_kernel", "class_partition_kernel", "bucket_count_merged_kernel", "bucket_scatter_merged_kernel"):
                assert k["max_flat_wg"] <= 256, (name, k)                                # one wave per SIMD (the others are launched with 256 lanes)
            assert 2 * g1_regs + gran(k["vgpr"]) <= 512, (name, k, g1_regs)            # registers of a SIMD: two pass waves + one sort wave
            lds = k["lds"] + ((4 << 13) if "merged" in name else 0)
            assert 8 * g1_lds + lds <= 160 * 1024, (name, k)                           # LDS of the compute unit
