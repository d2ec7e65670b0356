"""CPU tier: the generated sources committed under groth16_amd/csrc/ are what their generators write today (a drift guard: the
assembly blocks of fips_asm_gen.hpp are the field arithmetic every MSM and NTT kernel runs, and the Makefile only regenerates them when
the generator is newer than the header)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "groth16_amd", "csrc")


def test_fips_asm_header_is_current(tmp_path):
    out = tmp_path / "fips_asm_gen.hpp"
    subprocess.run([sys.executable, os.path.join(CSRC, "gen_fips_asm.py"), str(out)], check=True, cwd=CSRC, capture_output=True)
    assert out.read_text() == open(os.path.join(CSRC, "fips_asm_gen.hpp")).read()


def test_fips_asm_block_shape():
    """every block is a straight multiply-add chain: 2 NL^2 (+ 2 per extra accumulator) v_mad_u64_u32 and 5 NL - 1 other vector
    instructions for a plain product -- the counts the header's trailer records and DESIGN.md 4.1 quotes"""
    txt = open(os.path.join(CSRC, "fips_asm_gen.hpp")).read()
    rows = {}
    for line in txt.splitlines():
        if line.startswith("//   ") and "[" in line:
            name, form, total, mads = line[5:].replace("[", " ").replace("]", " ").split()
            rows[(name, form)] = (int(total), int(mads))
    assert rows[("Bls12_381FqP", "mul")] == (411, 348)      # 13 limbs: 338 + 2 * 5 multiply-adds, 13 * 3 + 12 * 2 others
    assert rows[("Bn254FqP", "mul")] == (205, 162)          # 9 limbs: no column needs a second accumulator
    for (name, form), (total, mads) in rows.items():
        nl = 13 if name == "Bls12_381FqP" else 9
        if form == "mul":
            assert total - mads == 5 * nl - 2, (name, form)
        assert mads >= 2 * nl * nl - (nl * (nl - 1) // 2 if form.startswith("sqr") else 0), (name, form)
