"""A C program proves (tests/abi_prove.c: strict C99, no Python in the process): g16_ctx_create -> g16_pk_load -> g16_circuit_load ->
g16_prove -> g16_serialize_points through include/g16_mi355x.h, on golden cases of tests/golden/<curve>.json written out as flat
binary files -- expected proof = the golden one (affine coordinates), expected bytes = oracle/pymodel.py's serialiser
(Proof: a || b || c, /root/reference/src/data_structures.rs:8-16) -- and on an oracle-generated 1022-constraint instance.
The CPU tier compiles and links the same program and checks that without a GPU it stops at the loud G16_ERR_NO_DEVICE."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

import pymodel as pm
from helpers import circuit_from_pymodel, g1_to_arr, g2_to_arr, ints_to_mont, oracle, pk_from_pymodel  # noqa: F401
from test_golden import _p1, _p2, load_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CP = {"bls12_381": pm.BLS12_381, "bn254": pm.BN254}
CURVE_ID = {"bls12_381": 0, "bn254": 1}


def write_case(path, curve, ck, fpk, r, s, want_proof, want_bytes):
    """the flat little-endian layout tests/abi_prove.c documents"""
    cp = CP[curve]
    L = cp.fq_limbs64
    u = lambda *v: np.array(v, dtype=np.uint64)  # noqa: E731
    flat = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)  # noqa: E731
    nnz = [int(m.row_ptr[-1]) for m in ck.abc]
    parts = [np.frombuffer(b"G16CASE1", dtype=np.uint64), u(CURVE_ID[curve], L, ck.num_inputs, ck.num_constraints, ck.num_vars, len(fpk.h_query)),
             u(*nnz), u(len(want_bytes))]
    parts += [flat(fpk.alpha_g1), flat(fpk.beta_g1), flat(fpk.delta_g1), flat(fpk.beta_g2), flat(fpk.delta_g2), flat(fpk.a_query),
              flat(fpk.b_g1_query), flat(fpk.b_g2_query), flat(fpk.h_query), flat(fpk.l_query)]
    assert len(fpk.a_query) == ck.num_vars and len(fpk.l_query) == ck.num_vars - ck.num_inputs
    for m in ck.abc:
        parts += [flat(m.row_ptr), m.col.astype(np.uint64), flat(m.val)]
    pad = (-len(want_bytes)) % 8
    parts += [flat(ck.z), flat(r), flat(s), flat(want_proof), np.frombuffer(want_bytes + b"\0" * pad, dtype=np.uint64)]
    np.concatenate(parts).tofile(path)


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("abi_prove") / "abi_prove")
    libdir = os.path.join(ROOT, "groth16_amd")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "abi_prove.c"), "-o", out, "-L", libdir, "-l:libg16_mi355x.so", f"-Wl,-rpath,{libdir}"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return out


def golden_case(curve, want_name):
    cp = CP[curve]
    for name, cp_, cs, z, r, s, pk, ex in load_cases(curve):
        if name == want_name:
            ck, fpk = circuit_from_pymodel(cp, cs, z), pk_from_pymodel(cp, pk)
            a, b, c = _p1(ex["proof_a"]), _p2(ex["proof_b"]), _p1(ex["proof_c"])
            want = np.concatenate([g1_to_arr([a], cp)[0], g2_to_arr([b], cp)[0], g1_to_arr([c], cp)[0]])
            return ck, fpk, ints_to_mont([r], cp.r, 4)[0], ints_to_mont([s], cp.r, 4)[0], want, pm.proof_bytes(cp, pm.Proof(a, b, c))
    raise KeyError(want_name)


@pytest.mark.skipif(torch.cuda.is_available(), reason="the GPU tier runs the real thing")
def test_c_prover_builds_and_fails_loudly_without_a_gpu(exe, tmp_path):
    ck, fpk, r, s, want, wb = golden_case("bn254", "syn3_rs")
    path = str(tmp_path / "case.bin")
    write_case(path, "bn254", ck, fpk, r, s, want, wb)
    run = subprocess.run([exe, path], capture_output=True, text=True, timeout=120)
    d = json.loads(run.stdout.strip().splitlines()[-1])
    assert run.returncode == 1 and d["stage"] == "ctx_create" and d["rc"] == 6, run.stdout   # G16_ERR_NO_DEVICE: no CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
@pytest.mark.parametrize("name", ["mimc7_rs", "syn5_dense_r0", "syn3_s0"])
def test_c_program_proves_golden_case(exe, tmp_path, curve, name):
    ck, fpk, r, s, want, wb = golden_case(curve, name)
    path = str(tmp_path / "case.bin")
    write_case(path, curve, ck, fpk, r, s, want, wb)
    run = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    d = json.loads(run.stdout.strip().splitlines()[-1])
    assert d["stage"] == "done" and d["proof_matches"] == 1 and d["second_proof_identical"] == 1 and d["bytes_match"] == 1, d
    assert d["proof_bytes"] == (192 if curve == "bls12_381" else 128) and d["table_fallback"] == 0 and d["window_bits"] >= 9


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_c_program_proves_oracle_instance(exe, tmp_path, orc, curve):
    """1022 constraints, valid CRS from the C++ oracle's generator, expected proof from the oracle's prover"""
    cp = CP[curve]
    ck = orc.syn_circuit(curve, 10, 9)
    fpk, _ = orc.setup(ck, 4)
    r, s = orc.rand_fr(curve, 21, 1)[0], orc.rand_fr(curve, 22, 1)[0]
    want, _ = orc.prove(fpk, ck, r, s)
    from helpers import arr_to_g1, arr_to_g2

    L = cp.fq_limbs64
    pr = pm.Proof(arr_to_g1(want[: 2 * L], cp)[0], arr_to_g2(want[2 * L: 6 * L], cp)[0], arr_to_g1(want[6 * L:], cp)[0])
    path = str(tmp_path / "case.bin")
    write_case(path, curve, ck, fpk, r, s, want, pm.proof_bytes(cp, pr))
    run = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    d = json.loads(run.stdout.strip().splitlines()[-1])
    assert d["proof_matches"] == 1 and d["bytes_match"] == 1 and d["domain_size"] == 1024, d
