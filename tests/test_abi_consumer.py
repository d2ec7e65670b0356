"""CPU tier: a strict-C99 program (tests/abi_consumer.c) is compiled against include/g16_mi355x.h, linked to the product library and
run; its view of every struct (size, field offsets, field sizes) must equal the hand-written ctypes mirrors in
groth16_amd/binding.py -- the check a Rust `extern "C"` author would want before trusting the header."""
import ctypes as C
import json
import os
import subprocess

import pytest

from groth16_amd import binding as b

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MIRRORS = {
    "g16_query": b.QueryC, "g16_pk_view": b.PkViewC, "g16_csr_view": b.CsrViewC, "g16_proof": b.ProofC, "g16_partial": b.PartialC,
    "g16_timings": b.TimingsC, "g16_diag": b.DiagC, "g16_toxic_waste": b.ToxicWasteC, "g16_params_view": b.ParamsViewC,
    "g16_pk_info": b.PkInfoC,
}


@pytest.fixture(scope="module")
def consumer(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("abi") / "abi_consumer")
    libdir = os.path.join(ROOT, "groth16_amd")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "abi_consumer.c"), "-o", exe, "-L", libdir, "-l:libg16_mi355x.so", f"-Wl,-rpath,{libdir}"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    return json.loads(run.stdout)


def test_header_compiles_as_c99_and_the_c_calls_work(consumer):
    assert consumer["failures"] == 0
    assert consumer["selftest_0"] == 0 and consumer["selftest_1"] == 0
    assert consumer["field_op_0"] == 1 and consumer["field_op_1"] == 1
    assert consumer["ctx_create"] in (0, 6)   # a context, or G16_ERR_NO_DEVICE on a box without a GPU
    assert consumer["version"]
    assert consumer["abi_version"] == 2


@pytest.mark.parametrize("name", sorted(MIRRORS))
def test_ctypes_mirror_equals_the_c_layout(consumer, name):
    lay = dict(consumer["layout"][name])
    mirror = MIRRORS[name]
    assert lay.pop("sizeof")[1] == C.sizeof(mirror), name
    fields = {n: (getattr(mirror, n).offset, getattr(mirror, n).size) for n, _ in mirror._fields_}
    assert {k: tuple(v) for k, v in lay.items()} == fields, name
