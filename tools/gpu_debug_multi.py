"""debug driver: multi-device context, distributed path, with a watchdog"""
import faulthandler, os, sys
faulthandler.dump_traceback_later(90, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import groth16_amd as g
from helpers import oracle
orc = oracle()
n_dev = int(sys.argv[1]) if len(sys.argv) > 1 else 2
curve = "bls12_381"
ck = orc.syn_circuit(curve, 11, 6)
pk, _ = orc.setup(ck, 4)
gm = g.ConstraintMatrices(ck.num_inputs, ck.num_vars - ck.num_inputs, ck.num_constraints, *[(m.row_ptr, m.col, m.val) for m in ck.abc])
gp = g.ProvingKey(pk.curve, pk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.beta_g2, pk.delta_g2, pk.a_query, pk.b_g1_query, pk.b_g2_query, pk.h_query, pk.l_query)
print("creating ctx", flush=True)
with g.Groth16(curve, [0] * n_dev) as prover:
    print("ctx ok", flush=True)
    dpk = prover._pk(gp, ck.num_inputs); print("pk ok", flush=True)
    dck = prover._ck(gm); print("circuit ok", flush=True)
    r, s = orc.rand_fr(curve, 91, 1)[0], orc.rand_fr(curve, 92, 1)[0]
    proof = prover.create_proof_with_reduction_and_matrices(gp, r, s, gm, ck.num_inputs, ck.num_constraints, ck.z)
    print("proof ok", (proof.flat() == orc.prove(pk, ck, r, s)[0]).all(), flush=True)
