"""N > 1 path on CPU: two processes (gloo), each owning one contiguous shard of every MSM base array (mode "base") or one residue
class of the BUCKETS over all bases (mode "bucket", g16_pk_load_bucket_shard's cut), exchange the
fixed-size partial records with ONE all-gather (the same ShardedProver.exchange that carries RCCL traffic on the GPUs)
and each finishes the proof with g16_finalize_host.  The per-shard MSM sums come from the product's CPU model of the
bucket method (g16_host_msm_model) because there is no GPU here; shard ranges, record layout, collective, N-way EC fold
and the prover.rs:76-131 glue are the production code.  Result must equal the single-process oracle proof bit for bit."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _xyzz_from_affine(aff: np.ndarray, one: np.ndarray, g2: bool) -> np.ndarray:
    """affine (x|y) -> XYZZ record (x, y, zz = 1, zzz = 1); identity -> all zero"""
    L = len(one)
    k = 2 if g2 else 1
    out = np.zeros(4 * k * L, dtype=np.uint64)
    if aff.any():
        out[: 2 * k * L] = aff
        out[2 * k * L: 2 * k * L + L] = one
        out[3 * k * L: 3 * k * L + L] = one
    return out


def _worker(rank, world, port, curve, q, mode="base"):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist

    import groth16_amd as g
    from groth16_amd.binding import CURVE_ID, PartialC, ptr64
    from helpers import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = oracle()
        orc.set_threads(2 if world <= 2 else 1)
        lb = g.lib()
        ck = orc.syn_circuit(curve, 6, 3)
        pk, _ = orc.setup(ck, 8)  # deterministic: both ranks build the same valid CRS
        r, s = orc.rand_fr(curve, 31, 1)[0], orc.rand_fr(curve, 32, 1)[0]
        h = orc.witness_map(ck)  # the witness map is replicated on every rank
        z = ck.z
        nin, m, w = ck.num_inputs, ck.num_vars - 1, ck.num_vars - ck.num_inputs
        rg = g.shard_ranges(m, w, len(pk.h_query), nin, rank, world)
        one_fq = orc.field_op(curve, 1, 5, np.array([1] + [0] * (pk.alpha_g1.shape[1] // 2 - 1), dtype=np.uint64))  # 1 in Montgomery form

        def msm(bases, scalars, g2):
            out = np.zeros(bases.shape[1], dtype=np.uint64)
            n = min(len(bases), len(scalars))
            if n:
                b, sc = np.ascontiguousarray(bases[:n]), np.ascontiguousarray(scalars[:n])
                if mode == "bucket":
                    # bucket-space shard: every rank sees ALL bases and scalars and owns the buckets b mod world == rank (the
                    # product's own plan / filter / fold code, g16_host_msm_model_shard).  Unlike base-range shards, the ranks MUST agree
                    # on the window size: the residue classes partition the buckets of ONE digit decomposition
                    assert lb.c.g16_host_msm_model_shard(CURVE_ID[curve], int(g2), ptr64(b), ptr64(sc), n, -11, rank, world, ptr64(out)) == 0
                else:
                    # rank 0 models the per-window bucket scheme, rank 1 the merged-window scheme of a key held as window tables:
                    # partial sums are group elements, so ranks need not agree on how they computed them
                    c_win = 0 if rank == 0 else -10
                    assert lb.c.g16_host_msm_model(CURVE_ID[curve], int(g2), ptr64(b), ptr64(sc), n, c_win, ptr64(out)) == 0
            return _xyzz_from_affine(out, one_fq, g2)

        if mode == "bucket":
            rg = g.shard_ranges(m, w, len(pk.h_query), nin, 0, 1)   # the bases are not cut
        (a_lo, a_hi), (l_lo, l_hi), (h_lo, h_hi) = rg["a"], rg["l"], rg["h"]
        part = PartialC()
        for name, val in (("h", msm(pk.h_query[h_lo:h_hi], h[h_lo:h_hi], False)),
                          ("l", msm(pk.l_query[l_lo:l_hi], z[nin + l_lo: nin + l_hi], False)),
                          ("a", msm(pk.a_query[1 + a_lo: 1 + a_hi], z[1 + a_lo: 1 + a_hi], False)),
                          ("b_g1", msm(pk.b_g1_query[1 + a_lo: 1 + a_hi], z[1 + a_lo: 1 + a_hi], False)),
                          ("b_g2", msm(pk.b_g2_query[1 + a_lo: 1 + a_hi], z[1 + a_lo: 1 + a_hi], True))):
            arr = getattr(part, name)
            for i, v in enumerate(val):
                arr[i] = int(v)
        parts = g.ShardedProver.exchange(bytes(part), dist)  # one all-gather of C.sizeof(PartialC) bytes per rank
        assert len(parts) == world and parts[rank] == bytes(part)
        gpk = g.ProvingKey(curve, pk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.beta_g2, pk.delta_g2, pk.a_query, pk.b_g1_query, pk.b_g2_query,
                           pk.h_query, pk.l_query)
        proof = g.finalize_host(curve, gpk, parts, r, s)
        want, _ = orc.prove(pk, ck, r, s)
        ok = bool((proof.flat() == want).all())
        # the glue is regrouped by linearity (the (r, s)-only half first, then s * sum_a and r * sum_b1): r = 0 (prover.rs:98-108: the
        # reference skips B in G1; the records above still carry that MSM, which r = 0 must cancel), s = 0 and r = s = 0
        z4 = np.zeros(4, dtype=np.uint64)
        for rr, ss in ((z4, s), (r, z4), (z4, z4)):
            want0, _ = orc.prove(pk, ck, rr, ss)
            ok = ok and bool((g.finalize_host(curve, gpk, parts, rr, ss).flat() == want0).all())
        q.put((rank, ok, rg))
    finally:
        dist.destroy_process_group()


def _run_world(world, curve, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, curve, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    rg = {r[0]: r[2] for r in res}
    for key in ("a", "l", "h"):
        for k in range(1, world):
            if mode == "bucket":   # every rank holds every base
                assert rg[0][key] == rg[k][key] and rg[0][key][0] == 0
            else:                  # the shards tile every base array exactly once
                assert rg[0][key][0] == 0 and rg[k - 1][key][1] == rg[k][key][0]


@pytest.mark.parametrize("mode", ["base", "bucket"])
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_two_rank_sharded_proof_gloo(curve, mode):
    _run_world(2, curve, mode)


@pytest.mark.parametrize("curve,mode", [("bls12_381", "bucket"), ("bn254", "base")])
def test_eight_rank_sharded_proof_gloo(curve, mode):
    """the world size of the driver's scaling run: b mod 8 residue classes (bucket space) / eight base ranges, the record all-gather
    over eight ranks and g16_finalize_host folding eight parts (r = 0, s = 0 included) -- on CPU, over gloo"""
    _run_world(8, curve, mode)


def test_shard_ranges_tile():
    import groth16_amd as g

    for (m, w, hl, nin) in ((1024, 1023, 1023, 2), (100, 96, 127, 5), (7, 3, 15, 5)):
        for cnt in (1, 2, 3, 8):
            prev = dict(a=0, l=0, h=0)
            for idx in range(cnt):
                rg = g.shard_ranges(m, w, hl, nin, idx, cnt)
                for k in prev:
                    assert rg[k][0] == prev[k] and rg[k][1] >= rg[k][0]
                    prev[k] = rg[k][1]
                # l stays inside the a shard's scalar window so that the witness sort is shared
                a_lo, a_hi = rg["a"]
                l_lo, l_hi = rg["l"]
                assert l_hi == l_lo or (l_lo + nin - 1 >= a_lo and l_hi + nin - 1 <= a_hi)
            assert prev == dict(a=m, l=w, h=hl)
