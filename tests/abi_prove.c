/* A strict-C99 program that PROVES through include/g16_mi355x.h -- what a Rust `extern "C"` author would write first:
 *   g16_ctx_create -> g16_pk_load -> g16_circuit_load -> g16_prove (host assignment) -> g16_serialize_points,
 * i.e. Groth16::create_proof_with_reduction_and_matrices (/root/reference/src/prover.rs:26-51) and the Proof's canonical bytes
 * (src/data_structures.rs:8-16), with no Python and no ctypes mirror in the process.  The case comes from a flat little-endian
 * file written by tests/test_gpu_c_prover.py (one golden case of tests/golden/<curve>.json, or an oracle-generated one): the
 * expected affine proof and its expected compressed bytes are in the file; this program compares and reports as one JSON line.
 *
 * File layout (u64 words unless said otherwise):
 *   magic "G16CASE1", curve, FQ limbs L, num_inputs, num_constraints, num_variables, domain_size - 1 (= h_query length),
 *   nnz[3], bytes of the expected compressed proof,
 *   alpha_g1[2L] beta_g1[2L] delta_g1[2L] beta_g2[4L] delta_g2[4L],
 *   a_query[num_variables][2L], b_g1_query[num_variables][2L], b_g2_query[num_variables][4L], h_query[..][2L],
 *   l_query[num_variables - num_inputs][2L],
 *   per matrix A, B, C: row_ptr[num_constraints + 1], col[nnz] (one u64 each), val[nnz][4],
 *   z[num_variables][4], r[4], s[4], expected proof a[2L] b[4L] c[2L], expected bytes (padded to 8). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "g16_mi355x.h"

static uint64_t* words = NULL;
static size_t pos = 0, total = 0;

static const uint64_t* take(size_t n) {
    const uint64_t* p = words + pos;
    if (pos + n > total) {
        fprintf(stderr, "case file too short\n");
        exit(2);
    }
    pos += n;
    return p;
}

int main(int argc, char** argv) {
    FILE* f;
    long bytes;
    uint64_t curve, L, nin, nc, nv, hlen, nnz[3], nbytes;
    const uint64_t *alpha, *beta1, *delta1, *beta2, *delta2, *aq, *b1q, *b2q, *hq, *lq, *z, *r, *s, *want, *want_bytes;
    g16_csr_view abc[3];
    uint32_t* col32[3];
    g16_pk_view view;
    g16_ctx* ctx = NULL;
    g16_pk* pk = NULL;
    g16_circuit* ck = NULL;
    g16_proof proof, proof2;
    g16_pk_info info;
    g16_timings tm;
    uint8_t out_bytes[512];
    uint64_t sz1, sz2;
    int m, rc, same, same2, bytes_ok, device = 0;
    size_t i;

    if (argc < 2) {
        fprintf(stderr, "usage: abi_prove <case file> [device]\n");
        return 2;
    }
    if (argc > 2) device = atoi(argv[2]);
    f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    total = (size_t)bytes / 8;
    words = (uint64_t*)malloc((size_t)bytes + 8);
    if (!words || fread(words, 1, (size_t)bytes, f) != (size_t)bytes) return 2;
    fclose(f);

    if (memcmp(take(1), "G16CASE1", 8) != 0) return 2;
    curve = *take(1); L = *take(1); nin = *take(1); nc = *take(1); nv = *take(1); hlen = *take(1);
    for (m = 0; m < 3; ++m) nnz[m] = *take(1);
    nbytes = *take(1);
    alpha = take(2 * L); beta1 = take(2 * L); delta1 = take(2 * L); beta2 = take(4 * L); delta2 = take(4 * L);
    aq = take(nv * 2 * L); b1q = take(nv * 2 * L); b2q = take(nv * 4 * L); hq = take(hlen * 2 * L); lq = take((nv - nin) * 2 * L);
    for (m = 0; m < 3; ++m) {
        const uint64_t* c64;
        abc[m].row_ptr = take(nc + 1);
        c64 = take(nnz[m]);
        col32[m] = (uint32_t*)malloc((nnz[m] ? nnz[m] : 1) * sizeof(uint32_t));
        if (!col32[m]) return 2;
        for (i = 0; i < nnz[m]; ++i) col32[m][i] = (uint32_t)c64[i];
        abc[m].col = col32[m];
        abc[m].val = take(nnz[m] * 4);
    }
    z = take(nv * 4); r = take(4); s = take(4); want = take(8 * L); want_bytes = take((nbytes + 7) / 8);

    /* &ProvingKey<E> as the prover reads it (include/g16_mi355x.h: query[0] apart, a / b_g1 / b_g2 in the index space query[1..]) */
    memset(&view, 0, sizeof(view));
    view.alpha_g1 = alpha; view.beta_g1 = beta1; view.delta_g1 = delta1; view.beta_g2 = beta2; view.delta_g2 = delta2;
    view.a_query0 = aq; view.b_g1_query0 = b1q; view.b_g2_query0 = b2q;
    view.a.points = aq + 2 * L; view.a.count = nv - 1;
    view.b_g1.points = b1q + 2 * L; view.b_g1.count = nv - 1;
    view.b_g2.points = b2q + 4 * L; view.b_g2.count = nv - 1;
    view.h.points = hq; view.h.count = hlen;
    view.l.points = lq; view.l.count = nv - nin;

    rc = g16_ctx_create((int)curve, device, &ctx);
    if (rc) { printf("{\"stage\": \"ctx_create\", \"rc\": %d, \"error\": \"%s\"}\n", rc, g16_strerror(rc)); return 1; }
    rc = g16_pk_load(ctx, &view, &pk);
    if (rc) { printf("{\"stage\": \"pk_load\", \"rc\": %d, \"error\": \"%s | %s\"}\n", rc, g16_strerror(rc), g16_last_error()); return 1; }
    rc = g16_circuit_load(ctx, abc, nin, nc, nv, &ck);
    if (rc) { printf("{\"stage\": \"circuit_load\", \"rc\": %d, \"error\": \"%s\"}\n", rc, g16_strerror(rc)); return 1; }
    memset(&proof, 0xff, sizeof(proof));
    rc = g16_prove(ctx, pk, ck, z, nv, 0, r, s, &proof);
    if (rc) { printf("{\"stage\": \"prove\", \"rc\": %d, \"error\": \"%s | %s\"}\n", rc, g16_strerror(rc), g16_last_error()); return 1; }
    rc = g16_prove(ctx, pk, ck, z, nv, 0, r, s, &proof2);   /* again: arena reuse, the key's fixed-base tables built by now */
    if (rc) { printf("{\"stage\": \"prove2\", \"rc\": %d}\n", rc); return 1; }
    same = memcmp(proof.a, want, 2 * L * 8) == 0 && memcmp(proof.b, want + 2 * L, 4 * L * 8) == 0 && memcmp(proof.c, want + 6 * L, 2 * L * 8) == 0;
    same2 = memcmp(proof2.a, proof.a, 2 * L * 8) == 0 && memcmp(proof2.b, proof.b, 4 * L * 8) == 0 && memcmp(proof2.c, proof.c, 2 * L * 8) == 0;

    /* Proof::serialize_compressed: a || b || c (src/data_structures.rs:8-16) */
    sz1 = g16_serialized_point_size((int)curve, 0, 1);
    sz2 = g16_serialized_point_size((int)curve, 1, 1);
    bytes_ok = 0;
    if (sz1 + sz2 + sz1 <= sizeof(out_bytes) && sz1 + sz2 + sz1 == nbytes &&
        g16_serialize_points((int)curve, 0, 1, proof.a, 1, out_bytes) == 0 &&
        g16_serialize_points((int)curve, 1, 1, proof.b, 1, out_bytes + sz1) == 0 &&
        g16_serialize_points((int)curve, 0, 1, proof.c, 1, out_bytes + sz1 + sz2) == 0)
        bytes_ok = memcmp(out_bytes, want_bytes, (size_t)nbytes) == 0;

    memset(&info, 0, sizeof(info));
    memset(&tm, 0, sizeof(tm));
    rc = g16_pk_get_info(pk, &info);
    rc |= g16_get_timings(ctx, &tm);
    printf("{\"stage\": \"done\", \"rc\": %d, \"proof_matches\": %d, \"second_proof_identical\": %d, \"bytes_match\": %d, \"proof_bytes\": %lu, "
           "\"window_bits\": %d, \"table_fallback\": %d, \"domain_size\": %lu, \"total_ms\": %.3f, \"version\": \"%s\"}\n",
           rc, same, same2, bytes_ok, (unsigned long)nbytes, info.window_bits_z, info.table_fallback,
           (unsigned long)g16_circuit_domain_size(ck), tm.total_ms, g16_version());
    g16_circuit_free(ck);
    g16_pk_free(pk);
    g16_ctx_destroy(ctx);
    for (m = 0; m < 3; ++m) free(col32[m]);
    free(words);
    return (same && same2 && bytes_ok && rc == 0) ? 0 : 1;
}
