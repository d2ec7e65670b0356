/* A C99 consumer of include/g16_mi355x.h (test infrastructure).
 *
 * Every other test reaches the library through ctypes, whose structs in groth16_amd/binding.py are written by hand; a Rust
 * `extern "C"` block would be written by hand too.  This program is what keeps the header honest: it must
 * compile as strict C99 (gcc -std=c99 -pedantic -Wall -Wextra -Werror), link against libg16_mi355x.so, and it prints the
 * size / field offsets of every struct the ABI passes by pointer -- tests/test_abi_consumer.py compares them with binding.py --
 * then runs the CPU-side entry points through real C calls: version / error strings, the randomized self-test of the
 * 30-bit arithmetic on both curves, a field operation, and g16_ctx_create (which must either give a context or fail
 * with G16_ERR_NO_DEVICE -- never crash -- when there is no GPU).
 */
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "g16_mi355x.h"

#define FIELD(T, f) printf("    \"%s\": [%lu, %lu],\n", #f, (unsigned long)offsetof(T, f), (unsigned long)sizeof(((T*)0)->f))
#define BEGIN(T) printf("  \"%s\": {\n", #T)
#define END(T) printf("    \"sizeof\": [0, %lu]\n  },\n", (unsigned long)sizeof(T))

static void layouts(void) {
    BEGIN(g16_query); FIELD(g16_query, points); FIELD(g16_query, count); FIELD(g16_query, start); END(g16_query);
    BEGIN(g16_pk_view);
    FIELD(g16_pk_view, alpha_g1); FIELD(g16_pk_view, beta_g1); FIELD(g16_pk_view, delta_g1); FIELD(g16_pk_view, beta_g2);
    FIELD(g16_pk_view, delta_g2); FIELD(g16_pk_view, a_query0); FIELD(g16_pk_view, b_g1_query0); FIELD(g16_pk_view, b_g2_query0);
    FIELD(g16_pk_view, a); FIELD(g16_pk_view, b_g1); FIELD(g16_pk_view, b_g2); FIELD(g16_pk_view, h); FIELD(g16_pk_view, l);
    FIELD(g16_pk_view, flags);
    END(g16_pk_view);
    BEGIN(g16_csr_view); FIELD(g16_csr_view, row_ptr); FIELD(g16_csr_view, col); FIELD(g16_csr_view, val); END(g16_csr_view);
    BEGIN(g16_proof); FIELD(g16_proof, a); FIELD(g16_proof, b); FIELD(g16_proof, c); END(g16_proof);
    BEGIN(g16_partial);
    FIELD(g16_partial, h); FIELD(g16_partial, l); FIELD(g16_partial, a); FIELD(g16_partial, b_g1); FIELD(g16_partial, b_g2);
    END(g16_partial);
    BEGIN(g16_timings);
    FIELD(g16_timings, witness_map_ms); FIELD(g16_timings, msm_h_ms); FIELD(g16_timings, msm_l_ms); FIELD(g16_timings, msm_a_ms);
    FIELD(g16_timings, msm_b_g1_ms); FIELD(g16_timings, msm_b_g2_ms); FIELD(g16_timings, scalar_prep_ms); FIELD(g16_timings, finish_ms);
    FIELD(g16_timings, total_ms); FIELD(g16_timings, bucket_pass_ms); FIELD(g16_timings, bucket_ms); FIELD(g16_timings, window_bits);
    FIELD(g16_timings, windows); FIELD(g16_timings, ntt_ms); FIELD(g16_timings, g1_pass_launches);
    END(g16_timings);
    BEGIN(g16_pk_info);
    FIELD(g16_pk_info, window_bits_z); FIELD(g16_pk_info, window_bits_h); FIELD(g16_pk_info, table_fallback);
    FIELD(g16_pk_info, bucket_shard_rank); FIELD(g16_pk_info, bucket_shard_world); FIELD(g16_pk_info, n_devices);
    FIELD(g16_pk_info, device_bytes);
    END(g16_pk_info);
    BEGIN(g16_diag);
    FIELD(g16_diag, mad_per_s); FIELD(g16_diag, mads_per_add_g1); FIELD(g16_diag, mads_per_add_g2); FIELD(g16_diag, mads_per_product);
    FIELD(g16_diag, limbs);
    END(g16_diag);
    BEGIN(g16_toxic_waste);
    FIELD(g16_toxic_waste, alpha); FIELD(g16_toxic_waste, beta); FIELD(g16_toxic_waste, gamma); FIELD(g16_toxic_waste, delta);
    FIELD(g16_toxic_waste, t);
    END(g16_toxic_waste);
    BEGIN(g16_params_view);
    FIELD(g16_params_view, alpha_g1); FIELD(g16_params_view, beta_g1); FIELD(g16_params_view, delta_g1); FIELD(g16_params_view, beta_g2);
    FIELD(g16_params_view, delta_g2); FIELD(g16_params_view, gamma_g2); FIELD(g16_params_view, gamma_abc_g1);
    FIELD(g16_params_view, a_query); FIELD(g16_params_view, b_g1_query); FIELD(g16_params_view, b_g2_query);
    FIELD(g16_params_view, h_query); FIELD(g16_params_view, l_query); FIELD(g16_params_view, flags);
    END(g16_params_view);
}

int main(int argc, char** argv) {
    int iters = 6, curve, rc, failures = 0;
    g16_ctx* ctx = NULL;
    uint64_t one_plus_one[4], a[4], out[4];
    (void)argv;
    if (argc > 1) iters = 2;
    printf("{\n\"layout\": {\n");
    layouts();
    printf("  \"_\": {}\n},\n");
    printf("\"version\": \"%s\",\n", g16_version());
    printf("\"strerror_ok\": \"%s\",\n\"strerror_degree\": \"%s\",\n", g16_strerror(G16_OK), g16_strerror(G16_ERR_DEGREE_TOO_LARGE));
    for (curve = G16_BLS12_381; curve <= G16_BN254; ++curve) {
        rc = g16_host_selftest(curve, 12345u + (uint64_t)curve, iters);
        printf("\"selftest_%d\": %d,\n", curve, rc);
        if (rc != 0) ++failures;
        /* from_canonical(1) + from_canonical(1) == from_canonical(2), through three C calls (Fr) */
        memset(a, 0, sizeof a);
        a[0] = 1;
        rc = g16_host_field_op(curve, 0, 5, a, NULL, out);
        if (rc == G16_OK) rc = g16_host_field_op(curve, 0, 0, out, out, one_plus_one);
        a[0] = 2;
        if (rc == G16_OK) rc = g16_host_field_op(curve, 0, 5, a, NULL, out);
        if (rc != G16_OK || memcmp(out, one_plus_one, sizeof out) != 0) ++failures;
        printf("\"field_op_%d\": %d,\n", curve, rc == G16_OK && memcmp(out, one_plus_one, sizeof out) == 0);
    }
    rc = g16_ctx_create(G16_BLS12_381, 0, &ctx);
    printf("\"ctx_create\": %d,\n", rc);
    if (rc == G16_OK) {
        if (g16_ctx_num_devices(ctx) != 1) ++failures;
        g16_ctx_destroy(ctx);
    } else if (rc != G16_ERR_NO_DEVICE) {
        ++failures;
    }
    /* ABI revision: the library's struct sizes are this header's, and a SHORT caller struct is never overrun (an "old caller":
     * the first 8 bytes of g16_pk_info asked for, a canary right behind them) */
    if (g16_abi_version() != G16_ABI_VERSION) ++failures;
    if (g16_struct_size(G16_STRUCT_TIMINGS) != sizeof(g16_timings) || g16_struct_size(G16_STRUCT_PK_INFO) != sizeof(g16_pk_info) ||
        g16_struct_size(G16_STRUCT_DIAG) != sizeof(g16_diag) || g16_struct_size(G16_STRUCT_PROOF) != sizeof(g16_proof) ||
        g16_struct_size(G16_STRUCT_PARTIAL) != sizeof(g16_partial) || g16_struct_size(G16_STRUCT_PK_VIEW) != sizeof(g16_pk_view) ||
        g16_struct_size(99) != 0)
        ++failures;
    if (g16_get_timings_sized(NULL, a, sizeof a) == G16_OK || g16_pk_get_info_sized(NULL, a, 8) == G16_OK) ++failures;
    printf("\"abi_version\": %d,\n", g16_abi_version());
    rc = g16_ctx_create(7, 0, &ctx); /* no such curve */
    if (rc == G16_OK) ++failures;
    printf("\"failures\": %d\n}\n", failures);
    return failures ? 1 : 0;
}
