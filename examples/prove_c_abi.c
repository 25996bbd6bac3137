/* A complete proof through the C ABI of libmidenhip, from plain C (what the Rust shim of INTEGRATION.md does):
 * build a DummyMidenAir constraint-DAG blob (crates/lifted-stark/src/testing/airs/miden.rs:36-95: one degree-9
 * constraint local[0] * ... * local[8] == 0 over `width` columns, `aux` all-zero EF aux columns), a deterministic
 * trace (column 0 zero, the rest from a 64-bit LCG), prove it with the production PCS parameters and print the
 * transcript digest.  tests/test_gpu_c_abi.py checks that digest against the CPU oracle.
 *
 *   gcc -O2 -Iinclude examples/prove_c_abi.c -Lmiden-vm_amd/lib -lmidenhip -Wl,-rpath,$PWD/miden-vm_amd/lib -o prove_c_abi
 *   ./prove_c_abi [log_rows=12] [width=51] [aux_ef=8]
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include "midenhip.h"

#define GL_P 0xFFFFFFFF00000001ULL
enum { OP_CONST = 0, OP_MAIN = 1, OP_MUL = 12 };

static uint64_t node(uint64_t op, uint64_t a, uint64_t b) { return op | (a << 8) | (b << 36); }

#define CHECK(call)                                                                  \
  do {                                                                               \
    int rc_ = (call);                                                                \
    if (rc_ != MH_OK) {                                                              \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctx ? mh_last_error(ctx) : "no ctx"); \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

int main(int argc, char** argv) {
  const int log_n = argc > 1 ? atoi(argv[1]) : 12;
  const size_t width = argc > 2 ? (size_t)atoi(argv[2]) : 51, aux = argc > 3 ? (size_t)atoi(argv[3]) : 8;
  /* the StarkConfig by its hash function (ProvingOptions of the reference): 0 Poseidon2, 1 Blake3_256 (its default), 2 Keccak,
   * 3 Rpo256, 4 Rpx256 */
  const int lmcs = argc > 4 ? atoi(argv[4]) : MH_LMCS_POSEIDON2;
  const size_t n = (size_t)1 << log_n;
  mh_ctx* ctx = NULL;
  CHECK(mh_ctx_create(0, &ctx));
  CHECK(mh_ctx_set_lmcs(ctx, lmcs));

  /* ---- the AIR as data: header, 1 CONST + 9 MAIN leaves + 9 MUL gates, one constraint ---- */
  uint64_t blob[12 + 2 * 19 + 1];
  size_t w = 0;
  const uint64_t header[12] = {0x4d48444147303031ULL, width, aux, /*num_randomness*/ 2, /*num_aux_values*/ aux,
                               /*publics*/ 0, /*periodic*/ 0, /*log_quotient_degree*/ 3, /*nodes*/ 19, /*constraints*/ 1, 0, 0};
  for (int i = 0; i < 12; i++) blob[w++] = header[i];
  blob[w++] = node(OP_CONST, 0, 0); blob[w++] = 1;               /* node 0: the constant ONE */
  uint64_t prod = 0, next_id = 1;
  for (uint64_t j = 0; j < 9; j++) {
    blob[w++] = node(OP_MAIN, j, 0); blob[w++] = 0;              /* local[j] */
    const uint64_t leaf = next_id++;
    blob[w++] = node(OP_MUL, prod, leaf); blob[w++] = 0;         /* prod *= local[j] */
    prod = next_id++;
  }
  blob[w++] = prod;
  mh_air* air = NULL;
  CHECK(mh_air_load(ctx, blob, w, &air));

  /* ---- the trace, built in page-locked memory so the upload is a straight DMA ---- */
  uint64_t* rows = (uint64_t*)mh_host_alloc(n * width * 8);
  if (!rows) { fprintf(stderr, "mh_host_alloc failed\n"); return 1; }
  uint64_t x = 0x9E3779B97F4A7C15ULL;
  for (size_t r = 0; r < n; r++)
    for (size_t c = 0; c < width; c++) {
      x = x * 6364136223846793005ULL + 1442695040888963407ULL;
      rows[r * width + c] = c == 0 ? 0 : x % GL_P;
    }
  /* the matrix stays on the host: mh_prove_host uploads it on the copy stream inside the call (prove_stark's own shape:
   * host RowMajorMatrix values in, proof out) */

  /* ---- prove: production parameters (air/src/config.rs:54-67); the challenger starts from the all-zero sponge and
   * has observed the protocol parameters and an empty statement (config.rs:188-198, lifted-air/src/air.rs:307-324) ---- */
  const mh_pcs_params params = {3, 2, 7, 4, 12, 27, 16};
  const uint64_t state[12] = {0};
  const uint64_t pre[11] = {27, 16, 12, 4, 3, 7, 4, 0, /*n publics*/ 0, 0, /*n aux inputs*/ 0};
  mh_air* airs[1] = {air};
  const uint64_t* host_traces[1] = {rows};
  const int log_heights[1] = {log_n};
  mh_proof* proof = NULL;
  CHECK(mh_prove_host(ctx, &params, 1, airs, host_traces, log_heights, NULL, 0, state, pre, 11, NULL, NULL, &proof));
  mh_host_free(rows);

  const uint64_t* d = mh_proof_digest(proof);
  printf("rows 2^%d width %zu aux %zu: %zu fields, %zu commitments, %zu bytes\n", log_n, width, aux, mh_proof_num_fields(proof),
         mh_proof_num_commitments(proof), mh_proof_serialize(proof, NULL, 0));
  printf("digest %016" PRIx64 " %016" PRIx64 " %016" PRIx64 " %016" PRIx64 "\n", d[0], d[1], d[2], d[3]);
  /* ---- and check it with the library's host-only verifier (no GPU involved) ---- */
  {
    const uint64_t* blobs[1] = {blob};
    const size_t blob_words[1] = {w};
    const uint8_t heights[1] = {(uint8_t)log_n};
    uint64_t vdigest[4];
    char why[256];
    int rc = mh_verify_lmcs(lmcs, &params, 1, blobs, blob_words, heights, NULL, 0, state, pre, 11, mh_proof_fields(proof),
                            mh_proof_num_fields(proof), mh_proof_commitments(proof), mh_proof_num_commitments(proof), NULL, NULL, NULL,
                            vdigest, why, sizeof why);
    if (rc != MH_OK || vdigest[0] != d[0] || vdigest[3] != d[3]) {
      fprintf(stderr, "mh_verify_lmcs: %s\n", rc != MH_OK ? why : "digest mismatch");
      return 1;
    }
    printf("verified\n");
  }
  mh_proof_free(proof);
  mh_air_free(air);
  mh_ctx_destroy(ctx);
  return 0;
}
