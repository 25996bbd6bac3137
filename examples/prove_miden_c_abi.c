/* A REAL Miden proof from plain C: the shape of `miden_prover::prove_stark` (prover/src/lib.rs:317-355) through libmidenhip.
 *
 * Input: one statement file (what the processor hands the prover) --
 *     u64 log_core, log_chiplets, log_poseidon2, n_aux_inputs
 *     u64 public_values[32]                      stack inputs ++ stack outputs
 *     u64 aux_inputs[n_aux_inputs]               program hash | deferred root | kernel digests
 *     u64 core[2^log_core][51], chiplets[2^log_chiplets][22], poseidon2[2^log_poseidon2][16]     row-major, little endian
 * tests/test_gpu_miden_c_abi.py writes the reference processor's snapshot case 13 (the SYSCALL program, a non-empty kernel:
 * processor/src/trace/parallel/snapshots/..case_13.snap) in this form and compares the printed digest with the CPU oracle's.
 *
 * No Python, no C++, no constraint system on the caller's side: the three AIRs, their LogUp programs, the observe schedule of
 * `MidenMultiAir` and its external assertion live in the library (mh_miden_load / mh_prove_miden / mh_verify_miden).
 *
 *   gcc -O2 -Iinclude examples/prove_miden_c_abi.c -Lmiden-vm_amd/lib -lmidenhip -Wl,-rpath,$PWD/miden-vm_amd/lib -o prove_miden
 *   ./prove_miden statement.bin [hash_fn = 0 Poseidon2 | 1 Blake3_256 | 2 Keccak | 3 Rpo256 | 4 Rpx256] [proof_out.bin]
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include "midenhip.h"

#define CHECK(call)                                                                              \
  do {                                                                                           \
    int rc_ = (call);                                                                            \
    if (rc_ != MH_OK) {                                                                          \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctx ? mh_last_error(ctx) : "no ctx"); \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)

static uint64_t* read_words(FILE* f, size_t n) {
  uint64_t* p = (uint64_t*)malloc((n ? n : 1) * sizeof(uint64_t));
  if (!p || fread(p, sizeof(uint64_t), n, f) != n) {
    fprintf(stderr, "short statement file\n");
    exit(2);
  }
  return p;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s statement.bin [hash_fn] [proof_out.bin]\n", argv[0]);
    return 2;
  }
  const int hash_fn = argc > 2 ? atoi(argv[2]) : MH_LMCS_POSEIDON2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  uint64_t* head = read_words(f, 4);
  const int log_core = (int)head[0], log_chip = (int)head[1], log_p2 = (int)head[2];
  const size_t n_aux = (size_t)head[3];
  uint64_t* public_values = read_words(f, MH_MIDEN_NUM_PUBLIC_VALUES);
  uint64_t* aux_inputs = read_words(f, n_aux);
  uint64_t* core = read_words(f, ((size_t)51) << log_core);
  uint64_t* chiplets = read_words(f, ((size_t)22) << log_chip);
  uint64_t* poseidon2 = read_words(f, ((size_t)16) << log_p2);
  fclose(f);

  mh_ctx* ctx = NULL;
  CHECK(mh_ctx_create(0, &ctx));
  mh_miden* miden = NULL;
  CHECK(mh_miden_load(ctx, &miden)); /* the three AIRs + lookup programs; compiled kernels come from the cache */
  mh_proof* proof = NULL;
  CHECK(mh_prove_miden(ctx, miden, hash_fn, core, log_core, chiplets, log_chip, poseidon2, log_p2, public_values, aux_inputs, n_aux, &proof));

  const uint64_t* d = mh_proof_digest(proof);
  const size_t n_bytes = mh_proof_serialize(proof, NULL, 0);
  uint8_t* bytes = (uint8_t*)malloc(n_bytes);
  if (!bytes || mh_proof_serialize(proof, bytes, n_bytes) != n_bytes) return 3;
  printf("proof: %zu fields, %zu commitments, %zu bytes\n", mh_proof_num_fields(proof), mh_proof_num_commitments(proof), n_bytes);
  printf("digest %016" PRIx64 " %016" PRIx64 " %016" PRIx64 " %016" PRIx64 "\n", d[0], d[1], d[2], d[3]);

  /* the verifier's side: bytes + statement in, nothing else (verifier/src/lib.rs:320-330) */
  uint64_t vd[4];
  char err[256] = "";
  int rc = mh_verify_miden(hash_fn, public_values, aux_inputs, n_aux, bytes, n_bytes, vd, err, sizeof err);
  if (rc != MH_OK || vd[0] != d[0] || vd[1] != d[1] || vd[2] != d[2] || vd[3] != d[3]) {
    fprintf(stderr, "mh_verify_miden refused the proof: %s\n", err);
    return 4;
  }
  printf("verified\n");
  /* a different claimed output is not accepted */
  public_values[16] ^= 1;
  rc = mh_verify_miden(hash_fn, public_values, aux_inputs, n_aux, bytes, n_bytes, vd, err, sizeof err);
  public_values[16] ^= 1;
  if (rc == MH_OK) {
    fprintf(stderr, "a forged stack output was accepted\n");
    return 5;
  }
  printf("forged output refused: %s\n", err);
  if (argc > 3) {
    FILE* o = fopen(argv[3], "wb");
    if (!o || fwrite(bytes, 1, n_bytes, o) != n_bytes) return 6;
    fclose(o);
  }
  mh_proof_free(proof);
  mh_miden_free(miden);
  mh_ctx_destroy(ctx);
  free(bytes); free(head); free(public_values); free(aux_inputs); free(core); free(chiplets); free(poseidon2);
  return 0;
}
