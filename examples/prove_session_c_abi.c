/* A deferred-precompile session proof from plain C: the shape of `SessionTraces::prove_stark`
 * (precompiles-prover/src/session/prove.rs:295-330, 385-416) through libmidenhip.
 *
 * Input: one statement file (what the session's trace builders hand the prover) --
 *     u64 magic "MHPCSES1"
 *     u64 log_heights[12]                  `ChipletAir::all()` order: ChunkNode, Poseidon2, KeccakRound, BytePairLut, KeccakSponge,
 *                                          TranscriptEval, UintStoreMul, UintAdd, EcGroups, EcPointStore, EcGroupAdd, EcMsm
 *     u64 public_root[4]                   the transcript root (`air_inputs`)
 *     u64 main_i[2^log_heights[i]][width_i]   row-major, little endian, widths 42 32 68 3 67 39 44 30 6 14 21 38
 * tests/test_gpu_precompile_c_abi.py writes a session (Keccak-256 claims, a 256-bit arithmetic claim, an EC addition and an MSM
 * claim folded into one root) in this form and compares the printed digest and the proof bytes with the CPU oracle's.
 *
 * No Python, no C++, no constraint system on the caller's side: the twelve AIRs, their lookup programs, the byte-pair table and its
 * setup commitment, the transcript framing and `ChipletMultiAir::eval_external` live in the library
 * (mh_precompile_load / mh_prove_precompile / mh_verify_precompile).
 *
 *   gcc -O2 -Iinclude examples/prove_session_c_abi.c -Lmiden-vm_amd/lib -lmidenhip -Wl,-rpath,$PWD/miden-vm_amd/lib -o prove_session
 *   ./prove_session session.bin [hash_fn = 0 Poseidon2 | 1 Blake3_256 | 2 Keccak | 3 Rpo256 | 4 Rpx256] [proof_out.bin]
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "midenhip.h"

#define CHECK(call)                                                                              \
  do {                                                                                           \
    int rc_ = (call);                                                                            \
    if (rc_ != MH_OK) {                                                                          \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctx ? mh_last_error(ctx) : "no ctx"); \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)

static const size_t WIDTHS[MH_PRECOMPILE_NUM_AIRS] = {42, 32, 68, 3, 67, 39, 44, 30, 6, 14, 21, 38};

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
    fprintf(stderr, "usage: %s session.bin [hash_fn] [proof_out.bin]\n", argv[0]);
    return 2;
  }
  const int hash_fn = argc > 2 ? atoi(argv[2]) : MH_LMCS_POSEIDON2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  uint64_t* head = read_words(f, 1 + MH_PRECOMPILE_NUM_AIRS);
  if (memcmp(head, "MHPCSES1", 8) != 0) {
    fprintf(stderr, "not a session statement file\n");
    return 2;
  }
  int log_heights[MH_PRECOMPILE_NUM_AIRS];
  for (int i = 0; i < MH_PRECOMPILE_NUM_AIRS; i++) log_heights[i] = (int)head[1 + i];
  uint64_t* public_root = read_words(f, 4);
  const uint64_t* mains[MH_PRECOMPILE_NUM_AIRS];
  for (int i = 0; i < MH_PRECOMPILE_NUM_AIRS; i++) mains[i] = read_words(f, WIDTHS[i] << log_heights[i]);
  fclose(f);

  mh_ctx* ctx = NULL;
  CHECK(mh_ctx_create(0, &ctx));
  mh_precompile* session = NULL;
  CHECK(mh_precompile_load(ctx, &session)); /* the twelve AIRs + lookup programs + the byte-pair table; kernels come from the cache */
  mh_proof* proof = NULL;
  CHECK(mh_prove_precompile(ctx, session, hash_fn, mains, log_heights, public_root, &proof));

  const uint64_t* d = mh_proof_digest(proof);
  const size_t n_bytes = mh_proof_serialize(proof, NULL, 0);
  uint8_t* bytes = (uint8_t*)malloc(n_bytes);
  if (!bytes || mh_proof_serialize(proof, bytes, n_bytes) != n_bytes) return 3;
  printf("proof: %zu fields, %zu commitments, %zu bytes\n", mh_proof_num_fields(proof), mh_proof_num_commitments(proof), n_bytes);
  printf("digest %016" PRIx64 " %016" PRIx64 " %016" PRIx64 " %016" PRIx64 "\n", d[0], d[1], d[2], d[3]);

  /* the verifier's side (`verify_stark`, session/prove.rs:365-425): bytes + the public root + the setup commitment */
  uint64_t setup[4], vd[4];
  CHECK(mh_precompile_preprocessed_root(session, hash_fn, setup));
  printf("setup %016" PRIx64 " %016" PRIx64 " %016" PRIx64 " %016" PRIx64 "\n", setup[0], setup[1], setup[2], setup[3]);
  char err[256] = "";
  int rc = mh_verify_precompile(hash_fn, setup, public_root, bytes, n_bytes, vd, err, sizeof err);
  if (rc != MH_OK || vd[0] != d[0] || vd[1] != d[1] || vd[2] != d[2] || vd[3] != d[3]) {
    fprintf(stderr, "mh_verify_precompile refused the proof: %s\n", err);
    return 4;
  }
  printf("verified\n");
  /* another transcript root is not accepted */
  public_root[0] ^= 1;
  rc = mh_verify_precompile(hash_fn, setup, public_root, bytes, n_bytes, vd, err, sizeof err);
  public_root[0] ^= 1;
  if (rc == MH_OK) {
    fprintf(stderr, "a forged root was accepted\n");
    return 5;
  }
  printf("forged root refused: %s\n", err);
  if (argc > 3) {
    FILE* o = fopen(argv[3], "wb");
    if (!o || fwrite(bytes, 1, n_bytes, o) != n_bytes) return 6;
    fclose(o);
  }
  mh_proof_free(proof);
  mh_precompile_free(session);
  mh_ctx_destroy(ctx);
  free(bytes);
  return 0;
}
