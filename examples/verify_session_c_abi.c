/* The second client's verifier side from plain C: a proof of a deferred-precompile session (the twelve AIRs of `ChipletAir::all()`,
 * precompiles-prover/src/session/prove.rs:111-126) is read from a file and checked with mh_verify_ex + the library's own
 * `ChipletMultiAir::eval_external` (mh_external_precompile_session) -- what `VerifierInstance::verify` does for
 * `ChipletMultiAir` (session/prove.rs:330-352).  Host only: no GPU, no Python on this side.
 *
 *   gcc -O2 -Iinclude examples/verify_session_c_abi.c -Lmiden-vm_amd/lib -lmidenhip -Wl,-rpath,miden-vm_amd/lib -o verify_session
 *   ./verify_session session_proof.bin [--root-off-by-one] [--ec-only]
 *
 * File = little-endian u64 words (written by tests/test_session_c_abi.py from the statement and the proof):
 *   magic "MHSESS01" | n_airs | n_public | n_pre | n_fields | n_commitments | has_preprocessed_root
 *   7 parameters (mh_pcs_params order) | n_airs log heights | public values | 12 challenger-state words | pre_observe
 *   fields | commitments (4 words each) | preprocessed root (4 words, if any) | per AIR: blob length, blob words */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "midenhip.h"

static uint64_t* read_words(FILE* f, size_t n) {
  uint64_t* p = (uint64_t*)malloc((n ? n : 1) * sizeof(uint64_t));
  if (!p || fread(p, sizeof(uint64_t), n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
  return p;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s session_proof.bin [--root-off-by-one] [--ec-only]\n", argv[0]);
    return 2;
  }
  int forge_root = 0, ec_only = 0;
  for (int i = 2; i < argc; i++) {
    if (!strcmp(argv[i], "--root-off-by-one")) forge_root = 1;
    if (!strcmp(argv[i], "--ec-only")) ec_only = 1;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  uint64_t* hdr = read_words(f, 7);
  if (memcmp(hdr, "MHSESS01", 8) != 0) {
    fprintf(stderr, "not a session proof file\n");
    return 2;
  }
  const size_t n_airs = hdr[1], n_public = hdr[2], n_pre = hdr[3], n_fields = hdr[4], n_commitments = hdr[5], has_root = hdr[6];
  uint64_t* prm = read_words(f, 7);
  mh_pcs_params params = {(int)prm[0], (int)prm[1], (int)prm[2], (int)prm[3], (int)prm[4], (int)prm[5], (int)prm[6]};
  uint64_t* heights64 = read_words(f, n_airs);
  uint8_t* heights = (uint8_t*)malloc(n_airs);
  for (size_t i = 0; i < n_airs; i++) heights[i] = (uint8_t)heights64[i];
  uint64_t* publics = read_words(f, n_public);
  uint64_t* state = read_words(f, 12);
  uint64_t* pre = read_words(f, n_pre);
  uint64_t* fields = read_words(f, n_fields);
  uint64_t* commitments = read_words(f, 4 * n_commitments);
  uint64_t* root = has_root ? read_words(f, 4) : NULL;
  const uint64_t** blobs = (const uint64_t**)malloc(n_airs * sizeof(uint64_t*));
  size_t* blob_words = (size_t*)malloc(n_airs * sizeof(size_t));
  for (size_t i = 0; i < n_airs; i++) {
    uint64_t* len = read_words(f, 1);
    blob_words[i] = (size_t)len[0];
    blobs[i] = read_words(f, blob_words[i]);
    free(len);
  }
  fclose(f);
  if (forge_root) publics[0] += 1; /* another transcript root: the statement the proof is NOT of (pre_observe still frames the true one) */

  uint64_t digest[4] = {0, 0, 0, 0};
  char err[512] = {0};
  int rc = mh_verify_ex(&params, (int)n_airs, blobs, blob_words, heights, publics, n_public, state, pre, n_pre, fields, n_fields, commitments,
                        n_commitments, root, ec_only ? mh_external_precompile_session_ec_only : mh_external_precompile_session, NULL, digest, err, sizeof err);
  if (rc == MH_OK)
    printf("ACCEPTED digest %016llx%016llx%016llx%016llx\n", (unsigned long long)digest[0], (unsigned long long)digest[1],
           (unsigned long long)digest[2], (unsigned long long)digest[3]);
  else
    printf("REJECTED rc %d: %s\n", rc, err);
  return rc == MH_OK ? 0 : 1;
}
