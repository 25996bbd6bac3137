// Internal interfaces between the translation units of libmidenhip (not part of the C ABI).
#pragma once
#include "ctx.hpp"
#include <vector>

// ---- ntt.hip ---------------------------------------------------------------------------------
void launch_transpose_rm_to_cm(mh_ctx* c, const u64* in_rowmajor, u64* out_colmajor, size_t n, size_t w);
void ntt_inverse_dif_inplace(mh_ctx* c, u64* cols, size_t n_cols, int log_n);
void ntt_forward_cosets(mh_ctx* c, const u64* coef_br, size_t n_cols, int log_n, const std::vector<u64>& bases, u64* out);
void lde_columns(mh_ctx* c, const u64* cols_in, size_t n_cols, int log_n, u64 in_shift, const std::vector<u64>& out_shifts,
                 u64* out, u64* scratch);

// ---- device-resident objects -------------------------------------------------------------------
// A trace matrix as uploaded: column-major, natural row order, canonical felts.
struct mh_trace {
  mh_ctx* ctx;
  int log_n;
  size_t width;
  DevBuf cols;  // [width][N]
};

// A committed (LDE'd) matrix: coset-major column-major: lde[(c*B + j)*N + r] = f_c(shift*w_K^j*w_H^r)
// = evaluation at natural index i = r*B + j of the max-domain-lifted polynomial.
struct LdeMatrix {
  int log_n;  // trace height
  size_t width;
  DevBuf lde;
  const u64* col(size_t c, int log_blowup) const { return lde.u() + ((c << log_blowup) << log_n); }
};

// LMCS tree over a group of LDE matrices (ascending heights).  Node (depth d, natural position p)
// lives at layer_ptr(d) + 4*node_slot(d,p): leaf-side layers are coset-major (see lmcs.hip).
struct mh_tree {
  mh_ctx* ctx;
  int log_blowup;  // number of coset bits in the leaf layer layout (may be 0)
  int log_height;  // tree depth L (leaves = 2^L)
  std::vector<LdeMatrix> mats;
  DevBuf nodes;                    // all layers, leaf layer first
  std::vector<size_t> layer_off;   // layer_off[d] = element offset (in digests) of depth-d layer
  u64 root[4];
  size_t node_slot(int d, size_t p) const {
    int cbits = d - (log_height - log_blowup);
    if (cbits <= 0) return p;
    size_t j = p & (((size_t)1 << cbits) - 1), r = p >> cbits;
    return (j << (log_height - log_blowup)) + r;
  }
};

// ---- lmcs.hip --------------------------------------------------------------------------------
void poseidon2_permute_device(mh_ctx* c, u64* states_soa, size_t n);  // [12][n]
// Build leaf digests + all layers for `t->mats` (already filled); sets t->root.
void lmcs_build_tree(mh_ctx* c, mh_tree* t);
// Gather opened rows (aligned, per sorted unique index) and missing siblings.
void lmcs_open(mh_ctx* c, const mh_tree* t, const std::vector<size_t>& sorted_unique_idx, size_t alignment,
               std::vector<u64>& fields, std::vector<u64>& commitments);
std::vector<std::pair<int, size_t>> lmcs_missing_siblings(const std::vector<size_t>& sorted_unique_idx, int depth);
