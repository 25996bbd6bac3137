// extern "C" boundary of libmidenhip (include/midenhip.h).  Exceptions stop here.
#include "../../include/midenhip.h"
#include "ctx.hpp"
#include "gl.cuh"
#include "kernels.hpp"
#include "blake3.cuh"
#include <algorithm>
#include <cstring>
#include <memory>

#define MH_TRY(ctx_expr) mh_ctx* _c = (ctx_expr); PoolScope _ps(_c); try {
#define MH_CATCH                                                   \
  }                                                                \
  catch (const MhError& e) {                                       \
    if (_c) _c->err = e.what();                                    \
    return e.code;                                                 \
  }                                                                \
  catch (const std::exception& e) {                                \
    if (_c) _c->err = e.what();                                    \
    return MH_ERR_INTERNAL;                                        \
  }                                                                \
  return MH_OK;

// coset-major column-major LDE -> reference layout (row-major, bit-reversed physical rows)
__global__ void k_lde_to_reference_layout(const u64* __restrict__ lde, u64* __restrict__ out, int log_n, int lb, size_t w) {
  size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t rows = (size_t)1 << (log_n + lb);
  if (t >= rows * w) return;
  size_t pr = t / w, cidx = t % w;
  size_t i = bitrev32((u32)pr, log_n + lb);
  size_t j = i & (((size_t)1 << lb) - 1), r = i >> lb;
  out[t] = lde[(((cidx << lb) + j) << log_n) + r];
}

extern "C" {

int mh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mh_ctx_set_lmcs(mh_ctx* c, int lmcs) {
  if (!c) return MH_ERR_INVALID;
  if (lmcs < MH_LMCS_POSEIDON2 || lmcs > MH_LMCS_RPX) {
    c->err = "unknown LMCS hasher id";
    return MH_ERR_INVALID;
  }
  c->lmcs = lmcs;
  return 0;
}
int mh_ctx_get_lmcs(const mh_ctx* c) { return c ? c->lmcs : -1; }
void mh_blake3(const uint8_t* data, size_t n, uint8_t out32[32]) { b3::hash_bytes(data, n, out32); }

int mh_ctx_create(int device_id, mh_ctx** out) {
  if (!out) return MH_ERR_INVALID;
  *out = nullptr;
  mh_ctx* c = new mh_ctx();
  try {
    int n = 0;
    HIP_CHECK(hipGetDeviceCount(&n));
    MH_REQUIRE(device_id >= 0 && device_id < n, "no such HIP device (libmidenhip needs an AMD GPU; there is no CPU fallback)");
    HIP_CHECK(hipSetDevice(device_id));
    c->device = device_id;
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->primary_stream = c->stream;
  } catch (const std::exception& e) {
    fprintf(stderr, "mh_ctx_create: %s\n", e.what());
    delete c;
    return MH_ERR_HIP;
  }
  *out = c;
  return MH_OK;
}

void mh_ctx_destroy(mh_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
  if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : c->event_pool) (void)hipEventDestroy(e);
  {
    PoolScope ps(c);
    c->tw_fwd.clear(); c->tw_inv.clear(); c->tables.clear(); c->table_index.clear();
  }
  c->pool.trim();
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->pinned_top) (void)hipHostFree(c->pinned_top);
  if (c->ring) (void)hipHostFree(c->ring);
  for (auto& hp : c->host_pool) (void)hipHostFree(hp.first);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
  delete c;
}

// Give the cached device buffers of this context back to the driver (a proof keeps its ~2x LDE-sized working
// set pooled between calls so that the next proof does not pay hipMalloc / hipFree).
int mh_ctx_trim(mh_ctx* c) {
  MH_TRY(c)
  MH_REQUIRE(c, "null ctx");
  HIP_CHECK(hipSetDevice(c->device));
  c->sync();
  // the full [z][pos] coset-scale tables of the LDE (ntt.hip coset_scale_full: 8 N bytes per output coset -- 64 MB per trace height at
  // blowup 8, 1 GB at 2^24 rows) are rebuilt in one launch when next needed; a service that proves varied heights would otherwise
  // accumulate gigabytes of them.  The small twiddle / coset tables stay.  They go FIRST: their buffers were taken from this pool
  // inside a prove call, so releasing them hands them to the pool's free list, and only the trim below returns them to the driver
  // (the order mh_ctx_destroy uses).
  for (auto it = c->tables.begin(); it != c->tables.end();) {
    if (it->first.rfind("cosetfull:", 0) == 0) {
      c->table_index.erase(it->first);
      it = c->tables.erase(it);
    } else {
      ++it;
    }
  }
  c->pool.trim();
  // page-locked aux scratch (host_take / host_give) is cached per size class: a long-lived context proving varied shapes would keep
  // every class for ever
  if (c->copy_stream) HIP_CHECK(hipStreamSynchronize(c->copy_stream));
  for (auto& b : c->host_pool) (void)hipHostFree(b.first);
  c->host_pool.clear();
  MH_CATCH
}

int mh_ctx_mem_stats(mh_ctx* c, uint64_t out[4]) {
  MH_TRY(c)
  MH_REQUIRE(c && out, "null argument");
  HIP_CHECK(hipSetDevice(c->device));
  size_t fr = 0, tot = 0, tab = 0;
  HIP_CHECK(hipMemGetInfo(&fr, &tot));
  for (auto& kv : c->tables) tab += kv.second.bytes;
  out[0] = c->pool.cached_bytes;
  out[1] = tab;
  out[2] = fr;
  out[3] = tot;
  MH_CATCH
}

const char* mh_last_error(const mh_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int mh_prof_enable(mh_ctx* c, int on) {
  MH_TRY(c)
  MH_REQUIRE(c, "null ctx");
  c->prof_resolve();
  c->prof_on = on != 0;
  MH_CATCH
}
int mh_prof_filter(mh_ctx* c, const char* name) {
  MH_TRY(c)
  MH_REQUIRE(c, "null ctx");
  c->prof_resolve();
  c->prof_only = name ? name : "";
  MH_CATCH
}
int mh_prof_reset(mh_ctx* c) {
  MH_TRY(c)
  MH_REQUIRE(c, "null ctx");
  c->prof_resolve();
  c->prof.clear();
  MH_CATCH
}
int mh_prof_get(mh_ctx* c, const char* name, double* ms, double* bytes, long* count) {
  MH_TRY(c)
  MH_REQUIRE(c && name, "null argument");
  c->prof_resolve();
  auto it = c->prof.find(name);
  ProfEntry e = it == c->prof.end() ? ProfEntry{} : it->second;
  if (ms) *ms = e.ms;
  if (bytes) *bytes = e.bytes;
  if (count) *count = e.count;
  MH_CATCH
}
int mh_prof_dump(mh_ctx* c, char* buf, size_t cap) {
  MH_TRY(c)
  MH_REQUIRE(c && buf && cap, "null argument");
  c->prof_resolve();
  std::string s;
  for (auto& kv : c->prof) {
    char line[256];
    snprintf(line, sizeof line, "%s %.6f %.0f %ld\n", kv.first.c_str(), kv.second.ms, kv.second.bytes, kv.second.count);
    s += line;
  }
  // The reference's tracing span names (commit.rs:172, lifted_tree.rs:229/263, quotient.rs:185,
  // deep/prover.rs:214, interpolate.rs:133, fri/prover.rs:164-183, stark-transcript grind) as aliases of the kernel classes, so
  // that a consumer of the reference's span schema (tracing-forest, blake3-bench's SpanRecorder) finds the same keys.  The
  // protocol stages themselves are recorded under "span:<reference name>" by the session (prover.hip).
  static const char* const alias[][2] = {{"lde", "span:LDE"}, {"lmcs_leaf_absorb", "span:hash leaves"}, {"lmcs_compress", "span:compress tree layers"},
                                         {"quotient_eval", "span:eval_instance"}, {"deep_ood_eval", "span:batch_eval_lifted"},
                                         {"deep_assemble", "span:DEEP reduce + assemble"}, {"grind", "span:DEEP grind + FRI folding grind + query grind"},
                                         {"transpose_in", "span:trace upload (no reference span: the CPU prover has none)"}};
  for (auto& al : alias) {
    auto it = c->prof.find(al[0]);
    if (it == c->prof.end()) continue;
    char line[256];
    snprintf(line, sizeof line, "%s %.6f %.0f %ld\n", al[1], it->second.ms, it->second.bytes, it->second.count);
    s += line;
  }
  size_t n = std::min(cap - 1, s.size());
  memcpy(buf, s.data(), n);
  buf[n] = 0;
  MH_CATCH
}

int mh_poseidon2_permute(mh_ctx* c, uint64_t* states, size_t n) {
  MH_TRY(c)
  MH_REQUIRE(c && (states || !n), "null argument");
  if (!n) return MH_OK;
  HIP_CHECK(hipSetDevice(c->device));
  DevBuf aos(n * 96), soa(n * 96);
  HIP_CHECK(hipMemcpyAsync(aos.p, states, n * 96, hipMemcpyHostToDevice, c->stream));
  launch_transpose_rm_to_cm(c, aos.u(), soa.u(), n, 12);  // [n][12] -> [12][n], canonicalises
  {
    ProfScope ps(c, "poseidon2_permute", 192.0 * n);
    poseidon2_permute_device(c, soa.u(), n);
  }
  launch_transpose_rm_to_cm(c, soa.u(), aos.u(), 12, n);  // back to [n][12]
  HIP_CHECK(hipMemcpyAsync(states, aos.p, n * 96, hipMemcpyDeviceToHost, c->stream));
  c->sync();
  MH_CATCH
}

int mh_poseidon2_register_rate(mh_ctx* c, double* perms_per_second) {
  MH_TRY(c)
  MH_REQUIRE(c && perms_per_second, "null argument");
  HIP_CHECK(hipSetDevice(c->device));
  *perms_per_second = poseidon2_register_rate(c);
  MH_CATCH
}

int mh_trace_upload(mh_ctx* c, const uint64_t* rowmajor, int log_n, size_t width, mh_trace** out) {
  MH_TRY(c)
  MH_REQUIRE(c && rowmajor && out, "null argument");
  MH_REQUIRE(log_n >= 0 && log_n <= 29 && width > 0, "bad trace shape");
  HIP_CHECK(hipSetDevice(c->device));
  *out = trace_upload(c, rowmajor, log_n, width);
  MH_CATCH
}
int mh_trace_upload_async(mh_ctx* c, const uint64_t* rowmajor, int log_n, size_t width, mh_trace** out) {
  MH_TRY(c)
  MH_REQUIRE(c && rowmajor && out, "null argument");
  MH_REQUIRE(log_n >= 0 && log_n <= 29 && width > 0, "bad trace shape");
  HIP_CHECK(hipSetDevice(c->device));
  *out = trace_upload_async(c, rowmajor, log_n, width);
  MH_CATCH
}
int mh_trace_upload_cols_async(mh_ctx* c, const uint64_t* colmajor, int log_n, size_t width, mh_trace** out) {
  MH_TRY(c)
  MH_REQUIRE(c && colmajor && out, "null argument");
  MH_REQUIRE(log_n >= 0 && log_n <= 29 && width > 0, "bad trace shape");
  HIP_CHECK(hipSetDevice(c->device));
  *out = trace_upload_cols_async(c, colmajor, log_n, width);
  MH_CATCH
}
int mh_trace_wait(mh_ctx* c, mh_trace* t) {
  MH_TRY(c)
  MH_REQUIRE(c && t, "null argument");
  HIP_CHECK(hipSetDevice(c->device));
  if (t->ready) HIP_CHECK(hipEventSynchronize(t->ready));
  {
    PoolScope ps(c);
    t->staging.release();  // the DMA has landed and been transposed: the landing buffer goes back to the pool
  }
  MH_CATCH
}
// A trace that already lives in device memory (row-major, any stream-ordered producer finished): transposed and
// canonicalised on the device, no PCIe traffic (trace generation hand-off).
int mh_trace_from_device(mh_ctx* c, const uint64_t* device_rowmajor, int log_n, size_t width, mh_trace** out) {
  MH_TRY(c)
  MH_REQUIRE(c && device_rowmajor && out, "null argument");
  MH_REQUIRE(log_n >= 0 && log_n <= 29 && width > 0, "bad trace shape");
  HIP_CHECK(hipSetDevice(c->device));
  hipPointerAttribute_t attr;
  MH_REQUIRE(hipPointerGetAttributes(&attr, device_rowmajor) == hipSuccess && attr.type == hipMemoryTypeDevice,
             "mh_trace_from_device needs a device pointer (use mh_trace_upload for host memory)");
  const size_t n = (size_t)1 << log_n;
  std::unique_ptr<mh_trace> t(new mh_trace());
  t->ctx = c; t->log_n = log_n; t->width = width;
  t->cols.alloc(n * width * 8);
  launch_transpose_rm_to_cm(c, device_rowmajor, t->cols.u(), n, width);
  c->sync();
  *out = t.release();
  MH_CATCH
}
// Page-locked host memory for traces: a RowMajorMatrix built in it is DMA'd at PCIe line rate by
// mh_trace_upload instead of being staged through the runtime's bounce buffers.
void* mh_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void mh_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}
void mh_trace_free(mh_trace* t) {
  if (!t) return;
  (void)hipSetDevice(t->ctx->device);
  PoolScope ps(t->ctx);
  delete t;
}

int mh_commit_traces(mh_ctx* c, int n_traces, mh_trace* const* traces, int log_blowup, mh_tree** out, uint64_t root[4]) {
  MH_TRY(c)
  MH_REQUIRE(c && traces && out && n_traces > 0, "null/empty argument");
  MH_REQUIRE(log_blowup >= 0 && log_blowup <= 8, "bad log_blowup");
  HIP_CHECK(hipSetDevice(c->device));
  std::vector<const mh_trace*> v;
  for (int i = 0; i < n_traces; i++) {
    MH_REQUIRE(traces[i], "null trace");
    v.push_back(traces[i]);
  }
  std::unique_ptr<mh_tree> t(commit_traces(c, v, log_blowup));
  if (root) memcpy(root, t->root, 32);
  *out = t.release();
  MH_CATCH
}
void mh_tree_free(mh_tree* t) {
  if (!t) return;
  (void)hipSetDevice(t->ctx->device);
  PoolScope ps(t->ctx);
  delete t;
}
int mh_tree_root(const mh_tree* t, uint64_t root[4]) {
  if (!t || !root) return MH_ERR_INVALID;
  memcpy(root, t->root, 32);
  return MH_OK;
}
int mh_tree_log_height(const mh_tree* t) { return t ? t->log_height : -1; }

int mh_tree_open(mh_ctx* c, const mh_tree* t, const uint64_t* indices, size_t n_idx, size_t alignment, uint64_t* fields,
                 size_t* n_fields, uint64_t* commits, size_t* n_commit_felts) {
  MH_TRY(c)
  MH_REQUIRE(c && t && (indices || !n_idx) && n_fields && n_commit_felts, "null argument");
  MH_REQUIRE(alignment > 0, "alignment must be non-zero");
  HIP_CHECK(hipSetDevice(c->device));
  std::vector<size_t> idx(indices, indices + n_idx);
  std::sort(idx.begin(), idx.end());
  idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
  std::vector<u64> f, cm;
  lmcs_open(c, t, idx, alignment, f, cm);
  if (!f.empty()) memcpy(fields, f.data(), f.size() * 8);
  if (!cm.empty()) memcpy(commits, cm.data(), cm.size() * 8);
  *n_fields = f.size();
  *n_commit_felts = cm.size();
  MH_CATCH
}

int mh_tree_download_lde(mh_ctx* c, const mh_tree* t, int mat, uint64_t* out) {
  MH_TRY(c)
  MH_REQUIRE(c && t && out && mat >= 0 && (size_t)mat < t->mats.size(), "bad argument");
  HIP_CHECK(hipSetDevice(c->device));
  const LdeMatrix& m = t->mats[mat];
  size_t total = (((size_t)1 << m.log_n) << t->log_blowup) * m.width;
  DevBuf tmp(total * 8);
  MH_LAUNCH(k_lde_to_reference_layout, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, m.lde.u(),
                     tmp.u(), m.log_n, t->log_blowup, m.width);
  HIP_CHECK(hipMemcpyAsync(out, tmp.p, total * 8, hipMemcpyDeviceToHost, c->stream));
  c->sync();
  MH_CATCH
}

int mh_tree_download_layers(mh_ctx* c, const mh_tree* t, uint64_t* out) {
  MH_TRY(c)
  MH_REQUIRE(c && t && out, "bad argument");
  HIP_CHECK(hipSetDevice(c->device));
  size_t total = ((size_t)2 << t->log_height) - 1;
  std::vector<u64> raw(total * 4);
  HIP_CHECK(hipMemcpyAsync(raw.data(), t->nodes.p, total * 32, hipMemcpyDeviceToHost, c->stream));
  c->sync();
  size_t o = 0;
  for (int d = t->log_height; d >= 0; d--)
    for (size_t p = 0; p < ((size_t)1 << d); p++, o += 4)
      memcpy(out + o, raw.data() + 4 * (t->layer_off[d] + t->node_slot(d, p)), 32);
  MH_CATCH
}

int mh_coset_lde_batch(mh_ctx* c, const uint64_t* rowmajor, int log_n, size_t width, int added_bits, uint64_t shift,
                       uint64_t* out) {
  MH_TRY(c)
  MH_REQUIRE(c && rowmajor && out && width > 0 && log_n >= 0 && added_bits >= 0, "bad argument");
  MH_REQUIRE(log_n + added_bits <= 32, "LDE order exceeds the field's two-adicity");
  HIP_CHECK(hipSetDevice(c->device));
  size_t N = (size_t)1 << log_n, total = (N << added_bits) * width;
  DevBuf staging(N * width * 8), cols(N * width * 8), scratch(N * width * 8), lde(total * 8), tmp(total * 8);
  HIP_CHECK(hipMemcpyAsync(staging.p, rowmajor, N * width * 8, hipMemcpyHostToDevice, c->stream));
  launch_transpose_rm_to_cm(c, staging.u(), cols.u(), N, width);
  u64 wk = gl_two_adic_generator(log_n + added_bits);
  std::vector<u64> shifts((size_t)1 << added_bits);
  u64 x = gl_canon(shift);
  for (auto& v : shifts) { v = x; x = gl_mul(x, wk); }
  lde_columns(c, cols.u(), width, log_n, 1, shifts, lde.u(), scratch.u());
  MH_LAUNCH(k_lde_to_reference_layout, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, lde.u(), tmp.u(),
                     log_n, added_bits, width);
  HIP_CHECK(hipMemcpyAsync(out, tmp.p, total * 8, hipMemcpyDeviceToHost, c->stream));
  c->sync();
  MH_CATCH
}

}  // extern "C"
