// Runtime context of the MI355X proving backend: one HIP device, one private stream, cached
// twiddle / coset tables, a kernel profiler (HIP events on the private stream) and error state.
// One ctx per proving thread (SURVEY.md section 8b "Threading"): no global mutable state.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

typedef uint64_t u64;

struct MhError : std::runtime_error {
  int code;
  MhError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define MH_ERR_INVALID 1
#define MH_ERR_HIP 2
#define MH_ERR_OOM 3
#define MH_ERR_INTERNAL 4
#define MH_ERR_COMM 5

#define HIP_CHECK(expr)                                                                          \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      throw MhError(_e == hipErrorOutOfMemory ? MH_ERR_OOM : MH_ERR_HIP,                         \
                    std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " + __FILE__ + ":" + \
                        std::to_string(__LINE__));                                               \
  } while (0)

// Every kernel launch goes through this: a bad launch configuration (grid of 0, too much LDS, an unset attribute) is reported
// HERE with file and line, not at the next synchronisation under somebody else's name.
#define MH_LAUNCH(...)                                                                                             \
  do {                                                                                                             \
    hipLaunchKernelGGL(__VA_ARGS__);                                                                               \
    hipError_t _le = hipGetLastError();                                                                            \
    if (_le != hipSuccess)                                                                                         \
      throw MhError(MH_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString(_le) + " at " + __FILE__ + ":" + \
                                    std::to_string(__LINE__));                                                     \
  } while (0)

#define MH_REQUIRE(cond, msg)                                        \
  do {                                                               \
    if (!(cond)) throw MhError(MH_ERR_INVALID, std::string(msg));    \
  } while (0)

// Per-context cache of device allocations.  A proof allocates the same few dozen buffer sizes every
// time; hipMalloc/hipFree cost milliseconds (hipFree also synchronises the device), so freed
// buffers are kept by size and handed back to the next request.  All work of a ctx is ordered on
// its one stream, so reuse is stream-ordered and safe.  mh_ctx_destroy / mh_ctx_trim release it.
struct DevPool {
  std::multimap<size_t, void*> free_list;
  size_t cached_bytes = 0;
  void* take(size_t n) {
    auto it = free_list.find(n);
    if (it == free_list.end()) return nullptr;
    void* p = it->second;
    free_list.erase(it);
    cached_bytes -= n;
    return p;
  }
  void give(void* p, size_t n) {
    free_list.emplace(n, p);
    cached_bytes += n;
  }
  void trim() {
    for (auto& kv : free_list) (void)hipFree(kv.second);
    free_list.clear();
    cached_bytes = 0;
  }
};
// The pool of the ctx whose API call is running on this thread (set by the C-ABI entry points).
extern thread_local DevPool* g_dev_pool;

// RAII device allocation.
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevPool* pool = nullptr;
  DevBuf() {}
  explicit DevBuf(size_t n) { alloc(n); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), pool(o.pool) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p; bytes = o.bytes; pool = o.pool; o.p = nullptr; o.bytes = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t n) {
    release();
    if (n == 0) return;
    pool = g_dev_pool;
    if (pool && (p = pool->take(n))) {
      bytes = n;
      return;
    }
    hipError_t e = hipMalloc(&p, n);
    if (e == hipErrorOutOfMemory && pool && pool->cached_bytes) {  // give cached buffers back and retry
      (void)hipGetLastError();
      pool->trim();
      e = hipMalloc(&p, n);
    }
    if (e != hipSuccess) {
      p = nullptr;
      HIP_CHECK(e);
    }
    bytes = n;
  }
  void release() {
    if (p) {
      if (pool) pool->give(p, bytes);
      else (void)hipFree(p);
    }
    p = nullptr; bytes = 0;
  }
  u64* u() const { return (u64*)p; }
};

struct ProfEntry {
  double ms = 0;
  double bytes = 0;  // algorithmic bytes attributed by the caller
  long count = 0;
};

struct mh_ctx {
  int device = 0;
  DevPool pool;
  hipStream_t stream = nullptr;
  hipStream_t primary_stream = nullptr;  // = stream at creation; `stream` is swapped to the side stream inside commit_traces_pipelined
  hipStream_t side_stream = nullptr;  // commit_traces: forward NTTs of the next coset group under the leaf sponges of the previous one (created on first use)
  hipStream_t copy_stream = nullptr;  // mh_trace_upload_async: DMA copies + transposes that run under the proof's kernels (created on first use)
  std::string err;
  // profiler
  bool prof_on = false;
  std::string prof_only;  // mh_prof_filter: the one scope name that is recorded ("" = all)
  struct Pending { std::string name; hipEvent_t a, b; double bytes; bool closed; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
  std::map<std::string, ProfEntry> prof;
  // caches: log_n -> device table of w_N^k (k < N/2) forward / inverse
  std::map<int, DevBuf> tw_fwd, tw_inv;
  // coset-scale tables keyed by (log_n, log_blowup, kind)
  std::map<std::string, DevBuf> tables;
  std::map<std::string, std::vector<size_t>> table_index;  // host-side offsets into `tables` entries
  int lmcs = 0;  // MH_LMCS_POSEIDON2 / MH_LMCS_BLAKE3: the commitment scheme's hasher (StarkConfig::Lmcs), mh_ctx_set_lmcs
  bool ntt_big_lds_attr = false;  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) done for this ctx's device

  hipEvent_t get_event();
  size_t prof_begin(const char* name, double bytes);  // returns the scope's slot: scopes may nest
  void prof_end(size_t slot);
  void prof_resolve();
  void sync();
  // Blocking device-to-host copy of a small result (roots, opened rows, partial sums) through a page-locked bounce
  // buffer: a straight DMA instead of the runtime's staged copy into pageable memory (~40 us less per call).
  void d2h(void* dst_host, const void* src_dev, size_t bytes);
  // Asynchronous host-to-device copy of a small parameter block (challenge powers, OOD points, program constants) through a
  // page-locked ring: from pageable memory the runtime stages the bytes itself and the GPU idles ~25 us per copy (kernel trace: ~15
  // such copies per proof).  The source may be reused as soon as the call returns.
  void h2d(void* dst_dev, const void* src_host, size_t bytes);
  void* ring = nullptr;
  size_t ring_pos = 0;
  static constexpr size_t RING_BYTES = 4 << 20;
  void* pinned = nullptr;
  void* pinned_top = nullptr;  // lmcs_compress_layers: the tree top computed on the host, on its way back to the device
  static constexpr size_t PINNED_BYTES = 1 << 20;
  // page-locked scratch for host-built aux traces (the aux_builder callback writes into it, the DMA reads it): kept between proofs
  std::vector<std::pair<void*, size_t>> host_pool;
  void* host_take(size_t bytes);
  void host_give(void* p, size_t bytes) { host_pool.emplace_back(p, bytes); }
  const u64* twiddles(int log_n, bool inverse);
};

// RAII: make `c`'s allocation pool current for the duration of one C-ABI call.
struct PoolScope {
  DevPool* prev;
  explicit PoolScope(mh_ctx* c) : prev(g_dev_pool) {
    if (c) g_dev_pool = &c->pool;
  }
  ~PoolScope() { g_dev_pool = prev; }
};

// RAII helper: time everything launched on ctx->stream within the scope under `name`.
struct ProfScope {
  mh_ctx* c;
  size_t slot = (size_t)-1;
  ProfScope(mh_ctx* ctx, const char* name, double bytes = 0) : c(ctx) {
    if (c->prof_on && (c->prof_only.empty() || c->prof_only == name)) slot = c->prof_begin(name, bytes);
  }
  ~ProfScope() {
    if (slot != (size_t)-1 && slot < c->pending.size()) c->prof_end(slot);
  }
};
