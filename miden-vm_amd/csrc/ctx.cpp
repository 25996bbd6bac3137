// Context plumbing: stream, profiler (HIP events on the private stream), sync.
#include "ctx.hpp"

thread_local DevPool* g_dev_pool = nullptr;

hipEvent_t mh_ctx::get_event() {
  if (!event_pool.empty()) {
    hipEvent_t e = event_pool.back();
    event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  return e;
}

void mh_ctx::prof_begin(const char* name, double bytes) {
  Pending p{name, get_event(), get_event(), bytes};
  HIP_CHECK(hipEventRecord(p.a, stream));
  pending.push_back(p);
}

void mh_ctx::prof_end() {
  // close the most recent open scope (scopes do not nest across kernels in practice)
  HIP_CHECK(hipEventRecord(pending.back().b, stream));
}

void mh_ctx::prof_resolve() {
  if (pending.empty()) return;
  HIP_CHECK(hipStreamSynchronize(stream));
  for (auto& p : pending) {
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
    auto& e = prof[p.name];
    e.ms += ms;
    e.bytes += p.bytes;
    e.count += 1;
    event_pool.push_back(p.a);
    event_pool.push_back(p.b);
  }
  pending.clear();
}

void mh_ctx::sync() { HIP_CHECK(hipStreamSynchronize(stream)); }
