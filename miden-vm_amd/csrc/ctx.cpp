// Context plumbing: stream, profiler (HIP events on the private stream), sync.
#include "ctx.hpp"
#include <cstring>

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

size_t mh_ctx::prof_begin(const char* name, double bytes) {
  Pending p{name, get_event(), get_event(), bytes, false};
  HIP_CHECK(hipEventRecord(p.a, stream));
  pending.push_back(p);
  return pending.size() - 1;
}

void mh_ctx::prof_end(size_t slot) {
  HIP_CHECK(hipEventRecord(pending[slot].b, stream));
  pending[slot].closed = true;
}

void mh_ctx::prof_resolve() {
  if (pending.empty()) return;
  HIP_CHECK(hipStreamSynchronize(stream));
  for (auto& p : pending) {
    float ms = 0;
    if (!p.closed) {  // a scope left by an exception: nothing to measure
      event_pool.push_back(p.a);
      event_pool.push_back(p.b);
      continue;
    }
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

void mh_ctx::d2h(void* dst_host, const void* src_dev, size_t bytes) {
  if (!bytes) return;
  if (bytes > PINNED_BYTES) {
    HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, stream));
    sync();
    return;
  }
  if (!pinned) HIP_CHECK(hipHostMalloc(&pinned, PINNED_BYTES, hipHostMallocDefault));
  HIP_CHECK(hipMemcpyAsync(pinned, src_dev, bytes, hipMemcpyDeviceToHost, stream));
  sync();
  memcpy(dst_host, pinned, bytes);
}

void mh_ctx::h2d(void* dst_dev, const void* src_host, size_t bytes) {
  if (!bytes) return;
  if (bytes > RING_BYTES / 4) {  // big blocks: the runtime's own path (blocking for pageable sources)
    HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, stream));
    return;
  }
  if (!ring) HIP_CHECK(hipHostMalloc(&ring, RING_BYTES, hipHostMallocDefault));
  const size_t need = (bytes + 63) & ~(size_t)63;
  if (ring_pos + need > RING_BYTES) {  // wrap (every few hundred proofs): copies still in flight on any stream of this context may read the start of the ring
    sync();
    if (primary_stream && primary_stream != stream) HIP_CHECK(hipStreamSynchronize(primary_stream));  // `stream` may be swapped to the side stream
    if (side_stream) HIP_CHECK(hipStreamSynchronize(side_stream));
    if (copy_stream) HIP_CHECK(hipStreamSynchronize(copy_stream));
    ring_pos = 0;
  }
  void* slot = static_cast<char*>(ring) + ring_pos;
  ring_pos += need;
  memcpy(slot, src_host, bytes);
  HIP_CHECK(hipMemcpyAsync(dst_dev, slot, bytes, hipMemcpyHostToDevice, stream));
}

void* mh_ctx::host_take(size_t bytes) {
  for (size_t i = 0; i < host_pool.size(); i++)
    if (host_pool[i].second >= bytes && host_pool[i].second <= 2 * bytes + 4096) {
      void* p = host_pool[i].first;
      host_pool.erase(host_pool.begin() + i);
      return p;
    }
  void* p = nullptr;
  HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
  return p;
}
