// The collectives of a sharded proof between the contexts of ONE process: one thread + one mh_ctx per rank, the ranks on
// different GPUs of the node (peer copies over xGMI) or -- in the tests -- several ranks sharing one GPU.
//
// Like the RCCL communicator (comm_rccl.cpp) it is stream ordered: a collective is enqueued on the caller's own stream and
// returns at once.  Rank r records an event when its send buffer is complete, all ranks meet at a host barrier to exchange
// pointers, every rank enqueues "wait for peer p's event, copy my block out of p's buffer" on its own stream, and a second
// event per rank tells the peers when their buffers have been read (their streams wait for it before running on).
// A Rust caller that keeps all GPUs of a node in one process needs nothing else: no RCCL, no launcher, no id to hand around.
#include "../../include/midenhip.h"
#include "ctx.hpp"
#include "gl.cuh"
#include <condition_variable>
#include <mutex>
#include <vector>

struct mh_local_fabric {
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long generation = 0;
  bool failed = false;
  std::vector<mh_ctx*> ctx;
  std::vector<const void*> send;
  std::vector<void*> recv;
  std::vector<hipEvent_t> ready, done;
  std::vector<int> joined;
  // false = the fabric has been aborted (a rank failed or left): nobody waits for that rank any more
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (failed) return false;
    const unsigned long g = generation;
    if (++arrived == world) {
      arrived = 0;
      generation++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != g || failed; });
    }
    return !failed;
  }
  void abort() {  // sticky: a fabric that lost a rank is dead; waiting ranks wake up and report an error
    {
      std::lock_guard<std::mutex> g(mu);
      failed = true;
    }
    cv.notify_all();
  }
};

namespace {

struct LocalComm {
  mh_comm pub;
  mh_local_fabric* f;
  mh_ctx* ctx;
  DevBuf tmp;  // all-reduce staging: world * n
};

__global__ void k_sum_ranks(const u64* __restrict__ parts, u64* __restrict__ out, size_t n, int world) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 s = 0;
  for (int p = 0; p < world; p++) s += parts[(size_t)p * n + i];  // every slot is contributed by exactly one rank
  out[i] = s;
}

void copy_from_peer(LocalComm* lc, int p, void* dst, const void* src, size_t bytes) {
  mh_local_fabric* f = lc->f;
  HIP_CHECK(hipStreamWaitEvent(lc->ctx->stream, f->ready[p], 0));
  if (f->ctx[p]->device == lc->ctx->device) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, lc->ctx->stream));
  else HIP_CHECK(hipMemcpyPeerAsync(dst, lc->ctx->device, src, f->ctx[p]->device, bytes, lc->ctx->stream));
}

int fail_aborted(LocalComm* lc) {
  if (lc->ctx->err.empty()) lc->ctx->err = "local fabric aborted: another rank failed or left the collective";
  return 1;
}

// common frame: publish, barrier, body (enqueue copies), done event, barrier, wait for the peers' done events
template <class Body>
int collective(LocalComm* lc, const void* send, void* recv, Body body) {
  mh_local_fabric* f = lc->f;
  const int r = lc->pub.rank;
  int rc = 0;
  try {
    HIP_CHECK(hipSetDevice(lc->ctx->device));
    HIP_CHECK(hipEventRecord(f->ready[r], lc->ctx->stream));
  } catch (const std::exception& e) {
    lc->ctx->err = e.what();
    rc = 1;
  }
  f->send[r] = send;
  f->recv[r] = recv;
  if (rc) f->abort();
  if (!f->barrier()) return fail_aborted(lc);
  try {
    body();
    HIP_CHECK(hipEventRecord(f->done[r], lc->ctx->stream));
  } catch (const std::exception& e) {
    lc->ctx->err = e.what();
    f->abort();
  }
  if (!f->barrier()) return fail_aborted(lc);
  try {
    for (int p = 0; p < f->world; p++)
      if (p != r) HIP_CHECK(hipStreamWaitEvent(lc->ctx->stream, f->done[p], 0));  // my buffers have been read
  } catch (const std::exception& e) {
    lc->ctx->err = e.what();
    f->abort();
    return 1;
  }
  return 0;
}

int local_all_to_all(void* user, const void* send, void* recv, size_t bytes_per_peer) {
  LocalComm* lc = static_cast<LocalComm*>(user);
  return collective(lc, send, recv, [&] {
    const int r = lc->pub.rank;
    for (int p = 0; p < lc->f->world; p++)
      copy_from_peer(lc, p, static_cast<char*>(recv) + (size_t)p * bytes_per_peer,
                     static_cast<const char*>(lc->f->send[p]) + (size_t)r * bytes_per_peer, bytes_per_peer);
  });
}
int local_all_gather(void* user, const void* send, void* recv, size_t bytes_per_rank) {
  LocalComm* lc = static_cast<LocalComm*>(user);
  return collective(lc, send, recv, [&] {
    for (int p = 0; p < lc->f->world; p++)
      copy_from_peer(lc, p, static_cast<char*>(recv) + (size_t)p * bytes_per_rank, lc->f->send[p], bytes_per_rank);
  });
}
int local_all_reduce(void* user, uint64_t* buf, size_t n) {
  LocalComm* lc = static_cast<LocalComm*>(user);
  const int W = lc->f->world;
  try {
    if (lc->tmp.bytes < (size_t)W * n * 8) {
      PoolScope ps(lc->ctx);
      lc->tmp.alloc((size_t)W * n * 8);
    }
  } catch (const std::exception& e) {
    lc->ctx->err = e.what();
    lc->f->abort();
  }
  // gather every rank's vector into the staging area; the sum overwrites `buf` only after all peers have read it
  const int rc = collective(lc, buf, buf, [&] {
    for (int p = 0; p < W; p++) copy_from_peer(lc, p, lc->tmp.u() + (size_t)p * n, lc->f->send[p], n * 8);
  });
  if (rc) return rc;
  try {
    MH_LAUNCH(k_sum_ranks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lc->ctx->stream, lc->tmp.u(), buf, n, W);
  } catch (const std::exception& e) {
    lc->ctx->err = e.what();
    return 1;
  }
  return 0;
}

}  // namespace

extern "C" {

mh_local_fabric* mh_local_fabric_create(int world) {
  if (world < 1 || (world & (world - 1))) return nullptr;
  mh_local_fabric* f = new mh_local_fabric();
  f->world = world;
  f->ctx.assign(world, nullptr);
  f->send.assign(world, nullptr);
  f->recv.assign(world, nullptr);
  f->ready.assign(world, nullptr);
  f->done.assign(world, nullptr);
  f->joined.assign(world, 0);
  return f;
}
void mh_local_fabric_destroy(mh_local_fabric* f) { delete f; }
void mh_local_fabric_abort(mh_local_fabric* f) {
  if (f) f->abort();
}

int mh_comm_create_local(mh_ctx* c, mh_local_fabric* f, int rank, mh_comm** out) {
  if (!c || !f || !out || rank < 0 || rank >= f->world) return MH_ERR_INVALID;
  try {
    HIP_CHECK(hipSetDevice(c->device));
    {
      std::lock_guard<std::mutex> g(f->mu);
      MH_REQUIRE(!f->joined[rank], "rank already joined this fabric");
      f->joined[rank] = 1;
      f->ctx[rank] = c;
      HIP_CHECK(hipEventCreateWithFlags(&f->ready[rank], hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&f->done[rank], hipEventDisableTiming));
    }
    // every rank has joined: peers' contexts and events are known from here on
    MH_REQUIRE(f->barrier(), "local fabric aborted while ranks were joining");
    for (int p = 0; p < f->world; p++)
      if (f->ctx[p]->device != c->device) {
        int can = 0;
        HIP_CHECK(hipDeviceCanAccessPeer(&can, c->device, f->ctx[p]->device));
        if (can) {
          hipError_t e = hipDeviceEnablePeerAccess(f->ctx[p]->device, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e);
          (void)hipGetLastError();
        }
      }
    LocalComm* lc = new LocalComm();
    lc->f = f;
    lc->ctx = c;
    lc->pub.rank = rank;
    lc->pub.world = f->world;
    lc->pub.user = lc;
    lc->pub.all_to_all = local_all_to_all;
    lc->pub.all_gather = local_all_gather;
    lc->pub.all_reduce_sum_u64 = local_all_reduce;
    lc->pub.stream_ordered = 1;
    *out = &lc->pub;
    return MH_OK;
  } catch (const MhError& e) {
    c->err = e.what();
    f->abort();  // the other ranks must not wait for this one
    return e.code;
  } catch (const std::exception& e) {
    c->err = e.what();
    f->abort();
    return MH_ERR_INTERNAL;
  }
}

void mh_comm_destroy_local(mh_comm* comm) {
  if (!comm || comm->all_to_all != local_all_to_all) return;
  LocalComm* lc = static_cast<LocalComm*>(comm->user);
  (void)hipSetDevice(lc->ctx->device);
  (void)hipStreamSynchronize(lc->ctx->stream);
  {
    PoolScope ps(lc->ctx);
    lc->tmp.release();
  }
  delete lc;
}

}  // extern "C"
