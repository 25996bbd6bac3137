// The collectives of a sharded proof over RCCL / xGMI, inside the library.
//
// One process + one mh_ctx per GPU.  mh_comm_create_rccl builds an ncclComm for the ctx's device and returns an mh_comm
// whose three operations (include/midenhip.h) enqueue RCCL collectives on the ctx's OWN stream and on the library's own
// device buffers: no host round trip, no staging copy, no Python in the loop; whatever the prover launches next on
// that stream is ordered behind the collective.  The Rust caller needs nothing but the 128-byte id from rank 0
// (mh_rccl_unique_id) delivered to every rank by whatever it already uses to start its workers.
//
// RCCL is loaded lazily (dlopen): single-GPU users never touch it and the library has no link-time dependency on it.
// It is opened by PATH ($MH_RCCL_LIB, else $ROCM_PATH/lib/librccl.so.1, else /opt/rocm/lib/librccl.so.1): a process that
// also hosts PyTorch already holds PyTorch's private librccl.so, which is bound to PyTorch's private HIP runtime and
// cannot see this library's allocations; a bare soname could resolve to that copy.
#include "../../include/midenhip.h"
#include "ctx.hpp"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<std::string> paths;
    if (const char* p = getenv("MH_RCCL_LIB")) paths.push_back(p);
    if (const char* r = getenv("ROCM_PATH")) paths.push_back(std::string(r) + "/lib/librccl.so.1");
    paths.push_back("/opt/rocm/lib/librccl.so.1");
    for (auto& p : paths) {
      api.handle = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
      api.error += p + ": " + (dlerror() ? dlerror() : "?") + "; ";
    }
    if (!api.handle) return;
#define MH_SYM(field, name)                                                        \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name));      \
  if (!api.field) {                                                                \
    api.error = std::string("RCCL symbol missing: ") + name;                       \
    api.handle = nullptr;                                                          \
    return;                                                                        \
  }
    MH_SYM(GetUniqueId, "ncclGetUniqueId")
    MH_SYM(CommInitRank, "ncclCommInitRank")
    MH_SYM(CommDestroy, "ncclCommDestroy")
    MH_SYM(GroupStart, "ncclGroupStart")
    MH_SYM(GroupEnd, "ncclGroupEnd")
    MH_SYM(AllGather, "ncclAllGather")
    MH_SYM(AllReduce, "ncclAllReduce")
    MH_SYM(Send, "ncclSend")
    MH_SYM(Recv, "ncclRecv")
    MH_SYM(GetErrorString, "ncclGetErrorString")
#undef MH_SYM
  });
  return api;
}

// The object behind an mh_comm created here: the public struct first, so that the mh_comm* handed out is this object.
struct RcclComm {
  mh_comm pub;
  mh_ctx* ctx;
  ncclComm_t comm;
};

int fail(RcclComm* rc, const char* what, ncclResult_t r) {
  if (rc && rc->ctx) rc->ctx->err = std::string("RCCL ") + what + ": " + rccl().GetErrorString(r);
  return 1;
}

int rccl_all_to_all(void* user, const void* send, void* recv, size_t bytes_per_peer) {
  RcclComm* rc = static_cast<RcclComm*>(user);
  RcclApi& a = rccl();
  if (bytes_per_peer % 8) return fail(rc, "all_to_all: size not a multiple of 8", ncclInvalidArgument);
  const size_t n = bytes_per_peer / 8;
  ncclResult_t r = a.GroupStart();
  if (r != ncclSuccess) return fail(rc, "group start", r);
  for (int p = 0; p < rc->pub.world && r == ncclSuccess; p++) {
    r = a.Send(static_cast<const char*>(send) + (size_t)p * bytes_per_peer, n, ncclUint64, p, rc->comm, rc->ctx->stream);
    if (r == ncclSuccess) r = a.Recv(static_cast<char*>(recv) + (size_t)p * bytes_per_peer, n, ncclUint64, p, rc->comm, rc->ctx->stream);
  }
  const ncclResult_t e = a.GroupEnd();
  if (r != ncclSuccess) return fail(rc, "send/recv", r);
  if (e != ncclSuccess) return fail(rc, "group end", e);
  return 0;
}
int rccl_all_gather(void* user, const void* send, void* recv, size_t bytes_per_rank) {
  RcclComm* rc = static_cast<RcclComm*>(user);
  if (bytes_per_rank % 8) return fail(rc, "all_gather: size not a multiple of 8", ncclInvalidArgument);
  const ncclResult_t r = rccl().AllGather(send, recv, bytes_per_rank / 8, ncclUint64, rc->comm, rc->ctx->stream);
  return r == ncclSuccess ? 0 : fail(rc, "all_gather", r);
}
int rccl_all_reduce(void* user, uint64_t* buf, size_t n) {
  RcclComm* rc = static_cast<RcclComm*>(user);
  const ncclResult_t r = rccl().AllReduce(buf, buf, n, ncclUint64, ncclSum, rc->comm, rc->ctx->stream);
  return r == ncclSuccess ? 0 : fail(rc, "all_reduce", r);
}

}  // namespace

extern "C" {

int mh_rccl_unique_id(uint8_t id[MH_RCCL_ID_BYTES]) {
  static_assert(MH_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id) return MH_ERR_INVALID;
  RcclApi& a = rccl();
  if (!a.handle) return MH_ERR_INTERNAL;
  ncclUniqueId u;
  if (a.GetUniqueId(&u) != ncclSuccess) return MH_ERR_INTERNAL;
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return MH_OK;
}

int mh_comm_create_rccl(mh_ctx* c, const uint8_t id[MH_RCCL_ID_BYTES], int rank, int world, mh_comm** out) {
  if (!c || !id || !out) return MH_ERR_INVALID;
  try {
    MH_REQUIRE(world >= 1 && (world & (world - 1)) == 0 && rank >= 0 && rank < world, "world must be a power of two and 0 <= rank < world");
    RcclApi& a = rccl();
    if (!a.handle) throw MhError(MH_ERR_INTERNAL, "RCCL is not available: " + a.error);
    HIP_CHECK(hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    std::unique_ptr<RcclComm> rc(new RcclComm());
    rc->ctx = c;
    const ncclResult_t r = a.CommInitRank(&rc->comm, world, u, rank);
    if (r != ncclSuccess) throw MhError(MH_ERR_INTERNAL, std::string("ncclCommInitRank: ") + a.GetErrorString(r));
    rc->pub.rank = rank;
    rc->pub.world = world;
    rc->pub.user = rc.get();
    rc->pub.all_to_all = rccl_all_to_all;
    rc->pub.all_gather = rccl_all_gather;
    rc->pub.all_reduce_sum_u64 = rccl_all_reduce;
    rc->pub.stream_ordered = 1;
    *out = &rc.release()->pub;
    return MH_OK;
  } catch (const MhError& e) {
    c->err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    c->err = e.what();
    return MH_ERR_INTERNAL;
  }
}

extern "C" void mh_comm_destroy_local(mh_comm* comm);  // comm_local.cpp (a no-op for other communicators)
void mh_comm_destroy(mh_comm* comm) {
  if (!comm) return;
  if (comm->all_to_all != rccl_all_to_all) {  // only communicators made by this library
    mh_comm_destroy_local(comm);
    return;
  }
  RcclComm* rc = static_cast<RcclComm*>(comm->user);
  (void)hipSetDevice(rc->ctx->device);
  (void)hipStreamSynchronize(rc->ctx->stream);
  (void)rccl().CommDestroy(rc->comm);
  delete rc;
}

// Runs the three collectives once on small device buffers with known contents and checks what comes back
// (every rank calls it; works for world = 1 as well, where RCCL moves the data to itself).
int mh_comm_selftest(mh_ctx* c, const mh_comm* comm) {
  if (!c || !comm || !comm->all_to_all || !comm->all_gather || !comm->all_reduce_sum_u64) return MH_ERR_INVALID;
  PoolScope ps(c);
  try {
    HIP_CHECK(hipSetDevice(c->device));
    const int W = comm->world, R = comm->rank;
    const size_t per = 64;  // u64 per peer
    std::vector<u64> h(per * W), back(per * W);
    DevBuf a(per * W * 8), b(per * W * 8);
    auto finish = [&]() {
      if (!comm->stream_ordered) return;  // host-synchronous communicators complete before returning
    };
    // all_to_all: block p of rank r carries (r, p, k); afterwards block p must carry (p, r, k)
    for (int p = 0; p < W; p++)
      for (size_t k = 0; k < per; k++) h[p * per + k] = ((u64)R << 40) | ((u64)p << 20) | k;
    HIP_CHECK(hipMemcpyAsync(a.p, h.data(), h.size() * 8, hipMemcpyHostToDevice, c->stream));
    if (!comm->stream_ordered) c->sync();
    MH_REQUIRE(comm->all_to_all(comm->user, a.p, b.p, per * 8) == 0, "selftest: all_to_all failed: " + c->err);
    finish();
    c->d2h(back.data(), b.p, back.size() * 8);
    for (int p = 0; p < W; p++)
      for (size_t k = 0; k < per; k++)
        MH_REQUIRE(back[p * per + k] == (((u64)p << 40) | ((u64)R << 20) | k), "selftest: all_to_all delivered wrong data");
    // all_gather: rank r contributes (r, k)
    for (size_t k = 0; k < per; k++) h[k] = ((u64)R << 32) | k;
    HIP_CHECK(hipMemcpyAsync(a.p, h.data(), per * 8, hipMemcpyHostToDevice, c->stream));
    if (!comm->stream_ordered) c->sync();
    MH_REQUIRE(comm->all_gather(comm->user, a.p, b.p, per * 8) == 0, "selftest: all_gather failed: " + c->err);
    c->d2h(back.data(), b.p, back.size() * 8);
    for (int p = 0; p < W; p++)
      for (size_t k = 0; k < per; k++) MH_REQUIRE(back[p * per + k] == (((u64)p << 32) | k), "selftest: all_gather delivered wrong data");
    // all_reduce: slot k is owned by rank k mod W
    for (size_t k = 0; k < per * W; k++) h[k] = (int)(k % W) == R ? 0x9E3779B97F4A7C15ULL * (k + 1) : 0;
    HIP_CHECK(hipMemcpyAsync(a.p, h.data(), h.size() * 8, hipMemcpyHostToDevice, c->stream));
    if (!comm->stream_ordered) c->sync();
    MH_REQUIRE(comm->all_reduce_sum_u64(comm->user, a.u(), per * W) == 0, "selftest: all_reduce failed: " + c->err);
    c->d2h(back.data(), a.p, back.size() * 8);
    for (size_t k = 0; k < per * W; k++) MH_REQUIRE(back[k] == 0x9E3779B97F4A7C15ULL * (k + 1), "selftest: all_reduce delivered wrong data");
    return MH_OK;
  } catch (const MhError& e) {
    c->err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    c->err = e.what();
    return MH_ERR_INTERNAL;
  }
}

}  // extern "C"
