// tdr_comm.cpp — the exchange steps of a particle filter sharded over several GPUs (include/tdr.h, "tdr_comm_*").
//
// One process per GPU, particles partitioned contiguously by rank.  A sharded tdr_filter (tdr_host.cpp) needs
// exactly two collectives per update — the all-gather of {raw weight, last_dist} of every shard and the all-gather of
// the state planes the resampler reads — plus the broadcast of the rasterised scan from rank 0.  This file provides them
// behind one small interface with two transports:
//   * RCCL over xGMI, called directly (ncclAllGather / ncclBroadcast on the filter's HIP stream).  librccl.so is loaded
//     on first use (dlopen): single-GPU users neither link nor load it.
//   * caller-supplied functions (tdr_comm_create): for hosts that already own a transport (MPI, a ROS multi-process
//     launch, a test double) — the filter's logic is identical on top of either.
// The reference has no counterpart (it is a single-process CPU program); the partition is the north star's.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "tdr.h"

extern "C" int tdr_set_error(int code, const char* msg);

namespace {
int failc(int code, const char* fmt, const char* a = "", const char* b = "") {
  char buf[400];
  snprintf(buf, sizeof(buf), fmt, a, b);
  return tdr_set_error(code, buf);
}

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;
const char* g_rccl_err = nullptr;

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) { g_rccl_err = "librccl.so not found"; return; }
#define TDR_SYM(field, name)                                              \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, name)); \
  if (!g_rccl.field) { g_rccl_err = "librccl.so lacks " name; return; }
  TDR_SYM(GetUniqueId, "ncclGetUniqueId")
  TDR_SYM(CommInitRank, "ncclCommInitRank")
  TDR_SYM(CommDestroy, "ncclCommDestroy")
  TDR_SYM(AllGather, "ncclAllGather")
  TDR_SYM(Broadcast, "ncclBroadcast")
  TDR_SYM(GetErrorString, "ncclGetErrorString")
#undef TDR_SYM
}
int need_rccl() {
  std::call_once(g_rccl_once, load_rccl);
  if (g_rccl_err) return failc(TDR_ERR_HIP, "RCCL: %s", g_rccl_err);
  return TDR_OK;
}
}  // namespace

struct tdr_comm {
  int world = 1, rank = 0;
  tdr_comm_ops ops{};          // the transport in use
  ncclComm_t nccl = nullptr;   // RCCL transport only
};

namespace {
int rccl_all_gather(void* ctx, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream) {
  tdr_comm* c = static_cast<tdr_comm*>(ctx);
  const ncclResult_t r = g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, ncclChar, c->nccl, (hipStream_t)stream);
  return r == ncclSuccess ? TDR_OK : failc(TDR_ERR_HIP, "ncclAllGather: %s", g_rccl.GetErrorString(r));
}
int rccl_broadcast(void* ctx, void* buf_dev, size_t bytes, int root, void* stream) {
  tdr_comm* c = static_cast<tdr_comm*>(ctx);
  const ncclResult_t r = g_rccl.Broadcast(buf_dev, buf_dev, bytes, ncclChar, root, c->nccl, (hipStream_t)stream);
  return r == ncclSuccess ? TDR_OK : failc(TDR_ERR_HIP, "ncclBroadcast: %s", g_rccl.GetErrorString(r));
}
}  // namespace

extern "C" {

int tdr_comm_rccl_unique_id(void* id_out128) {
  if (!id_out128) return failc(TDR_ERR_ARG, "comm_rccl_unique_id: null output");
  if (int rc = need_rccl()) return rc;
  static_assert(sizeof(ncclUniqueId) == TDR_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return failc(TDR_ERR_HIP, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
  std::memcpy(id_out128, &id, sizeof(id));
  return TDR_OK;
}

int tdr_comm_create_rccl(int world_size, int rank, const void* unique_id128, tdr_comm** out) {
  if (!out || !unique_id128 || world_size < 1 || rank < 0 || rank >= world_size)
    return failc(TDR_ERR_ARG, "comm_create_rccl: bad arguments");
  if (int rc = need_rccl()) return rc;
  tdr_comm* c = new tdr_comm();
  c->world = world_size;
  c->rank = rank;
  ncclUniqueId id;
  std::memcpy(&id, unique_id128, sizeof(id));
  const ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world_size, id, rank);   // binds the calling thread's current HIP device
  if (r != ncclSuccess) {
    delete c;
    return failc(TDR_ERR_HIP, "ncclCommInitRank: %s", g_rccl.GetErrorString(r));
  }
  c->ops.ctx = c;
  c->ops.all_gather = rccl_all_gather;
  c->ops.broadcast = rccl_broadcast;
  *out = c;
  return TDR_OK;
}

int tdr_comm_create(int world_size, int rank, const tdr_comm_ops* ops, tdr_comm** out) {
  if (!out || !ops || !ops->all_gather || !ops->broadcast || world_size < 1 || rank < 0 || rank >= world_size)
    return failc(TDR_ERR_ARG, "comm_create: bad arguments");
  tdr_comm* c = new tdr_comm();
  c->world = world_size;
  c->rank = rank;
  c->ops = *ops;
  *out = c;
  return TDR_OK;
}

void tdr_comm_destroy(tdr_comm* c) {
  if (!c) return;
  if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
  delete c;
}

int tdr_comm_world(const tdr_comm* c) { return c ? c->world : 1; }
int tdr_comm_rank(const tdr_comm* c) { return c ? c->rank : 0; }

int tdr_comm_all_gather(tdr_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream) {
  if (!c || !send_dev || !recv_dev) return failc(TDR_ERR_ARG, "comm_all_gather: null pointer");
  if (bytes_per_rank == 0) return TDR_OK;
  return c->ops.all_gather(c->ops.ctx, send_dev, recv_dev, bytes_per_rank, stream);
}
int tdr_comm_broadcast(tdr_comm* c, void* buf_dev, size_t bytes, int root, void* stream) {
  if (!c || !buf_dev || root < 0 || root >= c->world) return failc(TDR_ERR_ARG, "comm_broadcast: bad arguments");
  if (bytes == 0) return TDR_OK;
  return c->ops.broadcast(c->ops.ctx, buf_dev, bytes, root, stream);
}

}  // extern "C"
