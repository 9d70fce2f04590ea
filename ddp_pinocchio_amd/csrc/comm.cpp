// comm.cpp -- the one exchange step of the multi-GPU shard (no reference counterpart).
// Independent instances are sharded over ranks (instance s -> rank s mod G, local position s div G) with no data-path
// collective; after a sweep the ranks agree on the best instance:
//   ddp_hip_shard_pick       local argmin on the device (pick.hip) -> ONE 16-byte ncclAllGather of {cost, global index} over
//                            xGMI -> argmin of the G pairs on the device -> one 16-byte read-back, one stream sync
//   ddp_hip_shard_broadcast  optional: the winner's trajectory and gains (X, U, FB_*: ~4.9 MB at the Talos shape) from its
//                            owner to every rank, one grouped ncclBroadcast straight out of / into the resident sequences
//   ddp_hip_shard_best       round 1's host-scalar form (two 8-byte all-reduces: RCCL has no MINLOC), kept for callers
//                            that hold their costs on the host
// Latency-bound (~10 us per collective); bandwidth only matters for the broadcast (one direct xGMI link per peer).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include <new>

#include "ddp_hip/ddp_hip.h"
#include "internal.h"

struct ddp_hip_comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int device = 0, rank = 0, nranks = 1;
  double* d_cost = nullptr;
  int64_t* d_idx = nullptr;
  double* d_pair = nullptr;     // {cost, global index}: local best, then the global best
  double* d_gather = nullptr;   // [nranks] pairs
};

static_assert(sizeof(ncclUniqueId) <= DDP_HIP_COMM_ID_BYTES, "unique id must fit the C-ABI buffer");

extern "C" int ddp_hip_comm_unique_id(unsigned char id[DDP_HIP_COMM_ID_BYTES]) {
  if (!id) return DDP_HIP_E_ARG;
  ncclUniqueId uid;
  if (ncclGetUniqueId(&uid) != ncclSuccess) return DDP_HIP_E_COMM;
  memset(id, 0, DDP_HIP_COMM_ID_BYTES);
  memcpy(id, &uid, sizeof(uid));
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_comm_init(const unsigned char id[DDP_HIP_COMM_ID_BYTES], int rank, int nranks, int device,
                                 ddp_hip_comm** out) {
  if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return DDP_HIP_E_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return DDP_HIP_E_NODEVICE; }
  if (device < 0 || device >= ndev) return DDP_HIP_E_ARG;
  if (hipSetDevice(device) != hipSuccess) return DDP_HIP_E_HIP;
  ddp_hip_comm* c = new (std::nothrow) ddp_hip_comm();
  if (!c) return DDP_HIP_E_HIP;
  c->device = device; c->rank = rank; c->nranks = nranks;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(&c->d_cost, sizeof(double)) != hipSuccess || hipMalloc(&c->d_idx, sizeof(int64_t)) != hipSuccess ||
      hipMalloc(&c->d_pair, 2 * sizeof(double)) != hipSuccess || hipMalloc(&c->d_gather, 2 * sizeof(double) * (size_t)nranks) != hipSuccess) {
    ddp_hip_comm_destroy(c);
    return DDP_HIP_E_HIP;
  }
  if (ncclCommInitRank(&c->comm, nranks, uid, rank) != ncclSuccess) {
    ddp_hip_comm_destroy(c);
    return DDP_HIP_E_COMM;
  }
  *out = c;
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_comm_destroy(ddp_hip_comm* c) {
  if (!c) return DDP_HIP_E_ARG;
  (void)hipSetDevice(c->device);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->d_cost) (void)hipFree(c->d_cost);
  if (c->d_idx) (void)hipFree(c->d_idx);
  if (c->d_pair) (void)hipFree(c->d_pair);
  if (c->d_gather) (void)hipFree(c->d_gather);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return DDP_HIP_OK;
}

extern "C" int ddp_hip_shard_best(ddp_hip_comm* c, double local_cost, int64_t local_global_index, double* best_cost,
                                  int64_t* best_global_index) {
  if (!c || !best_cost || !best_global_index) return DDP_HIP_E_ARG;
  if (hipSetDevice(c->device) != hipSuccess) return DDP_HIP_E_HIP;
  double cost = local_cost;
  if (hipMemcpyAsync(c->d_cost, &cost, sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) return DDP_HIP_E_HIP;
  if (ncclAllReduce(c->d_cost, c->d_cost, 1, ncclDouble, ncclMin, c->comm, c->stream) != ncclSuccess) return DDP_HIP_E_COMM;
  if (hipMemcpyAsync(&cost, c->d_cost, sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return DDP_HIP_E_HIP;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return DDP_HIP_E_HIP;
  int64_t idx = (local_cost == cost) ? local_global_index : INT64_MAX;
  if (hipMemcpyAsync(c->d_idx, &idx, sizeof(int64_t), hipMemcpyHostToDevice, c->stream) != hipSuccess) return DDP_HIP_E_HIP;
  if (ncclAllReduce(c->d_idx, c->d_idx, 1, ncclInt64, ncclMin, c->comm, c->stream) != ncclSuccess) return DDP_HIP_E_COMM;
  if (hipMemcpyAsync(&idx, c->d_idx, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return DDP_HIP_E_HIP;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return DDP_HIP_E_HIP;
  *best_cost = cost;
  *best_global_index = idx;
  return DDP_HIP_OK;
}

// The cost of the trajectory forward_pass just produced, per instance: sum_t COSTS_OLD + the accepted step's cost
// difference -- both resident after ddp_hip_forward (which has synchronised the context's stream).  comm == NULL: one rank
// (the same device work without the collective: what a single-GPU run does, so that N = 1 and N > 1 lines are comparable).
extern "C" int ddp_hip_shard_pick(ddp_hip_comm* c, ddp_hip_ctx* ctx, double* best_cost, int64_t* best_global_index) {
  if (!ctx || !best_cost || !best_global_index) return DDP_HIP_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return DDP_HIP_E_HIP;
  double pair[2];
  if (!c) {
    if (!ctx->pick_pair_d && hipMalloc(&ctx->pick_pair_d, 2 * sizeof(double)) != hipSuccess) return DDP_HIP_E_HIP;
    int rc = pick_local_launch(ctx, 0, 1, ctx->pick_pair_d, ctx->stream);
    if (rc != DDP_HIP_OK) return rc;
    if (hipMemcpyAsync(pair, ctx->pick_pair_d, sizeof(pair), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return DDP_HIP_E_HIP;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return DDP_HIP_E_HIP;
  } else {
    if (c->device != ctx->device) return DDP_HIP_E_ARG;
    int rc = pick_local_launch(ctx, c->rank, c->nranks, c->d_pair, c->stream);
    if (rc != DDP_HIP_OK) return rc;
    if (ncclAllGather(c->d_pair, c->d_gather, 2, ncclDouble, c->comm, c->stream) != ncclSuccess) return DDP_HIP_E_COMM;
    rc = pick_final_launch(c->d_gather, c->nranks, c->d_pair, c->stream);
    if (rc != DDP_HIP_OK) return rc;
    if (hipMemcpyAsync(pair, c->d_pair, sizeof(pair), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return DDP_HIP_E_HIP;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return DDP_HIP_E_HIP;
  }
  *best_cost = pair[0];
  memcpy(best_global_index, &pair[1], sizeof(int64_t));
  return DDP_HIP_OK;
}

// The winner's trajectory and gains to every rank (kernel map C2): global instance s lives on rank
// s mod G at local position s div G; every rank receives it into its local instance `dst_local`.
extern "C" int ddp_hip_shard_broadcast(ddp_hip_comm* c, ddp_hip_ctx* ctx, int64_t best_global_index, int64_t dst_local) {
  if (!ctx || best_global_index < 0 || dst_local < 0 || dst_local >= ctx->d.batch) return DDP_HIP_E_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return DDP_HIP_E_HIP;
  const int G = c ? c->nranks : 1, me = c ? c->rank : 0;
  const int root = (int)(best_global_index % G);
  const int64_t src_local = best_global_index / G;
  if (me == root && src_local >= ctx->d.batch) return DDP_HIP_E_ARG;
  static const int seqs[5] = {DDP_HIP_SEQ_X, DDP_HIP_SEQ_U, DDP_HIP_SEQ_FB_ORIGIN, DDP_HIP_SEQ_FB_VAL, DDP_HIP_SEQ_FB_JAC};
  hipStream_t st = c ? c->stream : ctx->stream;
  if (c && ncclGroupStart() != ncclSuccess) return DDP_HIP_E_COMM;
  for (int k = 0; k < 5; ++k) {
    const SeqBuf& sb = ctx->seq[seqs[k]];
    if (!sb.ptr || sb.size == 0) continue;
    double* dst = sb.ptr + dst_local * sb.size;
    const double* src = me == root ? sb.ptr + src_local * sb.size : dst;
    if (c) {
      if (ncclBroadcast(src, dst, (size_t)sb.size, ncclDouble, root, c->comm, st) != ncclSuccess) { (void)ncclGroupEnd(); return DDP_HIP_E_COMM; }
    } else if (src != dst) {
      if (hipMemcpyAsync(dst, src, sizeof(double) * (size_t)sb.size, hipMemcpyDeviceToDevice, st) != hipSuccess) return DDP_HIP_E_HIP;
    }
  }
  if (c && ncclGroupEnd() != ncclSuccess) return DDP_HIP_E_COMM;
  if (hipStreamSynchronize(st) != hipSuccess) return DDP_HIP_E_HIP;
  return DDP_HIP_OK;
}
