// comm.cpp -- the one exchange step of the multi-GPU shard (no reference counterpart).
// Independent instances are sharded over ranks with no data-path collective; after a sweep the ranks agree on
// the best instance with two 8-byte RCCL all-reduces over xGMI: min of the cost, then min of the masked global
// index (RCCL has no MINLOC).  Latency-bound (~10 us); bandwidth is irrelevant.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include <new>

#include "ddp_hip/ddp_hip.h"

struct ddp_hip_comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int device = 0, rank = 0, nranks = 1;
  double* d_cost = nullptr;
  int64_t* d_idx = nullptr;
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
      hipMalloc(&c->d_cost, sizeof(double)) != hipSuccess || hipMalloc(&c->d_idx, sizeof(int64_t)) != hipSuccess) {
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
