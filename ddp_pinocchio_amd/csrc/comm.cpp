// comm.cpp -- multi-GPU shard exchange over RCCL (placeholder)
#include "ddp_hip/ddp_hip.h"
extern "C" int ddp_hip_comm_unique_id(unsigned char*) { return DDP_HIP_E_UNSUPPORTED; }
extern "C" int ddp_hip_comm_init(const unsigned char*, int, int, int, ddp_hip_comm**) { return DDP_HIP_E_UNSUPPORTED; }
extern "C" int ddp_hip_comm_destroy(ddp_hip_comm*) { return DDP_HIP_E_UNSUPPORTED; }
extern "C" int ddp_hip_shard_best(ddp_hip_comm*, double, int64_t, double*, int64_t*) { return DDP_HIP_E_UNSUPPORTED; }
