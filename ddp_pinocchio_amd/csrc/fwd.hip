// fwd.hip -- forward sweep (placeholder until the device ABA lands)
#include "internal.h"
int fwd_setup(ddp_hip_ctx*) { return DDP_HIP_OK; }
void fwd_teardown(ddp_hip_ctx*) {}
extern "C" int ddp_hip_rollout(ddp_hip_ctx*) { return DDP_HIP_E_UNSUPPORTED; }
extern "C" int ddp_hip_forward(ddp_hip_ctx*, const double*, int32_t, double*, double*) { return DDP_HIP_E_UNSUPPORTED; }
extern "C" int ddp_hip_cost_seq_aug(ddp_hip_ctx*, int, const double*) { return DDP_HIP_E_UNSUPPORTED; }
