// lin.hip -- linearisation (placeholder until the device ABA lands)
#include "internal.h"
int lin_setup(ddp_hip_ctx*) { return DDP_HIP_OK; }
void lin_teardown(ddp_hip_ctx*) {}
extern "C" int ddp_hip_linearize(ddp_hip_ctx*) { return DDP_HIP_E_UNSUPPORTED; }
