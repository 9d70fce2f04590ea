// pick.hip -- device side of the multi-GPU best-cost pick (no reference counterpart).
// After forward_pass the cost of instance b's new trajectory is  sum_t costs_old[b][t] + dcost[b]  (ddp_fwd.ipp:24-26,54-58:
// cost_seq_aug of the reference trajectory plus the accepted step's sum_t (cost_new - cost_old)); both are resident.
//   pick_local_kernel   one workgroup: every instance's total (fixed order: lane-strided partials, then a tree), then the
//                       smallest total and the smallest local index attaining it -> {cost, global index} (16 bytes)
//   pick_final_kernel   one wave over the G gathered pairs: smallest cost, then smallest global index (no MINLOC in RCCL)
// csrc/comm.cpp puts one 16-byte ncclAllGather between the two: ONE collective, one 16-byte read-back, one sync.
#include "internal.h"

namespace {

__global__ __launch_bounds__(256) void pick_local_kernel(const double* costs, const double* dcost, const int32_t* active_or_null,
                                                         int64_t B, int64_t T1, int64_t rank, int64_t nranks, double* out_pair) {
  __shared__ double s_c[256];
  __shared__ long long s_i[256];
  const int tid = threadIdx.x;
  double best = __builtin_huge_val();
  long long besti = 0x7fffffffffffffffLL;
  for (int64_t b = tid; b < B; b += 256) {
    (void)active_or_null;
    const double* c = costs + b * T1;
    double s = 0.0;
    for (int64_t t = 0; t < T1; ++t) s += c[t];            // the order of Eigen's (costs_new - costs_old).sum() is not pinned: left to right
    s += dcost[b];
    if (s < best) { best = s; besti = (long long)(rank + b * nranks); }   // instance b of rank r is global instance r + b G
  }
  s_c[tid] = best; s_i[tid] = besti;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (tid < w) {
      const double oc = s_c[tid + w]; const long long oi = s_i[tid + w];
      if (oc < s_c[tid] || (oc == s_c[tid] && oi < s_i[tid])) { s_c[tid] = oc; s_i[tid] = oi; }
    }
    __syncthreads();
  }
  if (tid == 0) { out_pair[0] = s_c[0]; reinterpret_cast<long long*>(out_pair)[1] = s_i[0]; }
}

__global__ __launch_bounds__(64) void pick_final_kernel(const double* pairs, int G, double* out_pair) {
  if (threadIdx.x != 0) return;
  double best = pairs[0];
  long long besti = reinterpret_cast<const long long*>(pairs)[1];
  for (int g = 1; g < G; ++g) {
    const double c = pairs[2 * g];
    const long long i = reinterpret_cast<const long long*>(pairs)[2 * g + 1];
    if (c < best || (c == best && i < besti)) { best = c; besti = i; }
  }
  out_pair[0] = best;
  reinterpret_cast<long long*>(out_pair)[1] = besti;
}

}  // namespace

int pick_local_launch(ddp_hip_ctx* ctx, int64_t rank, int64_t nranks, double* out_pair, hipStream_t stream) {
  const double* costs = ctx->seq[DDP_HIP_SEQ_COSTS_OLD].ptr;
  if (!costs || !ctx->fw_dcost_acc_d) return DDP_HIP_E_UNSUPPORTED;
  hipLaunchKernelGGL(pick_local_kernel, dim3(1), dim3(256), 0, stream, costs, ctx->fw_dcost_acc_d, (const int32_t*)nullptr,
                     ctx->d.batch, ctx->d.T + 1, rank, nranks, out_pair);
  HIP_TRY(hipGetLastError());
  return DDP_HIP_OK;
}

int pick_final_launch(const double* pairs, int G, double* out_pair, hipStream_t stream) {
  hipLaunchKernelGGL(pick_final_kernel, dim3(1), dim3(64), 0, stream, pairs, G, out_pair);
  HIP_TRY(hipGetLastError());
  return DDP_HIP_OK;
}
