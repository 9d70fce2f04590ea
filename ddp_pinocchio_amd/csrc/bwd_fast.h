// bwd_fast.h -- helpers of the compile-time-shaped streaming kernel (included by bwd.hip before bwd_split.h):
// a "slab" is a column-major (O x L) block of a flat sequence; 512 lanes read it with consecutive 16-byte loads
// (1 KiB per wave instruction) into a register buffer that is reduced later, so loads stay in flight meanwhile.
#pragma once

#ifndef BWD_NT
#define BWD_NT true
#endif
#ifndef BWD_EXP_SKIP_DENSE
#define BWD_EXP_SKIP_DENSE 0
#endif
#ifndef BWD_WAVES_PER_SIMD
#define BWD_WAVES_PER_SIMD 4
#endif
constexpr int BSF = 512;   // workgroup size of the streaming kernel: one N x N slab = 5.6 loads per lane

template <int O, int L>
struct SlabShape {
  static constexpr int HP = O / 2, LD = O / 2 + 1, TOTAL = (O / 2) * L, R = ((O / 2) * L + BSF - 1) / BSF;
};

template <int O, int L, bool NT>
__device__ __forceinline__ void slab_issue(const double* __restrict__ Tn, f64x2 (&buf)[SlabShape<O, L>::R]) {
  using S = SlabShape<O, L>;
  const f64x2* __restrict__ T2 = reinterpret_cast<const f64x2*>(Tn);
#pragma unroll
  for (int r = 0; r < S::R; ++r) {
    const int f = threadIdx.x + r * BSF;
    if (r < S::R - 1 || f < S::TOTAL) buf[r] = NT ? __builtin_nontemporal_load(&T2[f]) : T2[f];
  }
}


// the same, for the columns jmin .. L-1 of the slab only (the others are not fetched; their buffer words are zero, so that the
// reduction that follows leaves zeros for them): symmetric tensors, bwd_split.h
template <int O, int L, bool NT>
__device__ __forceinline__ void slab_issue_from(const double* __restrict__ Tn, f64x2 (&buf)[SlabShape<O, L>::R], int jmin) {
  using S = SlabShape<O, L>;
  const f64x2* __restrict__ T2 = reinterpret_cast<const f64x2*>(Tn);
  const int fmin = jmin * S::HP;
#pragma unroll
  for (int r = 0; r < S::R; ++r) {
    const int f = threadIdx.x + r * BSF;
    if ((r < S::R - 1 || f < S::TOTAL) && f >= fmin) buf[r] = NT ? __builtin_nontemporal_load(&T2[f]) : T2[f];
    else buf[r] = f64x2{0.0, 0.0};
  }
}
