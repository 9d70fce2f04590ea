// bwd_fast.h -- compile-time-shaped backward-sweep kernels (included by bwd.hip).
//
// bwd_assemble_fast<N, M>: same arithmetic as bwd_assemble, restructured around the HBM stream:
//   * every product against a column-major (N x L) block -- the tensor slabs f_xx(:,:,c), f_ux(:,:,c),
//     f_uu(:,:,c) AND the dense f_x^T w / f_u^T w products -- is one "slab contraction"
//     out[j] += sum_i vec[i] * T[i + j*N]:  256 lanes read the slab with consecutive 16-byte loads
//     (1 KiB per wave instruction, whole slab in flight at once), write two-term partials to LDS at
//     [column][pair] (column stride N/2+1: conflict free), then lane j adds the N/2 partials of column j;
//   * slabs alternate between two register buffers, so the loads of slab k+1 are in flight while slab k
//     is reduced: the kernel never waits on an empty memory pipe between slabs.
// bwd_gains_fast<N, M>: Cholesky with the forward substitution of all N+1 right-hand sides fused into
//   the column loop, then a column-oriented back substitution -- every phase uses all 256 lanes.
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

// out[j] += sum_i vec[i] * slab[i + j*O], j < L  (slab already in `buf`)
template <int O, int L>
__device__ __forceinline__ void slab_finish(const f64x2 (&buf)[SlabShape<O, L>::R], const double* s_vec, double* s_part,
                                            double* out) {
  using S = SlabShape<O, L>;
#pragma unroll
  for (int r = 0; r < S::R; ++r) {
    const int f = threadIdx.x + r * BSF;
    if (r < S::R - 1 || f < S::TOTAL) {
      const int j = f / S::HP;
      const int ip = f - j * S::HP;
      const f64x2 vv = *reinterpret_cast<const f64x2*>(s_vec + 2 * ip);
      s_part[j * S::LD + ip] = vv.x * buf[r].x + vv.y * buf[r].y;
    }
  }
  __syncthreads();
  if (threadIdx.x < L) {
    const double* pj = s_part + threadIdx.x * S::LD;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < S::HP; ++k) s += pj[k];
    out[threadIdx.x] += s;
  }
  __syncthreads();
}

template <int N, int M>
__global__ __launch_bounds__(BSF, BWD_WAVES_PER_SIMD) void bwd_assemble_fast(BwdParams p, int64_t t) {
  // XCD-aware placement: blocks b and b+8 share an XCD (and its L2); keep all jobs of one instance on one
  // XCD so that its f_x, f_u, V_xx are fetched into a single L2.  Speed only, never correctness.
  int b, jb;
  {
    const int njobs = (int)gridDim.x, B = (int)gridDim.y;
    const int lin = blockIdx.y * njobs + blockIdx.x;
    if ((B & 7) == 0) {
      const int xcd = lin & 7, k = lin >> 3;
      b = xcd + 8 * (k / njobs);
      jb = k % njobs;
    } else { b = blockIdx.y; jb = blockIdx.x; }
  }
  if (p.status[b] != 0) return;
  const BwdJob job = p.jobs[jb];
  constexpr int n = N, m = M;
  const int64_t T = p.d.T;
  const int e = (int)p.ne[t];
  const int64_t Eo = p.Epre[t];
  const int64_t Etot = p.d.Etot;
  const double mu = p.mu[b];
  const int tid = threadIdx.x;
  const int kind = job.kind, c0 = job.c0, cn = job.cn;
  const int rows = kind == 0 ? n + m : m;

  const double* Vx = p.ws_V + (int64_t)b * (n + n * n);
  const double* Vxx = Vx + n;
  double* Q = p.ws_Q + (int64_t)b * (n + m + n * n + m * n + m * m);
  double* Qx = Q;
  double* Qu = Qx + n;
  double* Qxx = Qu + m;
  double* Qux = Qxx + n * n;
  double* Quu = Qux + m * n;

  const int64_t bt = (int64_t)b * T + t;
  const double* fx = p.fx + bt * n * n;
  const double* fu = p.fu + bt * n * m;
  const double* eqv = p.eq_val + (int64_t)b * Etot + Eo;
  const double* eqx = p.eq_x + ((int64_t)b * Etot + Eo) * n;
  const double* equ = p.eq_u + ((int64_t)b * Etot + Eo) * m;
  const double* pe = p.mult_val + (int64_t)b * Etot + Eo;
  const double* pex = p.mult_jac + ((int64_t)b * Etot + Eo) * n;

  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* s_v = smem;                          // n
  double* s_W = s_v + n;                       // n * cn   (16-byte aligned: n even)
  double* s_out = s_W + n * cn;                // rows * cn
  double* s_part = s_out + (n + m) * cn;       // (n/2+1) * n
  double* s_tmp = s_part + (n / 2 + 1) * n;    // emax

  // V_x and this job's columns of F = f_x (x-job) or f_u (u-job) to LDS (s_part is free until the first slab)
  const double* Fc = (kind == 0 ? fx : fu) + c0 * n;
  double* s_F = s_part;
  for (int i = tid; i < n; i += BSF) s_v[i] = Vx[i];
  for (int i = tid; i < e; i += BSF) s_tmp[i] = pe[i] + mu * eqv[i];   // ddp_bwd.ipp:46
  for (int idx = tid; idx < n * cn; idx += BSF) s_F[idx] = Fc[idx];
  __syncthreads();

  // W = V_xx * F(:, c0:c0+cn): one lane per entry, 76 independent (coalesced) loads, 19 in flight
  for (int idx = tid; idx < n * cn; idx += BSF) {
    const int r = idx % n, c = idx / n;
    const double* fc = s_F + c * n;
    double s = 0.0;
#pragma unroll 19
    for (int l = 0; l < n; ++l) s += Vxx[r + l * n] * fc[l];
    s_W[idx] = s;
  }
  // l terms (ddp_bwd.ipp:70, :77, :83)
  for (int idx = tid; idx < rows * cn; idx += BSF) {
    const int r = idx % rows, c = idx / rows;
    const int col = c0 + c;
    double acc;
    if (kind == 0) acc = r < n ? p.lxx[bt * n * n + r + col * n] : p.lux[bt * m * n + (r - n) + col * m];
    else acc = p.luu[bt * m * m + r + col * m];
    s_out[idx] = acc;
  }
  // Q_x / Q_u entries of this job's columns (:61-68): one wave per column, lanes across the rows of F
  for (int c = tid >> 6; c < cn; c += BSF >> 6) {
    const int col = c0 + c;
    const int lane = tid & 63;
    double s = 0.0;
    if (lane < n / 2) {
      const f64x2 a = *reinterpret_cast<const f64x2*>(s_F + c * n + 2 * lane);
      const f64x2 vv = *reinterpret_cast<const f64x2*>(s_v + 2 * lane);
      s = a.x * vv.x + a.y * vv.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) {
      double acc;
      if (kind == 0) {
        acc = p.lx[bt * n + col];
        acc += s;
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < e; ++i) { s1 += eqx[i + col * e] * s_tmp[i]; s2 += pex[i + col * e] * eqv[i]; }
        acc += s1;
        acc += s2;
        Qx[col] = acc;
      } else {
        acc = p.lu[bt * m + col];
        acc += s;
        double s1 = 0.0;
        for (int i = 0; i < e; ++i) s1 += equ[i + col * e] * s_tmp[i];
        acc += s1;
        Qu[col] = acc;
      }
    }
  }
  const bool tens = p.has_tensors != 0;

  // multiplier terms (:72-74, :79-80, :85-86), added between the dense and the tensor terms as the
  // reference orders them; e is small, one lane per output entry
  auto add_eq_terms = [&]() {
    if (e <= 0) return;
    for (int idx = tid; idx < rows * cn; idx += BSF) {
      const int r = idx % rows, c = idx / rows;
      const int col = c0 + c;
      double acc = s_out[idx];
      if (kind == 0) {
        if (r < n) {
          double s1 = 0.0, s2 = 0.0;
          for (int i = 0; i < e; ++i) {
            const double tmp2 = pex[i + col * e] + mu * eqx[i + col * e];
            s1 += eqx[i + r * e] * tmp2;
            s2 += pex[i + r * e] * eqx[i + col * e];
          }
          acc += s1;
          acc += s2;
        } else {
          double s1 = 0.0;
          for (int i = 0; i < e; ++i) s1 += equ[i + (r - n) * e] * (pex[i + col * e] + mu * eqx[i + col * e]);
          acc += s1;
        }
      } else {
        double s1 = 0.0;
        for (int i = 0; i < e; ++i) s1 += equ[i + r * e] * equ[i + col * e];
        acc += s1 * mu;
      }
      if (tens) {
        const double* Te;
        int Ld, jrow;
        if (kind == 0) {
          if (r < n) { Te = p.eq_xx + ((int64_t)b * Etot + Eo) * n * n; Ld = n; jrow = r; }
          else { Te = p.eq_ux + ((int64_t)b * Etot + Eo) * m * n; Ld = m; jrow = r - n; }
        } else { Te = p.eq_uu + ((int64_t)b * Etot + Eo) * m * m; Ld = m; jrow = r; }
        const double* col_ptr = Te + ((int64_t)jrow + (int64_t)col * Ld) * e;
        double s3 = 0.0;
        for (int i = 0; i < e; ++i) s3 += s_tmp[i] * col_ptr[i];
        acc += s3;
      }
      s_out[idx] = acc;
    }
    __syncthreads();
  };

  // Everything that remains is a sequence of identical "units": a 76 x 38 column-major block (one f_u /
  // f_ux / f_uu slab, or half of an f_x / f_xx slab) contracted with a vector.  Units 0 .. UD-1 are the dense
  // products f^T w_c (:71, :78, :84), units UD .. 2 UD-1 the V_x-contracted tensor slabs (:75, :81, :87) --
  // the HBM stream.  Four register buffers rotate, so three units (69 KB) per workgroup are always in
  // flight behind the one being reduced; the LDS partials are double buffered: one barrier per unit.
  constexpr int UPC = 3;                                  // units per x-column: f_x lo, f_x hi, f_u
  const int upc = kind == 0 ? UPC : 1;
  const int UD = upc * cn;
  const int U = tens ? 2 * UD : UD;
  const double* Txx = p.fxx + (bt * n + c0) * (int64_t)n * n;   // f_xx(:,:,c0 + c): n x n slabs, contiguous in c
  const double* Tux = p.fux + (bt * n + c0) * (int64_t)n * m;   // f_ux(:,:,c0 + c): n x m slabs
  const double* Tuu = p.fuu + (bt * m + c0) * (int64_t)n * m;   // f_uu(:,:,c0 + c): n x m slabs
  auto unit_ptr = [&](int u) -> const double* {
    const bool dense = u < UD;
    const int d = dense ? u : u - UD;
    const int c = d / upc, part = d - c * upc;
    if (kind == 0) {
      if (dense) return part < 2 ? fx + part * (M * n) : fu;
      return part < 2 ? Txx + (int64_t)c * n * n + part * (M * n) : Tux + (int64_t)c * n * m;
    }
    return dense ? fu : Tuu + (int64_t)c * n * m;
  };
  auto unit_vec = [&](int u) -> const double* { return u < UD ? s_W + ((u) / upc) * n : s_v; };
  auto unit_out = [&](int u) -> double* {
    const int d = u < UD ? u : u - UD;
    const int c = d / upc, part = d - c * upc;
    return s_out + c * rows + part * M;
  };
  using US = SlabShape<N, M>;
  f64x2 buf0[US::R], buf1[US::R], buf2[US::R], buf3[US::R];
  double* s_p0 = s_part;
  double* s_p1 = s_part + US::LD * M;
#define UNIT_ISSUE(BUF, u)                                                                  \
  do {                                                                                      \
    if ((u) < U) {                                                                          \
      if ((u) < UD) slab_issue<N, M, false>(unit_ptr(u), BUF);                              \
      else slab_issue<N, M, BWD_NT>(unit_ptr(u), BUF);                                      \
    }                                                                                       \
  } while (0)
#define UNIT_STEP(BUF, u)                                                                   \
  do {                                                                                      \
    if ((u) < U) {                                                                          \
      if ((u) == UD && e > 0) { __syncthreads(); add_eq_terms(); }                          \
      double* sp = ((u) & 1) ? s_p1 : s_p0;                                                 \
      const double* vec = unit_vec(u);                                                      \
      _Pragma("unroll") for (int r = 0; r < US::R; ++r) {                                   \
        const int f = tid + r * BSF;                                                        \
        if (r < US::R - 1 || f < US::TOTAL) {                                               \
          const int j = f / US::HP;                                                         \
          const int ip = f - j * US::HP;                                                    \
          const f64x2 vv = *reinterpret_cast<const f64x2*>(vec + 2 * ip);                   \
          sp[j * US::LD + ip] = vv.x * BUF[r].x + vv.y * BUF[r].y;                          \
        }                                                                                   \
      }                                                                                     \
      __syncthreads();                                                                      \
      if (tid < M) {                                                                        \
        const double* pj = sp + tid * US::LD;                                               \
        double sacc = 0.0;                                                                  \
        _Pragma("unroll") for (int k = 0; k < US::HP; ++k) sacc += pj[k];                   \
        unit_out(u)[tid] += sacc;                                                           \
      }                                                                                     \
      UNIT_ISSUE(BUF, (u) + 4);                                                             \
    }                                                                                       \
  } while (0)

  { const int u0 = BWD_EXP_SKIP_DENSE ? ((UD + 3) & ~3) : 0;
    UNIT_ISSUE(buf0, u0); UNIT_ISSUE(buf1, u0 + 1); UNIT_ISSUE(buf2, u0 + 2); UNIT_ISSUE(buf3, u0 + 3); }
  __syncthreads();   // s_W, s_out (l terms) and s_F readers are done: s_part may be overwritten
  for (int u = BWD_EXP_SKIP_DENSE ? ((UD + 3) & ~3) : 0; u < U; u += 4) {
    UNIT_STEP(buf0, u);
    UNIT_STEP(buf1, u + 1);
    UNIT_STEP(buf2, u + 2);
    UNIT_STEP(buf3, u + 3);
  }
  __syncthreads();
  if (U == UD && e > 0) add_eq_terms();
#undef UNIT_STEP
#undef UNIT_ISSUE

  for (int idx = tid; idx < rows * cn; idx += BSF) {
    const int r = idx % rows, c = idx / rows;
    const int col = c0 + c;
    if (kind == 0) {
      if (r < n) Qxx[r + col * n] = s_out[idx];
      else Qux[(r - n) + col * m] = s_out[idx];
    } else {
      Quu[r + col * m] = s_out[idx];
    }
  }
}

template <int N, int M>
__global__ __launch_bounds__(BS) void bwd_gains_fast(BwdParams p, int64_t t) {
  const int b = blockIdx.x;
  if (p.status[b] != 0) return;
  constexpr int n = N, m = M, nx = N;
  const int64_t T = p.d.T;
  const int tid = threadIdx.x;
  const int64_t bt = (int64_t)b * T + t;

  double* Vx = p.ws_V + (int64_t)b * (n + n * n);
  double* Vxx = Vx + n;
  const double* Q = p.ws_Q + (int64_t)b * (n + m + n * n + m * n + m * m);
  const double* Qx = Q;
  const double* Qu = Qx + n;
  const double* Qxx = Qu + m;
  const double* Qux = Qxx + n * n;
  const double* Quu = Qux + m * n;

  constexpr int lda = M | 1, ldr = M | 1, NR = N + 1;
  constexpr int RS = 3;                      // lanes per right-hand side: lane (c, s) owns rows s, s+3, ... of column c
  constexpr int RQ = (M + RS - 1) / RS;      // rows per lane (13)
  static_assert(NR * RS <= BS, "right-hand sides must fit the workgroup");
  __shared__ double A[lda * M];              // Q_uu + reg I -> Cholesky factor (lower)
  __shared__ double R[ldr * NR];             // [k | K] for the V update
  __shared__ double S[ldr * N];              // Q_ux
  __shared__ double Y[2][NR];                // the pivot row of the right-hand sides, published per step

#ifdef DDP_GAINS_TIMING
  unsigned long long tk[8]; int tki = 0;
#define STAMP() do { tk[tki++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP() do {} while (0)
#endif
  STAMP();
  const double reg = p.reg[b];
  for (int idx = tid; idx < m * n; idx += BS) S[idx % m + (idx / m) * ldr] = Qux[idx];
  // right-hand sides [-Q_u | -Q_ux] (:135-136) in registers: lane (rc, rs) owns rows rs, rs+3, ... of column rc
  const int rc = tid % NR, rs = tid / NR;
  const bool rhs_lane = tid < NR * RS;
  double r[RQ];
  if (rhs_lane) {
    const double* src = rc == 0 ? Qu : Qux + (rc - 1) * m;
#pragma unroll
    for (int q = 0; q < RQ; ++q) { const int l = rs + RS * q; r[q] = l < m ? -src[l] : 0.0; }
  }
  // trailing matrix of the factorisation in registers: lane (ti, tj) owns row ti, columns tj, tj+6, ... <= ti
  constexpr int TJ = BS / M;                 // 6
  constexpr int AQ = (M + TJ - 1) / TJ;      // 7
  const int ti = tid % M, tj = tid / M;
  const bool a_lane = tj < TJ;
  double a[AQ];
#pragma unroll
  for (int q = 0; q < AQ; ++q) {
    const int j = tj + TJ * q;
    a[q] = (a_lane && j <= ti) ? Quu[ti + j * m] + (ti == j ? reg : 0.0) : 0.0;         // :104
  }
  if (a_lane && tj == 0) A[ti] = a[0];       // raw column 0
  __syncthreads();
  STAMP();

  // Cholesky (lower triangle only; fail <=> pivot <= 0, :105) with the forward substitution fused in: column
  // k of L is final after step k, so y_k = r_k / L_kk is published and r_l -= L_lk y_k (l > k) rides along
  // with the trailing update.  Per entry the updates arrive in ascending k: the order of Eigen's unblocked
  // LLT and of its row-wise substitution.  Every LDS read below is unconditional (clamped index + select), so
  // a step is two short phases instead of a chain of branchy, latency-exposed reads.
  bool failed = false;
  for (int k = 0; k < m; ++k) {
    const double piv = A[k + k * lda];
    if (piv <= 0.0) { failed = true; break; }
    const double dk = sqrt(piv);
    if (a_lane && tj == 0 && ti > k) A[ti + k * lda] = A[ti + k * lda] / dk;
    if (rhs_lane && rs == k % RS) {
      const int qk = k / RS;
      double rk = 0.0;
#pragma unroll
      for (int q = 0; q < RQ; ++q) rk = q == qk ? r[q] : rk;
      rk = rk / dk;
#pragma unroll
      for (int q = 0; q < RQ; ++q) r[q] = q == qk ? rk : r[q];
      Y[k & 1][rc] = rk;
    }
    __syncthreads();
    const double* Lk = A + k * lda;
    if (a_lane) {
      const double lik = Lk[ti];
#pragma unroll
      for (int q = 0; q < AQ; ++q) {
        const int j = tj + TJ * q;
        const double ljk = Lk[j < m ? j : m - 1];
        a[q] = (j > k && j <= ti) ? a[q] - lik * ljk : a[q];
      }
      const int k1 = k + 1;
      if (k1 < m && tj == k1 % TJ && ti >= k1) {
        const int q1 = k1 / TJ;
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < AQ; ++q) v = q == q1 ? a[q] : v;
        A[ti + k1 * lda] = v;                 // raw column k+1, final after this step's update
      }
    }
    if (rhs_lane) {
      const double yk = Y[k & 1][rc];
#pragma unroll
      for (int q = 0; q < RQ; ++q) {
        const int l = rs + RS * q;
        const double llk = Lk[l < m ? l : m - 1];
        r[q] = (l > k && l < m) ? r[q] - llk * yk : r[q];
      }
    }
    if (tid == k) A[k + k * lda] = dk;
    __syncthreads();
  }
  if (failed) {
    if (tid == 0) {
      double rg = p.reg[b], mu = p.mu[b];
      if (rg < mu) rg = mu;        // :106-108
      mu *= 2;                     // :109
      rg *= 2;                     // :110
      p.reg[b] = rg;
      p.mu[b] = mu;
      p.status[b] = 1;
      p.restarts[b] += 1;
    }
    return;
  }
  STAMP();
  // back substitution L^T x = y, column oriented; the published row alternates between two LDS buffers so
  // that one barrier per step is enough
  for (int k = m - 1; k >= 0; --k) {
    if (rhs_lane && rs == k % RS) {
      const double dk = A[k + k * lda];
      const int qk = k / RS;
      double rk = 0.0;
#pragma unroll
      for (int q = 0; q < RQ; ++q) rk = q == qk ? r[q] : rk;
      rk = rk / dk;
#pragma unroll
      for (int q = 0; q < RQ; ++q) r[q] = q == qk ? rk : r[q];
      Y[k & 1][rc] = rk;
    }
    __syncthreads();
    if (rhs_lane) {
      const double xk = Y[k & 1][rc];
#pragma unroll
      for (int q = 0; q < RQ; ++q) {
        const int i = rs + RS * q;
        const double lki = A[k + (i < m ? i : m - 1) * lda];
        r[q] = i < k ? r[q] - lki * xk : r[q];
      }
    }
  }

  STAMP();
  double* fbo = p.fb_origin + bt * nx;
  const double* xt = p.x + ((int64_t)b * (T + 1) + t) * nx;
  for (int i = tid; i < nx; i += BS) fbo[i] = xt[i];                         // :134
  if (rhs_lane) {
    double* dst = rc == 0 ? p.fb_val + bt * m : p.fb_jac + bt * m * n + (rc - 1) * m;
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const int l = rs + RS * q;
      if (l < m) { dst[l] = r[q]; R[l + rc * ldr] = r[q]; }
    }
  }
  __syncthreads();

  STAMP();
  // V_x = Q_x + Q_ux^T k (:142-143);  V_xx = Q_xx + Q_ux^T K (:145-146), 1 x 4 register tiles
  for (int i = tid; i < n; i += BS) {
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < m; ++l) s += S[l + i * ldr] * R[l];
    const double v = Qx[i] + s;
    Vx[i] = v;
    if (p.vx_trace) p.vx_trace[bt * n + i] = v;
  }
  static_assert(N % 4 == 0, "V_xx tiling");
  for (int idx = tid; idx < n * (n / 4); idx += BS) {
    const int i = idx % n, j0 = (idx / n) * 4;
    const double* si = S + i * ldr;
    const double* k0 = R + (j0 + 1) * ldr;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int l = 0; l < m; ++l) {
      const double sv = si[l];
      s0 += sv * k0[l];
      s1 += sv * k0[l + ldr];
      s2 += sv * k0[l + 2 * ldr];
      s3 += sv * k0[l + 3 * ldr];
    }
    const double sv4[4] = {s0, s1, s2, s3};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = i + (j0 + q) * n;
      const double v = Qxx[o] + sv4[q];
      Vxx[o] = v;
      if (p.vxx_trace) p.vxx_trace[bt * n * n + o] = v;
    }
  }
  STAMP();
#ifdef DDP_GAINS_TIMING
  if (b == 0 && tid == 0 && t == 5)
    printf("gains cycles: load %llu llt+fwd %llu back %llu store %llu vupdate %llu\n", tk[1] - tk[0], tk[2] - tk[1],
           tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4]);
#endif
  if (t == 0 && tid == 0) p.status[b] = 2;                                   // :149-151
}
