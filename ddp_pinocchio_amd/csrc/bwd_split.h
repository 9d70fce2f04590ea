// bwd_split.h -- the backward step as the two kernels of SURVEY.md's kernel map (included by bwd.hip):
//
//   K3  bwd_contract<N, M>   grid (jobs, batch).  Pure HBM stream: C(j,k) = sum_i V_x,i T(i,j,k) for the three
//       second-order tensors of one timestep (tensor.hpp:179-198 as called at ddp_bwd.ipp:75,81,87).
//       Reads every tensor byte exactly once with 16-byte coalesced loads, three 69 KB units per workgroup
//       always in flight; 0.25 flop/byte.
//   K4  bwd_riccati<N, M>    grid (batch).  Everything else of the step, LDS / register resident:
//       Q = l + f^T V_xx f + multiplier terms + C (ddp_bwd.ipp:61-87, in the reference's order of terms),
//       LLT (:104-105), gains (:134-136), V update (:142-146), and -- while the new V_xx is still in LDS --
//       the dense product D = [f_x f_u]^T V_xx [f_x f_u] of the NEXT step to be processed (t-1).
#pragma once

constexpr int BSR = 512;   // workgroup size of K4

template <int N, int M>
__global__ __launch_bounds__(BSF, BWD_WAVES_PER_SIMD) void bwd_contract(BwdParams p, int64_t t) {
  // XCD-aware placement: blocks b and b+8 share an XCD; keep the jobs of one instance on one XCD (speed only)
  int b, jb;
  {
    const int njobs = (int)gridDim.x, B = (int)gridDim.y;
    const int lin = blockIdx.y * njobs + blockIdx.x;
    if ((B & 7) == 0) {
      const int xcd = lin & 7, k = lin >> 3;
      b = xcd + 8 * (k / njobs);
      jb = k % njobs;
    } else { b = blockIdx.y; jb = blockIdx.x; }
    b += p.b0;
  }
  if (p.status[b] != 0) return;
  const BwdJob job = p.jobs[jb];
  constexpr int n = N, m = M;
  const int64_t T = p.d.T;
  const int tid = threadIdx.x;
  // kind 0: x-columns (f_xx in two half-slabs, f_ux), 1: u-columns (f_uu), 2: x-columns c >= M without their first half-slab.
  // SYMMETRIC tensors (p.sym_tensors: this context's own mode-2 or zero tensors -- the stencil forms one value for the entries
  // (i, j, c) and (i, c, j), problem.hpp:283-292): of slab c only the columns j >= c are read -- one contiguous tail of the slab
  // -- and the contraction C(j, c) is written to (j, c) and to its mirror image (c, j); the entries j < c come from the jobs of
  // the columns j.  The mirrored sum is the very sum the skipped column would have given, bit for bit.  Half of f_xx and of
  // f_uu (2.16 of 6.31 MB per (instance, t)) is never read, and the stencil does not have to write it (lin_static.hip)
  const int kind = job.kind, c0 = job.c0, cn = job.cn;
  const int rows = kind == 1 ? m : n + m;
  static_assert(N == 2 * M, "half-slabs: the state tangent is twice the control dimension");
  const int64_t bt = (int64_t)b * T + t;

  const double* Vx = p.ws_V + (int64_t)b * (n + n * n);
  // c_accumulate 0: store into the Q workspace (round 1's K4 adds the rest); 1: add to what K5 left there; 2: store into a
  // workspace of its own (K5 runs beside this kernel, K4' forms P + C)
  double* C = p.c_accumulate == 2 ? p.ws_D + (int64_t)b * (n * n + m * n + m * m)
                                  : p.ws_Q + (int64_t)b * (n + m + n * n + m * n + m * m) + n + m;
  double* Cxx = C;
  double* Cux = Cxx + n * n;
  double* Cuu = Cux + m * n;

  using US = SlabShape<N, M>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* s_v = smem;                          // n
  double* s_out = s_v + n;                     // rows * cn
  double* s_p0 = s_out + (n + m) * cn;         // US::LD * M
  double* s_p1 = s_p0 + US::LD * M;

  // units: 76 x 38 column-major blocks; an x-column is f_xx(:,0:38,c), f_xx(:,38:76,c), f_ux(:,:,c); a u-column f_uu(:,:,c)
  const int upc = kind == 0 ? 3 : (kind == 2 ? 2 : 1);
  const int part0 = kind == 2 ? 1 : 0;              // first part (row block of the column) this job reads
  const int U = upc * cn;
  const double* Txx = p.fxx + (bt * n + c0) * (int64_t)n * n;
  const double* Tux = p.fux + (bt * n + c0) * (int64_t)n * m;
  const double* Tuu = p.fuu + (bt * m + c0) * (int64_t)n * m;
  auto unit_ptr = [&](int u) -> const double* {
    const int c = u / upc, part = u - c * upc + part0;
    if (kind != 1) return part < 2 ? Txx + (int64_t)c * n * n + part * (M * n) : Tux + (int64_t)c * n * m;
    return Tuu + (int64_t)c * n * m;
  };
  const bool sym = p.sym_tensors != 0;
  auto unit_jmin = [&](int u) -> int {                 // first column of the unit that is read
    if (!sym) return 0;
    const int c = u / upc, part = u - c * upc + part0, col = c0 + c;
    if (kind == 1) return col;                        // f_uu(:, j, col): j >= col
    if (part == 0) return col < M ? col : M;          // f_xx(:, 0:M, col)
    if (part == 1) return col > M ? col - M : 0;      // f_xx(:, M:N, col)
    return 0;                                         // f_ux
  };
  f64x2 buf0[US::R], buf1[US::R], buf2[US::R], buf3[US::R];
#define C_ISSUE(BUF, u) do { if ((u) < U) slab_issue_from<N, M, BWD_NT>(unit_ptr(u), BUF, unit_jmin(u)); } while (0)
  C_ISSUE(buf0, 0); C_ISSUE(buf1, 1); C_ISSUE(buf2, 2); C_ISSUE(buf3, 3);

  for (int i = tid; i < n; i += BSF) s_v[i] = Vx[i];
  for (int i = tid; i < rows * cn; i += BSF) s_out[i] = 0.0;
  __syncthreads();

#define C_STEP(BUF, u)                                                                      \
  do {                                                                                      \
    if ((u) < U) {                                                                          \
      double* sp = ((u) & 1) ? s_p1 : s_p0;                                                 \
      _Pragma("unroll") for (int r = 0; r < US::R; ++r) {                                   \
        const int f = tid + r * BSF;                                                        \
        if (r < US::R - 1 || f < US::TOTAL) {                                               \
          const int j = f / US::HP;                                                         \
          const int ip = f - j * US::HP;                                                    \
          const f64x2 vv = *reinterpret_cast<const f64x2*>(s_v + 2 * ip);                   \
          sp[j * US::LD + ip] = vv.x * BUF[r].x + vv.y * BUF[r].y;                          \
        }                                                                                   \
      }                                                                                     \
      __syncthreads();                                                                      \
      if (tid < M) {                                                                        \
        const double* pj = sp + tid * US::LD;                                               \
        double sacc = 0.0;                                                                  \
        _Pragma("unroll") for (int k = 0; k < US::HP; ++k) sacc += pj[k];                   \
        const int c_ = (u) / upc, part_ = (u) - c_ * upc + part0;                           \
        s_out[c_ * rows + part_ * M + tid] = sacc;                                          \
      }                                                                                     \
      C_ISSUE(BUF, (u) + 4);                                                                \
    }                                                                                       \
  } while (0)
  for (int u = 0; u < U; u += 4) {
    C_STEP(buf0, u);
    C_STEP(buf1, u + 1);
    C_STEP(buf2, u + 2);
    C_STEP(buf3, u + 3);
  }
  __syncthreads();
#undef C_STEP
#undef C_ISSUE
  for (int idx = tid; idx < rows * cn; idx += BSF) {
    const int r = idx % rows, c = idx / rows;
    const int col = c0 + c;
    // c_accumulate: the workspace already holds every other term of Q (K5, bwd_v2.h); the tensor term comes last in the
    // reference as well (ddp_bwd.ipp:75,81,87)
    if (kind == 2 && r < M) continue;                // (the half-slab this job kind leaves out)
    const bool square = kind == 1 || r < n;          // an entry of C_xx / C_uu (f_ux is not symmetric)
    if (sym && square && r < col) continue;          // the mirror image: written by the job of column r
    double* dst = kind != 1 ? (r < n ? Cxx + r + col * n : Cux + (r - n) + col * m) : Cuu + r + col * m;
    *dst = p.c_accumulate == 1 ? *dst + s_out[idx] : s_out[idx];
    if (sym && square && r > col) {
      double* dm = kind != 1 ? Cxx + col + r * n : Cuu + col + r * m;   // C(col, r) = C(r, col)
      *dm = p.c_accumulate == 1 ? *dm + s_out[idx] : s_out[idx];
    }
  }
}

// ---- K3h: the same contraction for this context's own mode-2 tensors, reading what is not known in advance ------------------
// Two structural facts about the tensors the stencil (lin_static.hip / lin.hip) writes, both exact in floating point:
//  (1) symmetry: entries (i, j, c) and (i, c, j) of f_xx / f_uu hold one value (problem.hpp:283-292) -> of slab c only the
//      columns j >= c are read, the contraction goes to (j, c) and to (c, j);
//  (2) the CONFIGURATION rows i < M of every column are exact zeros except at i = c mod M and i = j mod M (x directions): the
//      first M rows of f are q + dt v, which does not see the dynamics, so row i of a stencil point differs from the base
//      point's only when a direction is q_i or v_i; everything else differences bit-identical numbers.
// So a column contributes  sum_{i >= M} V_x,i T(i, j, c)  plus at most two terms from its first M rows.  A unit is the lower
// half (rows M .. N-1: 304 contiguous bytes per column) of N consecutive columns -- one slab of f_xx, two of f_ux or f_uu -- as
// many bytes as bwd_contract's unit; the (at most) two entries of the upper half come with two 8-byte loads per column.  The sums
// are formed in bwd_contract's order -- two-term partials over the row pairs (2 ip, 2 ip + 1), then ip ascending, the zero
// partials adding nothing -- so the result is bwd_contract's bit for bit (tests/test_round3_boundary.py).
// Bytes per (instance, t): 2.08 MB of lower halves + 0.13 MB of single entries against 6.31 MB (SURVEY.md 8d's formula).
template <int N, int M>
struct HalfShape {
  static constexpr int HP = M / 2, LD = M / 2 + 1, TOTAL = (M / 2) * N, R = ((M / 2) * N + BSF - 1) / BSF;   // f64x2 words of a unit
};
static_assert(38 % 2 == 0, "row pairs");

// kinds: 10 = x-columns c0, c0 + 1 (units: f_xx slab c0 | f_xx slab c0 + 1 | f_ux slabs c0, c0 + 1); 11 = cn u-columns from c0,
// cn even (units: f_uu slabs in pairs)
template <int N, int M>
__global__ __launch_bounds__(BSF, BWD_WAVES_PER_SIMD) void bwd_contract_half(BwdParams p, int64_t t) {
  int b, jb;
  {
    const int njobs = (int)gridDim.x, B = (int)gridDim.y;
    const int lin = blockIdx.y * njobs + blockIdx.x;
    if ((B & 7) == 0) {
      const int xcd = lin & 7, k = lin >> 3;
      b = xcd + 8 * (k / njobs);
      jb = k % njobs;
    } else { b = blockIdx.y; jb = blockIdx.x; }
    b += p.b0;
  }
  if (p.status[b] != 0) return;
  const BwdJob job = p.jobs_half[jb];
  constexpr int n = N, m = M;
  static_assert(N == 2 * M && M % 2 == 0, "lower halves of M rows, in row pairs");
  const int64_t T = p.d.T;
  const int tid = threadIdx.x;
  const int kind = job.kind, c0 = job.c0, cn = job.cn;
  const bool m1 = p.half_mode == 2;            // analytic mode-1 tensors: every column read, no upper-half entries, f_uu all zeros
  const int64_t bt = (int64_t)b * T + t;
  const double* Vx = p.ws_V + (int64_t)b * (n + n * n);
  double* C = p.c_accumulate == 2 ? p.ws_D + (int64_t)b * (n * n + m * n + m * m)
                                  : p.ws_Q + (int64_t)b * (n + m + n * n + m * n + m * m) + n + m;
  double* Cxx = C;
  double* Cux = Cxx + n * n;
  double* Cuu = Cux + m * n;
  using HS = HalfShape<N, M>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* s_v = smem;                          // n
  double* s_out = s_v + n;                     // units x N
  double* s_p0 = s_out + 3 * N;                // HS::LD * N
  double* s_p1 = s_p0 + HS::LD * N;
  const int U = kind == 10 ? 3 : cn / 2;
  // unit u: its N columns are slab / column pairs (sl, j): which tensor, which slab, the direction index of the column
  auto unit_base = [&](int u) -> const double* {
    if (kind == 10) return u < 2 ? p.fxx + (bt * n + c0 + u) * (int64_t)n * n : p.fux + (bt * n + c0) * (int64_t)n * m;
    return p.fuu + (bt * m + c0 + 2 * u) * (int64_t)n * m;
  };
  // column jj of unit u: slab index (the first direction of the pair), column index within its tensor, and whether the column is
  // read at all (symmetric tensors: j >= slab)
  auto col_info = [&](int u, int jj, int& slab, int& j, bool& xcol) -> bool {
    if (kind == 10) {
      if (u < 2) { slab = c0 + u; j = jj; xcol = true; return m1 || j >= slab; }          // f_xx(:, j, slab)
      slab = c0 + (jj >= m ? 1 : 0); j = jj >= m ? jj - m : jj; xcol = false; return true;   // f_ux(:, j, slab)
    }
    slab = c0 + 2 * u + (jj >= m ? 1 : 0); j = jj >= m ? jj - m : jj; xcol = false; return m1 || j >= slab;   // f_uu(:, j, slab)
  };
  f64x2 buf0[HS::R], buf1[HS::R], buf2[HS::R], buf3[HS::R];
  double top0[2], top1[2], top2[2], top3[2];   // lane jj < N: the (at most) two entries of its column's upper half
  auto issue = [&](int u, f64x2 (&buf)[HS::R], double (&top)[2]) {
    const f64x2* base = reinterpret_cast<const f64x2*>(unit_base(u));
#pragma unroll
    for (int r = 0; r < HS::R; ++r) {
      const int f = tid + r * BSF;
      bool need = r < HS::R - 1 || f < HS::TOTAL;
      const int jj = f / HS::HP, ip = f - jj * HS::HP;
      int slab, j; bool xcol;
      if (need) need = col_info(u, jj < n ? jj : n - 1, slab, j, xcol) && !(m1 && kind == 11);
      if (need) buf[r] = BWD_NT ? __builtin_nontemporal_load(&base[jj * (N / 2) + HS::HP + ip]) : base[jj * (N / 2) + HS::HP + ip];
      else buf[r] = f64x2{0.0, 0.0};
    }
    top[0] = 0.0; top[1] = 0.0;
    if (tid < n) {
      int slab, j; bool xcol;
      if (col_info(u, tid, slab, j, xcol) && kind == 10 && !m1) {   // f_uu: its directions are controls, the upper half is all zeros
        const double* colp = unit_base(u) + (int64_t)tid * n;
        top[0] = colp[slab % M];                                     // the slab's direction is an x direction (f_xx and f_ux)
        if (xcol && (j % M) != (slab % M)) top[1] = colp[j % M];
      }
    }
  };
  if (U > 0) issue(0, buf0, top0);
  if (U > 1) issue(1, buf1, top1);
  if (U > 2) issue(2, buf2, top2);
  // this lane's output entry (U * N <= BSF: one per lane) and what the workspace holds there (c_accumulate: the dense terms K5
  // left), requested now so that the read-modify-write at the end does not wait for a round trip of its own
  double* dst = nullptr;
  double* dm = nullptr;
  double old_d = 0.0, old_m = 0.0;
  static_assert(3 * N <= BSF, "one output entry per lane");
  if (tid < U * N) {
    const int u = tid / N, jj = tid - u * N;
    int slab, j; bool xcol;
    if (col_info(u, jj, slab, j, xcol)) {                  // (else: the mirror image, written by the job of the other column)
      if (kind == 10) {
        if (u < 2) { dst = Cxx + j + slab * n; if (j > slab && !m1) dm = Cxx + slab + j * n; }
        else dst = Cux + j + slab * m;
      } else { dst = Cuu + j + slab * m; if (j > slab && !m1) dm = Cuu + slab + j * m; }
      if (p.c_accumulate == 1) { old_d = *dst; if (dm) old_m = *dm; }
    }
  }
  for (int i = tid; i < n; i += BSF) s_v[i] = Vx[i];
  __syncthreads();
  auto step = [&](int u, f64x2 (&buf)[HS::R], double (&top)[2]) {
    double* sp = (u & 1) ? s_p1 : s_p0;
#pragma unroll
    for (int r = 0; r < HS::R; ++r) {
      const int f = tid + r * BSF;
      if (r < HS::R - 1 || f < HS::TOTAL) {
        const int jj = f / HS::HP, ip = f - jj * HS::HP;
        const f64x2 vv = *reinterpret_cast<const f64x2*>(s_v + M + 2 * ip);
        sp[jj * HS::LD + ip] = vv.x * buf[r].x + vv.y * buf[r].y;
      }
    }
    __syncthreads();
    if (tid < n) {
      int slab, j; bool xcol;
      const bool need = col_info(u, tid, slab, j, xcol);
      double sacc = 0.0;
      if (need && kind == 10 && !m1) {
        // the upper half's partials, in bwd_contract's order: row pair (2 ip, 2 ip + 1), ip ascending; all others are zeros
        const int ra = slab % M, rb = xcol ? j % M : ra;
        const int lo = ra < rb ? ra : rb, hi = ra < rb ? rb : ra;
        const double tlo = ra <= rb ? top[0] : top[1], thi = ra <= rb ? (ra == rb ? 0.0 : top[1]) : top[0];
        if (lo == hi) {
          const int e = lo & ~1;
          const f64x2 vv = *reinterpret_cast<const f64x2*>(s_v + e);
          const double tx = (lo & 1) ? 0.0 : tlo, ty = (lo & 1) ? tlo : 0.0;
          sacc += vv.x * tx + vv.y * ty;
        } else if ((lo >> 1) == (hi >> 1)) {
          const f64x2 vv = *reinterpret_cast<const f64x2*>(s_v + lo);      // lo even, hi = lo + 1
          sacc += vv.x * tlo + vv.y * thi;
        } else {
          const f64x2 va = *reinterpret_cast<const f64x2*>(s_v + (lo & ~1));
          const f64x2 vb = *reinterpret_cast<const f64x2*>(s_v + (hi & ~1));
          sacc += va.x * ((lo & 1) ? 0.0 : tlo) + va.y * ((lo & 1) ? tlo : 0.0);
          sacc += vb.x * ((hi & 1) ? 0.0 : thi) + vb.y * ((hi & 1) ? thi : 0.0);
        }
      }
      const double* pj = sp + tid * HS::LD;
#pragma unroll
      for (int k = 0; k < HS::HP; ++k) sacc += pj[k];
      s_out[u * N + tid] = sacc;
    }
  };
  // three units at most per job: all in flight from the start
  if (U > 0) step(0, buf0, top0);
  if (U > 1) step(1, buf1, top1);
  if (U > 2) step(2, buf2, top2);
  (void)buf3; (void)top3;
  __syncthreads();
  if (dst) {
    const double v = s_out[tid];
    *dst = old_d + v;                                      // (old_d = 0 unless c_accumulate == 1: 0 + v == v)
    if (dm) *dm = old_m + v;
  }
}

// D = [f_x f_u]^T V_xx [f_x f_u] (blocks xx, ux, uu) for timestep td, with V_xx already in s_VW[0 .. N*N).
// LDS: s_VW (N*(N+M) doubles: V_xx, then W = V_xx F), s_F (N*(N+M) doubles).  All BSR lanes take part.
// This is the one genuinely dense product of the step (3.3 MFLOP per instance and step); it runs on the FP64 matrix
// cores: v_mfma_f64_16x16x4_f64, one 16 x 16 output tile per wave at a time, operands read from LDS.
// Lane l feeds A[row = l & 15][k = l >> 4] and B[k = l >> 4][col = l & 15]; result register r of lane l is
// D[row = (l >> 4) + 4 r][col = l & 15].  Rows / columns beyond the matrix are fed zeros.
typedef double f64x4 __attribute__((ext_vector_type(4)));

// ... and, while F = [f_x | f_u] of step td is in LDS, the dense part of that step's Q_x | Q_u: g = F^T V_x
// (ddp_bwd.ipp:62,67), left in the Q_x | Q_u slots of the instance's workspace for the K4 launch of step td
template <int N, int M>
__device__ __forceinline__ void dense_product(const BwdParams& p, int b, int64_t td, double* s_VW, double* s_F, const double* s_vxn) {
  constexpr int n = N, m = M, NM = N + M;
  const int tid = threadIdx.x;
  const int64_t bt = (int64_t)b * p.d.T + td;
  const double* fx = p.fx + bt * n * n;
  const double* fu = p.fu + bt * n * m;
  double* D = p.ws_D + (int64_t)b * (n * n + m * n + m * m);
  double* Dxx = D;
  double* Dux = Dxx + n * n;
  double* Duu = Dux + m * n;
  // F = [f_x | f_u], column-major, leading dimension N
  {
    const f64x2* a = reinterpret_cast<const f64x2*>(fx);
    const f64x2* c = reinterpret_cast<const f64x2*>(fu);
    f64x2* d = reinterpret_cast<f64x2*>(s_F);
    for (int i = tid; i < n * n / 2; i += BSR) d[i] = a[i];
    for (int i = tid; i < n * m / 2; i += BSR) d[n * n / 2 + i] = c[i];
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, l4 = lane >> 4;
  constexpr int NW = BSR / 64;
  constexpr int KS = N / 4;                                   // 19 k-steps of 4
  static_assert(N % 4 == 0, "k-steps of 4");
  constexpr int MT_V = (N + 15) / 16, NT = (NM + 15) / 16;    // 5 row tiles of V, 8 column tiles of F
  // W = V_xx F  (76 x 114): tiles (mt, nt); results kept in registers until every wave is done reading V_xx
  constexpr int W_TILES = MT_V * NT, W_PER_WAVE = (W_TILES + NW - 1) / NW;   // 40 tiles, 5 per wave
  f64x4 wacc[W_PER_WAVE];
#pragma unroll
  for (int it = 0; it < W_PER_WAVE; ++it) {
    const int tile = wave + it * NW;
    const int mt = tile % MT_V, nt = (tile / MT_V) % NT;
    const int row = 16 * mt + l15, col = 16 * nt + l15;
    const bool rok = row < n, cok = col < NM;
    const double* va = s_VW + (rok ? row : 0) + l4 * n;       // V(row, 4 s + l4)
    const double* fb = s_F + (cok ? col : 0) * n + l4;        // F(4 s + l4, col)
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const double av = rok ? va[4 * s * n] : 0.0;
      const double bv = cok ? fb[4 * s] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    wacc[it] = acc;
  }
  __syncthreads();   // every wave is done reading V_xx: W may overwrite it
#pragma unroll
  for (int it = 0; it < W_PER_WAVE; ++it) {
    const int tile = wave + it * NW;
    if (tile < W_TILES) {
      const int mt = tile % MT_V, nt = tile / MT_V;
      const int col = 16 * nt + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * mt + l4 + 4 * r;
        if (row < n && col < NM) s_VW[row + col * n] = wacc[it][r];
      }
    }
  }
  __syncthreads();
  // D(j, c) = sum_k F(k, j) W(k, c): tiles (jt, ct) over 114 x 114, skipping those entirely inside the unused
  // (j < N, c >= N) block
  constexpr int JT = NT, CT = NT;
  for (int tile = wave; tile < JT * CT; tile += NW) {
    const int jt = tile % JT, ct = tile / JT;
    if (16 * jt + 15 < n && 16 * ct >= n) continue;           // wave-uniform
    const int j = 16 * jt + l15, c = 16 * ct + l15;
    const bool jok = j < NM, cok = c < NM;
    const double* fa = s_F + (jok ? j : 0) * n + l4;          // F(4 s + l4, j)  = A(j, k)
    const double* wb = s_VW + (cok ? c : 0) * n + l4;         // W(4 s + l4, c)  = B(k, c)
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const double av = jok ? fa[4 * s] : 0.0;
      const double bv = cok ? wb[4 * s] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jr = 16 * jt + l4 + 4 * r;                    // result row
      if (jr >= NM || c >= NM) continue;
      if (jr < n) { if (c < n) Dxx[jr + c * n] = acc[r]; }
      else if (c < n) Dux[(jr - n) + c * m] = acc[r];
      else Duu[(jr - n) + (c - n) * m] = acc[r];
    }
  }
  // g = F^T V_x, one wave per column (two-term partials, then a shuffle tree)
  double* g = p.ws_Q + (int64_t)b * (n + m + n * n + m * n + m * m);
  for (int c = wave; c < NM; c += NW) {
    const double* col = s_F + c * n;
    double sacc = 0.0;
    if (lane < n / 2) {
      const f64x2 a = *reinterpret_cast<const f64x2*>(col + 2 * lane);
      const f64x2 vv = *reinterpret_cast<const f64x2*>(s_vxn + 2 * lane);
      sacc = a.x * vv.x + a.y * vv.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sacc += __shfl_down(sacc, off, 64);
    if (lane == 0) g[c] = sacc;
  }
}

// D for the first step to be processed (t = T-1), from V_xx = lfxx (ddp_bwd.ipp:27)
template <int N, int M>
__global__ __launch_bounds__(BSR) void bwd_dense0(BwdParams p) {
  const int b = p.b0 + blockIdx.x;
  if (p.status[b] != 0) return;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* s_VW = smem;
  double* s_F = smem + N * (N + M);
  const double* Vxx = p.ws_V + (int64_t)b * (N + N * N) + N;
  __shared__ __attribute__((aligned(16))) double s_vxn[N];
  for (int i = threadIdx.x; i < N * N; i += BSR) s_VW[i] = Vxx[i];
  for (int i = threadIdx.x; i < N; i += BSR) s_vxn[i] = Vxx[i - N];          // V_x = lfx^T (ddp_bwd.ipp:28)
  // (dense_product starts with a barrier after loading F)
  dense_product<N, M>(p, b, p.d.T - 1, s_VW, s_F, s_vxn);
}

template <int N, int M>
__global__ __launch_bounds__(BSR) void bwd_riccati(BwdParams p, int64_t t) {
  const int b = p.b0 + blockIdx.x;
  if (p.status[b] != 0) return;
  constexpr int n = N, m = M;
  const int nx = (int)p.d.nx;        // N + 1 with a free-flyer root
  const int64_t T = p.d.T;
  const int tid = threadIdx.x;
  const int64_t bt = (int64_t)b * T + t;
  const int e = (int)p.ne[t];
  const int64_t Eo = p.Epre[t], Etot = p.d.Etot;
  const double mu = p.mu[b];
  const bool tens = p.has_tensors != 0;

  double* Vx = p.ws_V + (int64_t)b * (n + n * n);
  double* Vxx = Vx + n;
  const double* C = p.ws_Q + (int64_t)b * (n + m + n * n + m * n + m * m) + n + m;   // K3's contraction
  const double* Cxx = C;
  const double* Cux = Cxx + n * n;
  const double* Cuu = Cux + m * n;
  const double* D = p.ws_D + (int64_t)b * (n * n + m * n + m * m);                  // dense f^T V_xx f
  const double* Dxx = D;
  const double* Dux = Dxx + n * n;
  const double* Duu = Dux + m * n;
  const double* eqv = p.eq_val + (int64_t)b * Etot + Eo;
  const double* eqx = p.eq_x + ((int64_t)b * Etot + Eo) * n;
  const double* equ = p.eq_u + ((int64_t)b * Etot + Eo) * m;
  const double* pe = p.mult_val + (int64_t)b * Etot + Eo;
  const double* pex = p.mult_jac + ((int64_t)b * Etot + Eo) * n;
  const double* eq_xx = p.eq_xx + ((int64_t)b * Etot + Eo) * n * n;
  const double* eq_ux = p.eq_ux + ((int64_t)b * Etot + Eo) * m * n;
  const double* eq_uu = p.eq_uu + ((int64_t)b * Etot + Eo) * m * m;

  constexpr int lda = M | 1, ldr = M | 1, NR = N + 1, NM = N + M;
  constexpr int TJ = 6, AQ = (M + TJ - 1) / TJ;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* s_VW = smem;                         // N*NM: new V_xx, later W (dense tail)
  double* s_F = smem + N * NM;                 // N*NM: F of step t-1 (dense tail); before that, the arrays below
  double* A = s_F;                             // lda*M   Q_uu + reg I -> Cholesky factor (lower)
  double* R = A + lda * M;                     // ldr*NR  [k | K]
  double* S = R + ldr * NR;                    // ldr*N   Q_ux
  double* s_q = S + ldr * N;                   // NM      Q_x | Q_u
  double* s_tmp = s_q + NM;                    // emax    pe + mu eq   (ddp_bwd.ipp:46)
  static_assert(lda * M + ldr * NR + ldr * N + NM + N + 64 <= N * NM, "phase-A arrays must fit the F region");

  // entries of Q in the reference's order of terms (ddp_bwd.ipp:70-87): l, f^T V_xx f, multiplier terms, multiplier
  // tensors, V_x-contracted tensors
  auto q_uu = [&](int i, int j) -> double {
    double acc = p.luu[bt * m * m + i + j * m];
    acc += Duu[i + j * m];
    if (e > 0) {
      double s1 = 0.0;
      for (int k = 0; k < e; ++k) s1 += equ[k + i * e] * equ[k + j * e];
      acc += s1 * mu;                                                          // :79
      if (tens) { double s3 = 0.0; for (int k = 0; k < e; ++k) s3 += s_tmp[k] * eq_uu[k + (i + j * m) * e]; acc += s3; }   // :80
    }
    if (tens) acc += Cuu[i + j * m];                                           // :81
    return acc;
  };
  auto q_ux = [&](int i, int j) -> double {
    double acc = p.lux[bt * m * n + i + j * m];
    acc += Dux[i + j * m];
    if (e > 0) {
      double s1 = 0.0;
      for (int k = 0; k < e; ++k) s1 += equ[k + i * e] * (pex[k + j * e] + mu * eqx[k + j * e]);   // :85
      acc += s1;
      if (tens) { double s3 = 0.0; for (int k = 0; k < e; ++k) s3 += s_tmp[k] * eq_ux[k + (i + j * m) * e]; acc += s3; }   // :86
    }
    if (tens) acc += Cux[i + j * m];                                           // :87
    return acc;
  };
  auto q_xx = [&](int i, int j) -> double {
    double acc = p.lxx[bt * n * n + i + j * n];
    acc += Dxx[i + j * n];
    if (e > 0) {
      double s1 = 0.0, s2 = 0.0;
      for (int k = 0; k < e; ++k) {
        s1 += eqx[k + i * e] * (pex[k + j * e] + mu * eqx[k + j * e]);         // :72
        s2 += pex[k + i * e] * eqx[k + j * e];                                 // :73
      }
      acc += s1;
      acc += s2;
      if (tens) { double s3 = 0.0; for (int k = 0; k < e; ++k) s3 += s_tmp[k] * eq_xx[k + (i + j * n) * e]; acc += s3; }   // :74
    }
    if (tens) acc += Cxx[i + j * n];                                           // :75
    return acc;
  };

  __shared__ __attribute__((aligned(16))) double s_vxn[N];   // the new V_x, for the matvec of the next step (dense tail)
  for (int i = tid; i < e; i += BSR) s_tmp[i] = pe[i] + mu * eqv[i];
  __syncthreads();
  // Q_x, Q_u (:61-68): the dense part f^T V_x was formed by the previous launch while f was in LDS
  if (tid < NM) {
    const int c = tid;
    const double* g = p.ws_Q + (int64_t)b * (n + m + n * n + m * n + m * m);
    double acc = c < n ? p.lx[bt * n + c] : p.lu[bt * m + (c - n)];
    acc += g[c];
    if (e > 0) {
      double s1 = 0.0, s2 = 0.0;
      if (c < n) { for (int k = 0; k < e; ++k) { s1 += eqx[k + c * e] * s_tmp[k]; s2 += pex[k + c * e] * eqv[k]; } acc += s1; acc += s2; }
      else { for (int k = 0; k < e; ++k) s1 += equ[k + (c - n) * e] * s_tmp[k]; acc += s1; }
    }
    s_q[c] = acc;
  }
  for (int idx = tid; idx < m * n; idx += BSR) S[idx % m + (idx / m) * ldr] = q_ux(idx % m, idx / m);
  __syncthreads();

  const double reg = p.reg[b];
  // right-hand sides [-Q_u | -Q_ux] (:135-136): lane 256 + c owns the whole column c in registers (waves 4-5), so the
  // substitutions need no cross-lane traffic at all; waves 0-3 carry the factorisation
  const int rc = tid - 256;
  const bool rhs_lane = tid >= 256 && rc < NR;
  double r[M];
  if (rhs_lane) {
#pragma unroll
    for (int l = 0; l < m; ++l) r[l] = -(rc == 0 ? s_q[n + l] : S[l + (rc - 1) * ldr]);
  }
  // trailing matrix of the factorisation in registers: lane (ti, tj) owns row ti, columns tj, tj+6, ... <= ti
  const int ti = tid % M, tj = tid / M;
  const bool a_lane = tj < TJ;
  double a[AQ];
#pragma unroll
  for (int q = 0; q < AQ; ++q) {
    const int j = tj + TJ * q;
    a[q] = (a_lane && j <= ti) ? q_uu(ti, j) + (ti == j ? reg : 0.0) : 0.0;           // :104
  }
  if (a_lane && tj == 0) A[ti] = a[0];       // raw column 0
  __syncthreads();

  // Cholesky (lower triangle only; fail <=> pivot <= 0, :105) with the forward substitution fused in: column k of L is
  // final after step k, so y_k = r_k / L_kk and r_l -= L_lk y_k (l > k) ride along with the trailing update.  Per
  // entry the updates arrive in ascending k: the order of Eigen's unblocked LLT and of its row-wise substitution.
  bool failed = false;
#pragma unroll
  for (int k = 0; k < m; ++k) {
    const double piv = A[k + k * lda];
    if (piv <= 0.0) { failed = true; break; }
    const double dk = sqrt(piv);
    if (a_lane && tj == 0 && ti > k) A[ti + k * lda] = A[ti + k * lda] / dk;
    if (rhs_lane) r[k] = r[k] / dk;
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
      if (k1 < m && tj == k1 % TJ && ti >= k1) A[ti + k1 * lda] = a[k1 / TJ];   // raw column k+1, final after this update
    }
    if (rhs_lane) {
#pragma unroll
      for (int l = k + 1; l < m; ++l) r[l] -= Lk[l] * r[k];
    }
    if (tid == k) A[k + k * lda] = dk;
    __syncthreads();
  }
  if (failed) {
    if (tid == 0) {
      double rg = p.reg[b], mu2 = p.mu[b];
      if (rg < mu2) rg = mu2;      // :106-108
      mu2 *= 2;                    // :109
      rg *= 2;                     // :110
      p.reg[b] = rg;
      p.mu[b] = mu2;
      p.status[b] = 1;
      p.restarts[b] += 1;
    }
    return;
  }
  // back substitution L^T x = y, column oriented, entirely inside each right-hand-side lane
  if (rhs_lane) {
#pragma unroll
    for (int k = m - 1; k >= 0; --k) {
      r[k] = r[k] / A[k + k * lda];
#pragma unroll
      for (int i = 0; i < k; ++i) r[i] -= A[k + i * lda] * r[k];
      __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting all 703 LDS reads (register pressure)
    }
  }

  double* fbo = p.fb_origin + bt * nx;
  const double* xt = p.x + ((int64_t)b * (T + 1) + t) * nx;
  for (int i = tid; i < nx; i += BSR) fbo[i] = xt[i];                        // :134
  if (rhs_lane) {
    double* dst = rc == 0 ? p.fb_val + bt * m : p.fb_jac + bt * m * n + (rc - 1) * m;
#pragma unroll
    for (int l = 0; l < m; ++l) { dst[l] = r[l]; R[l + rc * ldr] = r[l]; }
  }
  __syncthreads();

  // V_x = Q_x + Q_ux^T k (:142-143);  V_xx = Q_xx + Q_ux^T K (:145-146), 1 x 4 register tiles; the new V_xx also
  // goes to LDS for the dense product of the next step
  for (int i = tid; i < n; i += BSR) {
    double s = 0.0;
#pragma unroll 2
    for (int l = 0; l < m; ++l) s += S[l + i * ldr] * R[l];
    const double v = s_q[i] + s;
    Vx[i] = v;
    s_vxn[i] = v;
    if (p.vx_trace) p.vx_trace[bt * n + i] = v;
  }
  for (int idx = tid; idx < n * (n / 4); idx += BSR) {
    const int i = idx % n, j0 = (idx / n) * 4;
    const double* si = S + i * ldr;
    const double* k0 = R + (j0 + 1) * ldr;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 2
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
      const double v = q_xx(i, j0 + q) + sv4[q];
      Vxx[o] = v;
      s_VW[o] = v;
      if (p.vxx_trace) p.vxx_trace[bt * n * n + o] = v;
    }
  }
  if (t == 0) {
    if (tid == 0) p.status[b] = 2;                                           // :149-151
    return;
  }
  __syncthreads();   // V_xx complete in LDS; A, R, S, Y are dead: their space becomes F
  dense_product<N, M>(p, b, t - 1, s_VW, s_F, s_vxn);
}
