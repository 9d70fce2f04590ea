// bwd_v2.h -- the backward step of the Talos shape as three kernels (included by bwd.hip after bwd_split.h):
//
//   K5  bwd_dense2<N, M>    grid (column blocks, instances).  P = l + [f_x f_u]^T V_xx [f_x f_u] + multiplier terms: every
//       term of Q (ddp_bwd.ipp:61-87) except the V_x-contracted dynamics tensors, in the reference's order of terms;
//       the dense product (3.3 MFLOP per instance and step) on the FP64 matrix cores, spread over 8 workgroups per
//       instance instead of riding at the tail of the one Riccati workgroup.
//   K3  bwd_contract<N, M>  (bwd_split.h) streams the tensors; its epilogue finishes Q = P + C (the tensor term comes
//       last in the reference as well, :75,:81,:87).
//   K4' bwd_gains2<N, M>    grid (instances), 256 lanes.  LLT of Q_uu + reg I in ONE wave -- lane = row, the row in
//       registers, the pivot column broadcast through LDS within the wave: no workgroup barrier inside the factorisation
//       (the round-1 kernel took 76) -- then the 77 right-hand sides [-Q_u | -Q_ux] one per lane against L in LDS, V_x,
//       and V_xx = Q_xx + Q_ux^T K on the matrix cores.
//
// Per instance the chain is K4'(t+1) -> K5(t) -> K3(t) -> K4'(t).  Same arithmetic conventions as bwd_split.h: no
// symmetrisation, lower triangle only, fail <=> pivot <= 0, per-entry updates in ascending k.
#pragma once

// development: -DBWD_STAMPS builds in-kernel phase stamps (s_memrealtime, 100 MHz) of instance 0, read by tools/bwd_stamps.py
#ifdef BWD_STAMPS
__device__ unsigned long long g_bwd_stamps[32];
#define STAMP(i) do { if (tid == 0 && b == 0 && blockIdx.x == 0) g_bwd_stamps[i] = wall_clock64(); } while (0)
#define STAMP_T(i, T_) do { if (tid == (T_) && b == 0 && blockIdx.x == 0) g_bwd_stamps[i] = wall_clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMP_T(i, T_) do { } while (0)
#endif

constexpr int BS5 = 512;   // workgroup size of K5
constexpr int BS4 = 512;   // workgroup size of K4' (wave 0: LLT, waves 1-2: right-hand sides, all eight: loads and the V update)
constexpr int CB5 = 32;    // columns of F = [f_x | f_u] per K5 workgroup (two MFMA tiles): 4 workgroups per instance

// workgroup barrier for LDS traffic only: the LDS operations of the wave have been performed, then s_barrier.  Unlike
// __syncthreads() it does not wait for outstanding global loads (vmcnt), which is the point where a kernel keeps loads in flight
// across a phase boundary; the compiler still waits for the registers a store needs.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N, int M>
__global__ __launch_bounds__(BS5) void bwd_dense2(BwdParams p, int64_t t) {
  constexpr int n = N, m = M, NM = N + M;
  const int b = p.b0 + blockIdx.y;
  if (p.status[b] != 0) return;
  const int c0 = blockIdx.x * CB5;                 // first column of this block (0 .. NM-1)
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  constexpr int NW = BS5 / 64;
  const int64_t T = p.d.T;
  const int64_t bt = (int64_t)b * T + t;
  const int e = (int)p.ne[t];
  const int64_t Eo = p.Epre[t], Etot = p.d.Etot;
  const double mu = p.mu[b];
  const bool tens = p.has_tensors != 0;

  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* s_V = smem;                // N * N      V_xx(t+1), column-major
  double* s_F = s_V + N * N;         // N * NM     F = [f_x | f_u] of step t, column-major ld N
  double* s_W = s_V;                 // N * CB5    W = V_xx F(:, block): written over V_xx once every tile of the first product has
                                     //            left its accumulators (116 KB instead of 136: one K3 workgroup fits beside this one)
  double* s_vx = s_F + N * NM;       // N          V_x(t+1)
  double* s_tmp = s_vx + N;          // emax       pe + mu eq   (ddp_bwd.ipp:46)

  const double* Vx = p.ws_V + (int64_t)b * (n + n * n);
  const double* Vxx = Vx + n;
  STAMP(0);
  // Two-phase staging.  The first product W = V_xx F(:, block) only needs V_xx and the block's own columns of F: those are
  // waited for; the other columns of F (cold: written by the linearisation) are requested at the same time, stay in flight in
  // registers through the first product and are parked in front of the second (D = F^T W needs all of F).
  constexpr int H2 = N / 2;                        // a column of F in 16-byte words
  static_assert(N % 2 == 0, "columns are moved in 16-byte words");
  const int ncolA = (NM - c0) < CB5 ? (NM - c0) : CB5;
  const f64x2* fx2 = reinterpret_cast<const f64x2*>(p.fx + bt * n * n);
  const f64x2* fu2 = reinterpret_cast<const f64x2*>(p.fu + bt * n * m);
  auto fsrc = [&](int col, int k2) -> const f64x2* { return col < n ? fx2 + col * H2 + k2 : fu2 + (col - n) * H2 + k2; };
  constexpr int NRB = ((NM - 1) * H2 + BS5 - 1) / BS5;          // words per lane of the second phase (at most NM - 1 columns)
  f64x2 rb[NRB];
  double lq;
  {
    const f64x2* a = reinterpret_cast<const f64x2*>(Vxx);
    f64x2* d = reinterpret_cast<f64x2*>(s_V);
    f64x2* f = reinterpret_cast<f64x2*>(s_F);
    constexpr int NRV = (N * N / 2 + BS5 - 1) / BS5, NRA = (CB5 * H2 + BS5 - 1) / BS5;
    f64x2 rv[NRV], ra[NRA];
    const int nA = ncolA * H2, nB = (NM - ncolA) * H2;
#pragma unroll
    for (int r = 0; r < NRV; ++r) { const int i = tid + r * BS5; rv[r] = a[i < n * n / 2 ? i : n * n / 2 - 1]; }
#pragma unroll
    for (int r = 0; r < NRA; ++r) { int e = tid + r * BS5; e = e < nA ? e : nA - 1; ra[r] = *fsrc(c0 + e / H2, e % H2); }
    const double vxv = Vx[tid < n ? tid : n - 1];
    {                                              // the gradient term of the last wave's Q_x | Q_u columns: requested ahead of the
      const int cq = c0 + lane < NM ? c0 + lane : NM - 1;                  // second phase (a load behind it would wait for all of it)
      lq = cq < n ? p.lx[bt * n + cq] : p.lu[bt * m + (cq - n)];
    }
#pragma unroll
    for (int r = 0; r < NRB; ++r) {
      int e = tid + r * BS5;
      e = e < nB ? e : nB - 1;
      const int ci = e / H2;
      rb[r] = *fsrc(ci < c0 ? ci : ci + ncolA, e % H2);
    }
#pragma unroll
    // (branch-free parking: lanes past the end store the last word again; under a branch the optimiser sinks the load into it
    // and waits for everything in flight)
    for (int r = 0; r < NRV; ++r) { int i = tid + r * BS5; i = i < n * n / 2 ? i : n * n / 2 - 1; d[i] = rv[r]; }
#pragma unroll
    for (int r = 0; r < NRA; ++r) { int e = tid + r * BS5; e = e < nA ? e : nA - 1; f[(c0 + e / H2) * H2 + e % H2] = ra[r]; }
    s_vx[tid < n ? tid : n - 1] = vxv;
  }
  const double* eqv = p.eq_val + (int64_t)b * Etot + Eo;
  const double* eqx = p.eq_x + ((int64_t)b * Etot + Eo) * n;
  const double* equ = p.eq_u + ((int64_t)b * Etot + Eo) * m;
  const double* pe = p.mult_val + (int64_t)b * Etot + Eo;
  const double* pex = p.mult_jac + ((int64_t)b * Etot + Eo) * n;
  const double* eq_xx = p.eq_xx + ((int64_t)b * Etot + Eo) * n * n;
  const double* eq_ux = p.eq_ux + ((int64_t)b * Etot + Eo) * m * n;
  const double* eq_uu = p.eq_uu + ((int64_t)b * Etot + Eo) * m * m;
  for (int i = tid; i < e; i += BS5) s_tmp[i] = pe[i] + mu * eqv[i];
  lds_barrier();                                     // (LDS only: no vmcnt(0) -- the second phase of F stays in flight)
  STAMP(1);
  double* Q = p.ws_Q + (int64_t)b * (n + m + n * n + m * n + m * m);
  // Q_x | Q_u of this block's columns (:61-68): l + F^T V_x + multiplier terms -- one lane per column in the last wave (it has
  // a single tile of the product below), next to the other waves' matrix work
  if (wave == NW - 1 && lane < CB5 && c0 + lane < NM) {
    const int c = c0 + lane;
    const double* col = s_F + c * n;
    double sacc = 0.0;
#pragma unroll 4
    for (int k = 0; k < n; ++k) sacc += col[k] * s_vx[k];
    double acc = lq;
    acc += sacc;
    if (e > 0) {
      double s1 = 0.0, s2 = 0.0;
      if (c < n) { for (int k = 0; k < e; ++k) { s1 += eqx[k + c * e] * s_tmp[k]; s2 += pex[k + c * e] * eqv[k]; } acc += s1; acc += s2; }
      else { for (int k = 0; k < e; ++k) s1 += equ[k + (c - n) * e] * s_tmp[k]; acc += s1; }
    }
    Q[c] = acc;
  }

  constexpr int KS = N / 4;
  static_assert(N % 4 == 0, "k-steps of 4");
  // W(:, block) = V_xx F(:, block): row tiles of 16 over the waves.  A(row = i, k) = V(i, k), B(k, col = c) = F(k, c0 + c)
  constexpr int MT = (N + 15) / 16, CT = CB5 / 16, TPW1 = (MT * CT + NW - 1) / NW;
  f64x4 wacc[TPW1];
#pragma unroll
  for (int it_ = 0; it_ < TPW1; ++it_) {
    const int tile = wave + it_ * NW;
    wacc[it_] = f64x4{0.0, 0.0, 0.0, 0.0};
    if (tile >= MT * CT) continue;
    const int mt = tile % MT, ct = tile / MT;
    const int row = 16 * mt + l15, col = c0 + 16 * ct + l15;
    const bool rok = row < n, cok = col < NM;
    const double* va = s_V + (rok ? row : 0) + l4 * n;
    const double* fb = s_F + (cok ? col : 0) * n + l4;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const double av = rok ? va[4 * s * n] : 0.0;
      const double bv = cok ? fb[4 * s] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    wacc[it_] = acc;
  }
  lds_barrier();                                     // every wave is done reading V_xx: W takes its place
#pragma unroll
  for (int it_ = 0; it_ < TPW1; ++it_) {
    const int tile = wave + it_ * NW;
    if (tile >= MT * CT) continue;
    const int mt = tile % MT, ct = tile / MT;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 16 * mt + l4 + 4 * q;
      if (r < n) s_W[r + (16 * ct + l15) * n] = wacc[it_][q];
    }
  }
  {                                                  // the rest of F has arrived meanwhile
    f64x2* f = reinterpret_cast<f64x2*>(s_F);
    const int nB = (NM - ncolA) * H2;
#pragma unroll
    for (int r = 0; r < NRB; ++r) {
      int e = tid + r * BS5;
      e = e < nB ? e : nB - 1;
      const int ci = e / H2;
      f[(ci < c0 ? ci : ci + ncolA) * H2 + e % H2] = rb[r];
    }
  }
  lds_barrier();
  STAMP(2);

  double* Pxx = Q + n + m;
  double* Pux = Pxx + n * n;
  double* Puu = Pux + m * n;
  // D^T(c, j) = sum_k W(k, c) F(k, j): A(row = c, k) = W(k, c), B(k, col = j) = F(k, j); result (row = c = l4 + 4 q,
  // col = j = 16 jt + l15): consecutive lanes -> consecutive j, the fast index of every block of Q
  constexpr int JT = (NM + 15) / 16;
  constexpr int TPW2 = (JT * CT + NW - 1) / NW;      // tiles per wave
  double av_[TPW2][4];                               // l + D of this lane's entries, tile by tile
#pragma unroll
  for (int it_ = 0; it_ < TPW2; ++it_) {
    const int tile = wave + it_ * NW;
#pragma unroll
    for (int q = 0; q < 4; ++q) av_[it_][q] = 0.0;
    if (tile >= JT * CT) continue;
    const int jt = tile % JT, ct = tile / JT;
    const int cb = c0 + 16 * ct;                    // first column of this tile
    if (cb >= NM) continue;
    if (cb >= n && 16 * jt + 15 < n) continue;      // (j < n, c >= n): the unused x-u block (wave-uniform)
    const int j = 16 * jt + l15;
    const bool jok = j < NM;
    const double* wa = s_W + (16 * ct + l15) * n + l4;   // W(4 s + l4, c = 16 ct + l15)
    const double* fb = s_F + (jok ? j : 0) * n + l4;
    double lv[4];                                    // the cost terms of this lane's four entries, fetched ahead of the product
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = cb + l4 + 4 * q;
      lv[q] = 0.0;
      if (c < NM && jok) {
        if (c < n) lv[q] = j < n ? p.lxx[bt * n * n + j + c * n] : p.lux[bt * m * n + (j - n) + c * m];
        else if (j >= n) lv[q] = p.luu[bt * m * m + (j - n) + (c - n) * m];
      }
    }
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const double av = wa[4 * s];
      const double bv = jok ? fb[4 * s] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {                   // the reference's first two terms (ddp_bwd.ipp:70-71,77-78,83-84): l, f^T V_xx f
      double a = lv[q];
      a += acc[q];
      av_[it_][q] = a;
    }
  }
  // Multiplier terms (:72-73,:79,:85).  Round 2 formed them entry by entry from global memory (38 x 4 loads per entry of
  // Q_xx at e = 38: 175 us per step against 56 without constraints).  Now eq_x, eq_u, pe_x and tmp2 = pe_x + mu eq_x are
  // staged once in LDS over the dead V / F region (k fastest, k padded with zeros to a multiple of 4) and the products
  //   eq_x^T tmp2, pe_x^T eq_x (Q_xx),  eq_u^T tmp2 (Q_ux),  eq_u^T eq_u (Q_uu)
  // run on the FP64 matrix cores like the dense term.
  const int LDE = (e + 3) & ~3;
  double* s_EX = smem;                               // LDE x n
  double* s_EU = s_EX + LDE * n;                     // LDE x m
  double* s_PX = s_EU + LDE * m;                     // LDE x n
  double* s_T2 = s_PX + LDE * n;                     // LDE x n
  if (e > 0) {
    lds_barrier();                                   // every wave is done with W and F
    for (int idx = tid; idx < LDE * n; idx += BS5) {
      const int k = idx % LDE, c = idx / LDE;
      const double ex = k < e ? eqx[k + c * e] : 0.0, px = k < e ? pex[k + c * e] : 0.0;
      s_EX[idx] = ex; s_PX[idx] = px; s_T2[idx] = k < e ? px + mu * ex : 0.0;        // :47
    }
    for (int idx = tid; idx < LDE * m; idx += BS5) {
      const int k = idx % LDE, c = idx / LDE;
      s_EU[idx] = k < e ? equ[k + c * e] : 0.0;
    }
    __syncthreads();
  }
#pragma unroll
  for (int it_ = 0; it_ < TPW2; ++it_) {
    const int tile = wave + it_ * NW;
    if (tile >= JT * CT) continue;
    const int jt = tile % JT, ct = tile / JT;
    const int cb = c0 + 16 * ct;
    if (cb >= NM) continue;
    if (cb >= n && 16 * jt + 15 < n) continue;
    const int j = 16 * jt + l15;
    const bool jok = j < NM;
    f64x4 s1v = {0.0, 0.0, 0.0, 0.0}, s2v = {0.0, 0.0, 0.0, 0.0};
    if (e > 0) {
      // s1(j, c) = sum_k R(k, j) C(k, c), R = [eq_x | eq_u], C = tmp2 (x columns) / eq_u (u columns): A(row = c, k), B(k, col = j)
      const int ca = cb + l15;                       // this lane's A column
      const double* A1 = ca < n ? s_T2 + ca * LDE : s_EU + ((ca < NM ? ca : NM - 1) - n) * LDE;
      const double* B1 = j < n ? s_EX + j * LDE : s_EU + ((jok ? j : NM - 1) - n) * LDE;
      const double* A2 = s_EX + (ca < n ? ca : 0) * LDE;          // s2(j, c) = sum_k pe_x(k, j) eq_x(k, c): x-x entries only
      const double* B2 = s_PX + (j < n ? j : 0) * LDE;
      const bool aok = ca < NM, xx = cb < n && 16 * jt < n;       // the tile holds x-x entries (wave-uniform)
      for (int ks = 0; ks < LDE; ks += 4) {
        const double a1 = aok ? A1[ks + l4] : 0.0, b1 = jok ? B1[ks + l4] : 0.0;
        s1v = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, s1v, 0, 0, 0);
        if (xx) {
          const double a2 = ca < n ? A2[ks + l4] : 0.0, b2 = j < n ? B2[ks + l4] : 0.0;
          s2v = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, s2v, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = cb + l4 + 4 * q;                // column of Q (x column if < n, else u column)
      if (c >= NM || !jok) continue;
      // entries of Q in the reference's order of terms (ddp_bwd.ipp:70-86): l, f^T V_xx f, multiplier terms, multiplier tensors
      double a = av_[it_][q];
      if (c < n) {
        if (j < n) {                                // Q_xx(j, c)
          if (e > 0) {
            a += s1v[q];                                                              // :72
            a += s2v[q];                                                              // :73
            if (tens) { double s3 = 0.0; for (int k = 0; k < e; ++k) s3 += s_tmp[k] * eq_xx[k + (j + c * n) * e]; a += s3; }   // :74
          }
          Pxx[j + c * n] = a;
        } else {                                    // Q_ux(j - n, c)
          const int i = j - n;
          if (e > 0) {
            a += s1v[q];                                                              // :85
            if (tens) { double s3 = 0.0; for (int k = 0; k < e; ++k) s3 += s_tmp[k] * eq_ux[k + (i + c * m) * e]; a += s3; }   // :86
          }
          Pux[i + c * m] = a;
        }
      } else if (j >= n) {                          // Q_uu(j - n, c - n)
        const int i = j - n, cu = c - n;
        if (e > 0) {
          a += s1v[q] * mu;                                                           // :79
          if (tens) { double s3 = 0.0; for (int k = 0; k < e; ++k) s3 += s_tmp[k] * eq_uu[k + (i + cu * m) * e]; a += s3; }   // :80
        }
        Puu[i + cu * m] = a;
      }
    }
  }
  __syncthreads();
  STAMP(4);
}

// readlane of a double (two 32-bit halves through SGPRs)
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffull), l);
  const unsigned hi = __builtin_amdgcn_readlane((int)(u >> 32), l);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// K4'.  Compact on purpose: a fully unrolled elimination is ~50 KB of straight-line code executed once, and with one wave
// per SIMD every 64-byte line of it costs an instruction-cache miss (the unrolled first version of this kernel ran 168 us,
// the round-1 kernel 99 us: both fetch-bound).  Here every elimination is a rolled loop over the pivot whose body has
// static register indices because the running row / right-hand side is SHIFTED by one entry per step (the pivot is always
// entry 0): r[p] <- r[p+1] - L(k+1+p, k) r_k.  Entries beyond the triangle meet zero padding of L in LDS.
constexpr int PHASE = 2;    // columns per phase of the LLT and the substitutions of K4' (even)

// The column loop of K4's LLT (wave 0, lane = row), in phases of PHASE columns: inside a phase every step updates the W = M - 2 - K0
// entries of the shifted row that are still alive at the phase's first step (a rolled loop needs one width; with the full
// width everywhere, half of the 38 x 36 multiply-adds and operand reads worked on zero padding).
// after a step: a[q] = entry (lane, k+1+q) updated with columns 0 .. k (a[0] is consumed into column k+1); cj[q] = L(k+1+q, k).
template <int W, int STEPS, int M, int LP>
__device__ __forceinline__ void chol_phase(double (&a)[M], double& lik, bool& failed, int& k, const double*& cj, int lane, int m,
                                           double* sL, double* sLt, double& diag) {
#pragma unroll 1
  for (int s = 0; s < STEPS && !failed; ++s, ++k, cj += LP) {
    double Lq[W + 1];
#pragma unroll
    for (int q = 1; q <= W; ++q) Lq[q] = cj[q];
    // the critical chain: column k+1
    const double ljk1 = readlane_f64(lik, k + 1);
    const double a0n = a[1] - lik * ljk1;
    const double pivn = readlane_f64(a0n, k + 1);
    failed = !(pivn > 0.0);
    const double dkn = sqrt(pivn);
    const double likn = lane == k + 1 ? dkn : a0n / dkn;
    // the rest of the row, off the critical path
#pragma unroll
    for (int q = 1; q <= W; ++q) a[q] = a[q + 1] - lik * Lq[q];
    if (lane > k + 1 && lane < m) { sL[(k + 1) * LP + (lane - 2 - k)] = likn; sLt[lane * LP + (lane - 2 - k)] = likn; }
    diag = lane == k + 1 ? dkn : diag;           // 1 / L(j, j) is formed once per lane behind the loop, not once per column
    __builtin_amdgcn_wave_barrier();
    lik = likn;
  }
}
template <int K0, int M, int LP>
__device__ __forceinline__ void chol_phases(double (&a)[M], double& lik, bool& failed, int& k, const double*& cj, int lane, int m,
                                            double* sL, double* sLt, double& diag) {
  constexpr int TOTAL = M - 1;                                     // columns 1 .. M-1 are finished by steps k = 0 .. M-2
  constexpr int STEPS = (TOTAL - K0) >= PHASE ? PHASE : (TOTAL - K0);
  constexpr int W = (M - 2 - K0) > 0 ? (M - 2 - K0) : 0;
  chol_phase<W, STEPS, M, LP>(a, lik, failed, k, cj, lane, m, sL, sLt, diag);
  if constexpr (K0 + STEPS < TOTAL) chol_phases<K0 + STEPS, M, LP>(a, lik, failed, k, cj, lane, m, sL, sLt, diag);
}

// One triangular substitution of K4', lane = right-hand side, the running column shifted so that the pivot is r[0] (static
// register indices in a rolled loop).  The steps are grouped in phases of PHASE: inside a phase every step updates W = M - 1 - K0
// entries (those that are still alive at the phase's first step; the tail of a column of L is zero padding), the next phase
// runs on a narrower window.  794 multiply-adds and operand reads per pass instead of 38 x 37 = 1 406.
// DIR = +1: forward (columns of L ascending, y ascending); -1: backward (rows of L reversed, x descending).
// La holds the operands of the current step on entry; operands of step k+1 are fetched while step k is applied.
template <int W, int STEPS, int M, int LP, int DIR>
__device__ __forceinline__ void subst_phase(double (&r)[M], double (&La)[M - 1], double (&Lb)[M - 1], const double*& lp, double*& outp,
                                            const double*& dinvp) {
  static_assert(STEPS % 2 == 0, "unrolled by two");
#pragma unroll 1
  for (int s = 0; s < STEPS; s += 2) {
    {
      const double yk = r[0] * dinvp[0];
      outp[0] = yk;
#pragma unroll
      for (int q = 0; q < W; ++q) Lb[q] = lp[DIR * LP + q];
#pragma unroll
      for (int q = 0; q < W; ++q) r[q] = r[q + 1] - La[q] * yk;
    }
    {
      const double yk = r[0] * dinvp[DIR];
      outp[DIR] = yk;
#pragma unroll
      for (int q = 0; q < W; ++q) La[q] = lp[2 * DIR * LP + q];      // two columns / rows of zero padding behind the last
#pragma unroll
      for (int q = 0; q < W; ++q) r[q] = r[q + 1] - Lb[q] * yk;
    }
    lp += 2 * DIR * LP;
    outp += 2 * DIR;
    dinvp += 2 * DIR;
  }
}
template <int K0, int M, int LP, int DIR>
__device__ __forceinline__ void subst_phases(double (&r)[M], double (&La)[M - 1], double (&Lb)[M - 1], const double*& lp, double*& outp,
                                             const double*& dinvp) {
  constexpr int STEPS = (M - K0) >= PHASE ? PHASE : (M - K0);
  constexpr int W = (M - 1 - K0) > 0 ? (M - 1 - K0) : 1;
  subst_phase<W, STEPS, M, LP, DIR>(r, La, Lb, lp, outp, dinvp);
  if constexpr (K0 + STEPS < M) subst_phases<K0 + STEPS, M, LP, DIR>(r, La, Lb, lp, outp, dinvp);
}

template <int N, int M>
__global__ __launch_bounds__(BS4) void bwd_gains2(BwdParams p, int64_t t) {
  constexpr int n = N, m = M, NR = N + 1;
  const int nx = (int)p.d.nx;        // N + 1 with a free-flyer root
  constexpr int LD = M | 1;          // odd leading dimensions: conflict-free column walks
  constexpr int LP = M;              // a shifted column / reversed row holds at most M-1 entries; the rest stays zero (even: 16-byte aligned columns)
  const int b = p.b0 + blockIdx.x;
  if (p.status[b] != 0) return;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  constexpr int NW = BS4 / 64;
  const int64_t T = p.d.T;
  const int64_t bt = (int64_t)b * T + t;

  __shared__ __attribute__((aligned(16))) double sL[(M + 2) * LP];       // Cholesky factor (lower), column k SHIFTED to its sub-diagonal: sL[k LP + q] = L(k+1+q, k), zero beyond row M-1 (16-byte aligned walks)
  __shared__ __attribute__((aligned(16))) double sLt_[(M + 2) * LP];      // its rows reversed: sLt[k LP + q] = L(k, k-1-q) (q < k), zero beyond: the back substitution's walk
  __shared__ __attribute__((aligned(16))) double sU[N * LD];       // Q_ux, column-major (38 x 76)
  __shared__ __attribute__((aligned(16))) double sK[NR * LD];      // [k | K], column-major (38 x 77)
  __shared__ double sDinv[M], sQ[N + M];
  double* const sLt = sLt_ + 2 * LP;   // rows -2, -1 exist (zero): the backward loop prefetches two rows ahead without a clamp
  __shared__ int s_failed;

  double* Vx = p.ws_V + (int64_t)b * (n + n * n);
  double* Vxx = Vx + n;
  const double* Q = p.ws_Q + (int64_t)b * (n + m + n * n + m * n + m * m);
  const double* Qxx = Q + n + m;
  const double* Qux = Qxx + n * n;
  const double* Quu = Qux + m * n;
  // c_accumulate == 2: K3 ran beside K5 and left its contracted blocks in a workspace of its own; Q = P + C is formed here
  // (the same single addition K3's accumulating epilogue makes: the tensor term comes last, ddp_bwd.ipp:75,81,87)
  const bool addc = p.c_accumulate == 2 && p.has_tensors != 0;
  const double* Cxx = p.ws_D + (int64_t)b * (n * n + m * n + m * m);
  const double* Cux = Cxx + n * n;
  const double* Cuu = Cux + m * n;

  STAMP(8);
  // wave 0 only fetches what the factorisation needs -- its row of Q_uu -- and clears the two images of L itself, so that it can
  // start the LLT as soon as those 38 words per lane are there; the other seven waves stage Q_x | Q_u and Q_ux meanwhile (they are
  // first read behind the barrier that follows the factorisation)
  double a[M];
  if (wave == 0) {
    if (tid == 0) s_failed = 0;
    const double reg = p.reg[b];
    // row `lane` of the lower triangle of Q_uu + reg I (ddp_bwd.ipp:104), entry j at a[j]
#pragma unroll
    for (int j = 0; j < M; ++j) {
      double qv_ = 0.0;
      if (lane < m && j <= lane) { qv_ = Quu[lane + j * m]; if (addc) qv_ = qv_ + Cuu[lane + j * m]; qv_ = qv_ + (lane == j ? reg : 0.0); }
      a[j] = qv_;
    }
    for (int i = lane; i < (M + 2) * LP; i += 64) { sL[i] = 0.0; sLt_[i] = 0.0; }
  } else {
    constexpr int BSO = BS4 - 64;
    const int to = tid - 64;
    for (int i = to; i < n + m; i += BSO) sQ[i] = Q[i];
    for (int idx = to; idx < m * n; idx += BSO) sU[idx % m + (idx / m) * LD] = addc ? Qux[idx] + Cux[idx] : Qux[idx];
  }
  STAMP(9);
  if (wave == 0) {
    // Cholesky, lower triangle only; fail <=> pivot <= 0 (:105).  Per entry the updates arrive in ascending k: the order
    // of Eigen's unblocked LLT.  After step k the row is shifted: a[p] holds entry (lane, k+1+p).
    // Software-pipelined over the columns: the next pivot only needs entry (lane, k+1) updated with column k -- one
    // broadcast (v_readlane) and one FMA -- so its sqrt and division (the long dependent chain) are issued while the other
    // 36 entries of the row are still being updated with column k from LDS.
    bool failed = false;
    double lik;                                                    // L(lane, k), the finished column k (dk on the diagonal)
    double diag = 1.0;                                             // L(lane, lane) once column `lane` is finished
    {
      const double piv = readlane_f64(a[0], 0);
      failed = !(piv > 0.0);
      const double dk = sqrt(piv);
      lik = lane == 0 ? dk : a[0] / dk;
      if (lane > 0 && lane < m) { sL[lane - 1] = lik; sLt[lane * LP + (lane - 1)] = lik; }
      diag = dk;                                   // lane 0's; the others take theirs in the column loop
      __builtin_amdgcn_wave_barrier();
    }
    const double* cj = static_cast<const double*>(__builtin_assume_aligned(sL, 16));
    int k = 0;
    chol_phases<0, M, LP>(a, lik, failed, k, cj, lane, m, sL, sLt, diag);
    if (lane < m) sDinv[lane] = 1.0 / diag;        // (garbage on a failed factorisation: nobody reads it then)
    if (failed && lane == 0) s_failed = 1;
  } else if (wave == 3) {
    double* fbo = p.fb_origin + bt * nx;
    const double* xt = p.x + ((int64_t)b * (T + 1) + t) * nx;
    for (int i = lane; i < nx; i += 64) fbo[i] = xt[i];                        // :134 (harmless if the step fails: the sweep restarts)
  }
  __syncthreads();
  STAMP(10);
  if (s_failed) {
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
  // right-hand sides [-Q_u | -Q_ux] (:135-136): lane rc of waves 1-2 owns column rc in registers; L from LDS (broadcasts)
  const int rc = tid - 64;
  if (rc >= 0 && rc < NR) {
    double r[M];
#pragma unroll
    for (int l = 0; l < m; ++l) r[l] = -(rc == 0 ? sQ[n + l] : sU[l + (rc - 1) * LD]);
    double* out = sK + rc * LD;
    // forward: y_k = r_k / L_kk, r_l -= L_lk y_k (l > k); the running column is shifted so that the pivot is r[0].
    // Column k+1 of L is fetched (LDS broadcasts) while column k is applied: two register buffers, the loop unrolled by 2.
    static_assert(M % 2 == 0, "substitutions unrolled by two");
    STAMP_T(16, 64);
    double La[M - 1], Lb[M - 1];
    {
      const double* colp = static_cast<const double*>(__builtin_assume_aligned(sL, 16));   // column k (one pointer, constant offsets)
      double* outp = out;
      const double* dinvp = sDinv;
#pragma unroll
      for (int q = 0; q < M - 1; ++q) La[q] = colp[q];
      subst_phases<0, M, LP, +1>(r, La, Lb, colp, outp, dinvp);
    }
    STAMP_T(17, 64);
    // backward: x_k = y_k / L_kk, y_i -= L_ki x_k (i < k), k descending; the column is reloaded reversed (r[q] = y_{M-1-q})
#pragma unroll
    for (int q = 0; q < m; ++q) r[q] = out[m - 1 - q];
    {
      const double* rowp = static_cast<const double*>(__builtin_assume_aligned(sLt + (m - 1) * LP, 16));   // row k reversed: L(k, k-1-q)
      double* outp = out + (m - 1);
      const double* dinvp = sDinv + (m - 1);
#pragma unroll
      for (int q = 0; q < M - 1; ++q) La[q] = rowp[q];
      subst_phases<0, M, LP, -1>(r, La, Lb, rowp, outp, dinvp);
    }
    STAMP_T(18, 64);
    double* dst = rc == 0 ? p.fb_val + bt * m : p.fb_jac + bt * m * n + (rc - 1) * m;
    for (int l = 0; l < m; ++l) dst[l] = out[l];
    STAMP_T(19, 64);
  }
  __syncthreads();
  STAMP(11);
  // V_x = Q_x + Q_ux^T k (:142-143)
  for (int i = tid; i < n; i += BS4) {
    double s = 0.0;
#pragma unroll 2
    for (int l = 0; l < m; ++l) s += sU[l + i * LD] * sK[l];
    const double v = sQ[i] + s;
    Vx[i] = v;
    if (p.vx_trace) p.vx_trace[bt * n + i] = v;
  }
  // V_xx = Q_xx + Q_ux^T K (:145-146) on the matrix cores, transposed so that consecutive lanes own consecutive rows:
  // D'(j, i) = sum_l K(l, j) Q_ux(l, i): A(row = j, k = l) = K(l, j), B(k = l, col = i) = Q_ux(l, i)
  constexpr int TT = (N + 15) / 16, KS = (M + 3) / 4, TPW = (TT * TT + NW - 1) / NW;
  // the Q_xx entries of this wave's tiles are fetched up front, so that their latency hides behind the products
  double qv[TPW][4];
#pragma unroll
  for (int it_ = 0; it_ < TPW; ++it_) {
    const int tile = wave + it_ * NW;
    const int jt = tile / TT, it = tile % TT;
    const int ib = 16 * it + l15;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = 16 * jt + l4 + 4 * q;
      qv[it_][q] = (tile < TT * TT && j < n && ib < n) ? (addc ? Qxx[ib + j * n] + Cxx[ib + j * n] : Qxx[ib + j * n]) : 0.0;
    }
  }
#pragma unroll
  for (int it_ = 0; it_ < TPW; ++it_) {
    const int tile = wave + it_ * NW;
    if (tile >= TT * TT) break;
    const int jt = tile / TT, it = tile % TT;
    const int ja = 16 * jt + l15, ib = 16 * it + l15;
    const bool jok = ja < n, iok = ib < n;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int l = 4 * s + l4;
      const double av = (jok && l < m) ? sK[l + (ja + 1) * LD] : 0.0;
      const double bv = (iok && l < m) ? sU[l + ib * LD] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = 16 * jt + l4 + 4 * q;
      if (j >= n || !iok) continue;
      const int o = ib + j * n;
      const double v = qv[it_][q] + acc[q];
      Vxx[o] = v;
      if (p.vxx_trace) p.vxx_trace[bt * n * n + o] = v;
    }
  }
  __syncthreads();
  STAMP(12);
  if (t == 0 && tid == 0) p.status[b] = 2;                                     // :149-151
}
