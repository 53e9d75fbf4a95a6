// Blocked right-looking Cholesky for gfx950, fp64, column-major lower triangle, in place.
// Replaces scipy.linalg.cho_factor(Kcov_precon, lower=True) (reference Kernel.py:251; LAPACK dpotrf).
//
// Panel (width nb_outer, default 256) = 64-wide inner steps of
//     potrf64   one wave, a matrix row per lane held in registers
//     trsm64    X <- X L_kk^-T, a matrix row per lane (rows are contiguous in column-major -> coalesced)
//     gemm      update of the remaining panel columns
// followed by the trailing update  C -= A_panel A_panel^T  on v_mfma_f64_16x16x4_f64
// (64 cycles / instruction / SIMD measured = 78.6 TFLOP/s chip peak), 128x128 tiles, LDS-staged
// k-chunks, the C tile loaded straight into the accumulators.
// Right-hand-side rows stored below the matrix (rows Npad .. ld) ride along every trsm / gemm, which
// yields L^-1 B for free (no separate forward substitution on the likelihood path).
#include "gpg_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

// ------------------------------------------------------------------------------------------------
// potrf64: Cholesky of one 64 x 64 diagonal block by ONE wave, lane i <-> matrix row i.
// Left-looking over four 16-column sub-blocks: (1) update the sub-block's 16 entries of every row with
// the finished columns (LDS image, column-major: own entry per lane + wave-uniform broadcast reads),
// (2) factor the 16 columns in registers (pivot / column broadcast by v_readlane; rows below the
// diagonal sub-block are scaled in the same sweep), (3) publish to LDS and to global memory.
// Entries above the diagonal carry garbage that never feeds a valid entry.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_d(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// Wave-level core: `src` is the 64 x 64 block to factor (column-major, leading dimension sld; global memory
// or an LDS tile), the factor goes to `blk` (global, leading dimension ld) and its reciprocal pivots to dinv.
// St is a [64][64] LDS scratch private to the calling wave (St[k][i] = L[i][k]).  Returns the 1-based index
// of the first non-positive / NaN pivot inside the block (0 = none), identical in every lane.
// Per 16-column sub-block: (1) the update with the finished columns runs on MFMA (operands straight from the
// St image; 4 row tiles x K/4 instructions) and is transposed into the lane = row layout through the part
// of St that this sub-block is about to fill; (2) the 16 columns are factored in registers, pivot and
// column broadcast by v_readlane.  The reciprocal pivot comes from v_rsq_f64 plus two coupled
// Goldschmidt steps (sqrt and 1/sqrt to ~1 ulp in 7 dependent operations; the pivot chain is the critical
// path of every factorisation in this file).
__device__ __forceinline__ int potrf64_wave(const double* src, int sld, double (*St)[64], double* __restrict__ blk, int ld,
                                            double* __restrict__ dinv, int* piece_flags = nullptr) {
  const int i = threadIdx.x & 63, l15 = i & 15, l4 = i >> 4;
  int bad = 0;
  double myinv = 0.0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    double a[16];
    if (s == 0) {
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = src[i + (size_t)c * sld];
    } else {
      d4 acc[4];
#pragma unroll
      for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[R][r] = src[(16 * R + l15) + (size_t)(16 * s + 4 * r + l4) * sld];
#pragma unroll
      for (int kk = 0; kk < 4 * s; ++kk) {
        const double fn = St[4 * kk + l4][16 * s + l15];
        double fm[4];
#pragma unroll
        for (int R = 0; R < 4; ++R) fm[R] = -St[4 * kk + l4][16 * R + l15];
#pragma unroll
        for (int R = 0; R < 4; ++R) acc[R] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn, fm[R], acc[R], 0, 0, 0);
      }
#pragma unroll
      for (int R = 0; R < 4; ++R)
#pragma unroll
        for (int r = 0; r < 4; ++r) St[16 * s + 4 * r + l4][16 * R + l15] = acc[R][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave writes and reads St: ordering only
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = St[16 * s + c][i];
    }
    // Pivot chain, software-pipelined: the next pivot a[c+1][c+1] - L[c+1][c]^2 only needs the diagonal lane's own
    // scaled entry, so it is formed and broadcast BEFORE column c is applied to the other columns; the 15 - c
    // broadcast-FMA updates then fill the latency of the next v_rsq_f64 + Goldschmidt chain.
    double ajj = readlane_d(a[0], 16 * s);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      bad = (bad == 0 && !(ajj > 0.0)) ? 16 * s + c + 1 : bad;   // first non-positive / NaN pivot (LAPACK info)
      const double y0 = __builtin_amdgcn_rsq(ajj);
      double g = ajj * y0, h = 0.5 * y0;
      double r = __builtin_fma(-h, g, 0.5);
      g = __builtin_fma(g, r, g);
      h = __builtin_fma(h, r, h);
      r = __builtin_fma(-h, g, 0.5);
      const double dj = __builtin_fma(g, r, g);        // sqrt(ajj)
      const double inv = 2.0 * __builtin_fma(h, r, h); // 1 / sqrt(ajj)
      const double lc = a[c] * inv;
      if (c < 15) {
        const double t = __builtin_fma(-lc, lc, a[c + 1]);     // exact in the lane of row 16 s + c + 1
        ajj = readlane_d(t, 16 * s + c + 1);
      }
      a[c] = (i == 16 * s + c) ? dj : lc;
      myinv = (i == 16 * s + c) ? inv : myinv;   // reciprocal pivots for the panel solves: lane j keeps 1 / L_jj
#pragma unroll
      for (int k2 = c + 1; k2 < 16; ++k2) a[k2] -= a[c] * readlane_d(a[c], 16 * s + k2);
      // pin the updated columns here: left to itself the compiler defers these FMAs to their consumers and
      // keeps all 120 broadcast multipliers of the sub-block alive in SGPRs (spilled through v_writelane)
#pragma unroll
      for (int k2 = c + 1; k2 < 16; ++k2) asm volatile("" : "+v"(a[k2]));
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      St[16 * s + c][i] = a[c];
      if (i >= 16 * s + c) blk[i + (size_t)(16 * s + c) * ld] = a[c];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (i >= 16 * s && i < 16 * s + 16) dinv[i] = myinv;
    if (piece_flags) {   // dataflow kernels: these 16 columns (and their reciprocal pivots) are final -- publish them
      __threadfence();
      if (i == 0) __hip_atomic_store(piece_flags + s, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return bad;
}

__global__ void __launch_bounds__(64) potrf64_kernel(double* __restrict__ A, int ld, int j0, int N, int* __restrict__ info,
                                                      double* __restrict__ dinv) {
  __shared__ __attribute__((aligned(16))) double St[64][64];   // St[k][i] = L[i][k]
  double* blk = A + (size_t)j0 + (size_t)j0 * ld;
  const int bad = potrf64_wave(blk, ld, St, blk, ld, dinv + j0);
  if (bad && threadIdx.x == 0 && j0 + bad - 1 < N) atomicCAS(info, 0, j0 + bad);
}

// ------------------------------------------------------------------------------------------------
// trsm64: X <- X L^-T for `rows` rows against a factorised 64 x 64 diagonal block (X is rows x 64).
// Four lanes share one matrix row: lane (rr = lane >> 2, q = lane & 3) keeps the 16 columns k = 4m + q
// of row rr in registers.  Column step j: the owner lane scales x_j by the reciprocal pivot, the quad
// broadcasts it with one DPP quad_perm per dword, and every lane updates its remaining columns with
// L[k][j] read from an LDS image laid out per (j, q) so that a lane's values are contiguous (b128
// reads, the four q-slices 144 B apart: conflict-free).  544 FMAs per lane instead of 2016 for a
// lane-per-row sweep; a wave covers 16 rows (4 columns x 128 contiguous bytes per global access).
// ------------------------------------------------------------------------------------------------
template <int Q>
__device__ __forceinline__ double quad_bcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, Q * 0x55, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, Q * 0x55, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(256) trsm64_kernel(const double* __restrict__ Lblk, int ldl,
                                                     const double* __restrict__ dinv, double* __restrict__ X, int ldx,
                                                     int rows) {
  __shared__ __attribute__((aligned(16))) double Ls[64][4][18];   // Ls[j][q][m] = L[4m + q][j]
  __shared__ double sdinv[64];
  for (int t = threadIdx.x; t < 64 * 64; t += 256) {
    const int j = t >> 6, k = t & 63;
    Ls[j][k & 3][k >> 2] = Lblk[k + (size_t)j * ldl];
  }
  if (threadIdx.x < 64) sdinv[threadIdx.x] = dinv[threadIdx.x];
  __syncthreads();
  const int q = threadIdx.x & 3;
  int r = blockIdx.x * 64 + (threadIdx.x >> 2);
  const bool active = r < rows;
  r = active ? r : rows - 1;
  double* Xr = X + r + (size_t)q * ldx;
  double x[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) x[m] = Xr[(size_t)(4 * m) * ldx];
  // software pipeline: the L values of step j+1 are read from LDS while step j computes; the
  // sched_barrier keeps the compiler from hoisting every read to the top (register blow-up).
  double lv[2][16];
#pragma unroll
  for (int m = 0; m < 16; ++m) lv[0][m] = Ls[0][q][m];
#define GPG_TRSM_STEP(QJ)                                                                   \
  {                                                                                         \
    constexpr int cur = QJ & 1, nxt = cur ^ 1;                                               \
    const int j = 4 * mj + QJ;                                                              \
    const int jn = j + 1 < 64 ? j + 1 : 63;                                                 \
    const int m0n = (j + 1) >> 2;                                                           \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) if (m >= m0n) lv[nxt][m] = Ls[jn][q][m]; \
    const double xs = x[mj] * sdinv[j];                                                     \
    x[mj] = (q == QJ) ? xs : x[mj];                                                         \
    const double xj = quad_bcast<QJ>(x[mj]);                                                \
    if (QJ < 3) {                                                                           \
      const double t = x[mj] - xj * lv[cur][mj];                                            \
      x[mj] = (q > QJ) ? t : x[mj];                                                         \
    }                                                                                       \
    _Pragma("unroll") for (int m = mj + 1; m < 16; ++m) x[m] -= xj * lv[cur][m];            \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  }
#pragma unroll
  for (int mj = 0; mj < 16; ++mj) {
    GPG_TRSM_STEP(0)
    GPG_TRSM_STEP(1)
    GPG_TRSM_STEP(2)
    GPG_TRSM_STEP(3)
  }
#undef GPG_TRSM_STEP
  if (active) {
#pragma unroll
    for (int m = 0; m < 16; ++m) Xr[(size_t)(4 * m) * ldx] = x[m];
  }
}

// ------------------------------------------------------------------------------------------------
// wave_tile_gemm: acc (wave tile 16 x 64 of a 64 x 64 workgroup tile) -= A[64 x K] B[64 x K]^T with
// K = 16 nchunk (nchunk a positive multiple of 4).  Both operands are staged through LDS in 16-deep
// chunks; the global loads run two chunks ahead in registers (one wave per SIMD: nothing else hides
// the L2 latency).  ga / gb are this thread's staging sources (row pair sp, k-row sk of chunk 0).
// Ends without a barrier: the caller synchronises before re-using sA / sB.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_tile_gemm(d4 (&acc)[4], const double* ga, int lda, const double* gb, int ldb, int nchunk,
                                               double* sA, double* sB, int w, int l15, int l4, int sp, int sk) {
  constexpr int KB = 16, SA = 80, BUF = KB * SA;
  const size_t a8 = (size_t)8 * lda, b8 = (size_t)8 * ldb;
  double2 ra0_a, ra0_b, rb0_a, rb0_b, ra1_a, ra1_b, rb1_a, rb1_b;
#define GPG_PS_GLOAD(set)                                        \
  ra##set##_a = *reinterpret_cast<const double2*>(ga);           \
  ra##set##_b = *reinterpret_cast<const double2*>(ga + a8);      \
  rb##set##_a = *reinterpret_cast<const double2*>(gb);           \
  rb##set##_b = *reinterpret_cast<const double2*>(gb + b8);      \
  ga += 2 * a8;                                                  \
  gb += 2 * b8;
#define GPG_PS_SSTORE(buf, set)                                                              \
  {                                                                                          \
    double2 v0, v1;                                                                          \
    v0.x = -ra##set##_a.x; v0.y = -ra##set##_a.y; v1.x = -ra##set##_b.x; v1.y = -ra##set##_b.y;  \
    *reinterpret_cast<double2*>(sA + (buf) * BUF + sk * SA + 2 * sp) = v0;                    \
    *reinterpret_cast<double2*>(sA + (buf) * BUF + (sk + 8) * SA + 2 * sp) = v1;              \
    *reinterpret_cast<double2*>(sB + (buf) * BUF + sk * SA + 2 * sp) = rb##set##_a;           \
    *reinterpret_cast<double2*>(sB + (buf) * BUF + (sk + 8) * SA + 2 * sp) = rb##set##_b;     \
  }
#define GPG_PS_COMPUTE(buf)                                                                  \
  _Pragma("unroll") for (int kk = 0; kk < KB; kk += 4) {                                      \
    const double fm = sA[(buf) * BUF + (kk + l4) * SA + 16 * w + l15];                        \
    double fn[4];                                                                            \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) fn[ni] = sB[(buf) * BUF + (kk + l4) * SA + ni * 16 + l15]; \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                          \
      acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn[ni], fm, acc[ni], 0, 0, 0);           \
  }
  GPG_PS_GLOAD(0);            // chunk 0
  GPG_PS_GLOAD(1);            // chunk 1
  GPG_PS_SSTORE(0, 0);
  __syncthreads();
  for (int ch = 0; ch < nchunk; ch += 2) {   // unrolled by two so that the register sets are static
    if (ch + 2 < nchunk) { GPG_PS_GLOAD(0); }      // chunk ch + 2
    GPG_PS_COMPUTE(0);                             // chunk ch
    GPG_PS_SSTORE(1, 1);                           // chunk ch + 1
    __syncthreads();
    if (ch + 3 < nchunk) { GPG_PS_GLOAD(1); }      // chunk ch + 3
    GPG_PS_COMPUTE(1);                             // chunk ch + 1
    if (ch + 2 < nchunk) { GPG_PS_SSTORE(0, 0); }  // chunk ch + 2
    __syncthreads();
  }
#undef GPG_PS_GLOAD
#undef GPG_PS_SSTORE
#undef GPG_PS_COMPUTE
}

// quad-row substitution x <- x L^-T of one matrix row spread over a lane quad (see trsm64_kernel): x[m] is
// column 4m + q; Ls is the LDS image Ls[j][q][m] = L[4m + q][j], sdinv the reciprocal pivots.
#define GPG_QUAD_SUBST(x, Ls, sdinv, q)                                                      \
  {                                                                                         \
    double lv[2][16];                                                                       \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) lv[0][m] = Ls[0][q][m];                   \
    _Pragma("unroll") for (int mj = 0; mj < 16; ++mj) {                                      \
      GPG_QS_STEP(x, Ls, sdinv, q, 0)                                                        \
      GPG_QS_STEP(x, Ls, sdinv, q, 1)                                                        \
      GPG_QS_STEP(x, Ls, sdinv, q, 2)                                                        \
      GPG_QS_STEP(x, Ls, sdinv, q, 3)                                                        \
    }                                                                                       \
  }
#define GPG_QS_STEP(x, Ls, sdinv, q, QJ)                                                     \
  {                                                                                         \
    constexpr int cur = QJ & 1, nxt = cur ^ 1;                                               \
    const int jc = 4 * mj + QJ;                                                             \
    const int jn = jc + 1 < 64 ? jc + 1 : 63;                                               \
    const int m0n = (jc + 1) >> 2;                                                          \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) if (m >= m0n) lv[nxt][m] = Ls[jn][q][m]; \
    const double xs = x[mj] * sdinv[jc];                                                    \
    x[mj] = (q == QJ) ? xs : x[mj];                                                         \
    const double xj = quad_bcast<QJ>(x[mj]);                                                \
    if (QJ < 3) {                                                                           \
      const double t = x[mj] - xj * lv[cur][mj];                                            \
      x[mj] = (q > QJ) ? t : x[mj];                                                         \
    }                                                                                       \
    _Pragma("unroll") for (int m = mj + 1; m < 16; ++m) x[m] -= xj * lv[cur][m];            \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  }

// one 16-column piece (columns 16 s .. 16 s + 15) of GPG_QUAD_SUBST: the dataflow kernels substitute against the
// diagonal block while it is still being factored, piece by piece as its columns are published
#define GPG_QUAD_SUBST_PIECE(x, Ls, sdinv, q, s)                                             \
  {                                                                                         \
    double lv[2][16];                                                                       \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) lv[0][m] = Ls[16 * (s)][q][m];            \
    _Pragma("unroll") for (int mj = 4 * (s); mj < 4 * (s) + 4; ++mj) {                       \
      GPG_QS_STEP(x, Ls, sdinv, q, 0)                                                        \
      GPG_QS_STEP(x, Ls, sdinv, q, 1)                                                        \
      GPG_QS_STEP(x, Ls, sdinv, q, 2)                                                        \
      GPG_QS_STEP(x, Ls, sdinv, q, 3)                                                        \
    }                                                                                       \
  }
// two matrix rows per lane quad (x0, x1): the L values are read once for both rows
#define GPG_QUAD_SUBST2(x0, x1, Ls, sdinv, q)                                                \
  {                                                                                         \
    double lv[2][16];                                                                       \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) lv[0][m] = Ls[0][q][m];                   \
    _Pragma("unroll") for (int mj = 0; mj < 16; ++mj) {                                      \
      GPG_QS2_STEP(x0, x1, Ls, sdinv, q, 0)                                                  \
      GPG_QS2_STEP(x0, x1, Ls, sdinv, q, 1)                                                  \
      GPG_QS2_STEP(x0, x1, Ls, sdinv, q, 2)                                                  \
      GPG_QS2_STEP(x0, x1, Ls, sdinv, q, 3)                                                  \
    }                                                                                       \
  }
#define GPG_QS2_STEP(x0, x1, Ls, sdinv, q, QJ)                                               \
  {                                                                                         \
    constexpr int cur = QJ & 1, nxt = cur ^ 1;                                               \
    const int jc = 4 * mj + QJ;                                                             \
    const int jn = jc + 1 < 64 ? jc + 1 : 63;                                               \
    const int m0n = (jc + 1) >> 2;                                                          \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) if (m >= m0n) lv[nxt][m] = Ls[jn][q][m]; \
    const double dv = sdinv[jc];                                                            \
    const double xs0 = x0[mj] * dv, xs1 = x1[mj] * dv;                                      \
    x0[mj] = (q == QJ) ? xs0 : x0[mj];                                                      \
    x1[mj] = (q == QJ) ? xs1 : x1[mj];                                                      \
    const double xj0 = quad_bcast<QJ>(x0[mj]), xj1 = quad_bcast<QJ>(x1[mj]);                \
    if (QJ < 3) {                                                                           \
      const double t0 = x0[mj] - xj0 * lv[cur][mj], t1 = x1[mj] - xj1 * lv[cur][mj];        \
      x0[mj] = (q > QJ) ? t0 : x0[mj];                                                      \
      x1[mj] = (q > QJ) ? t1 : x1[mj];                                                      \
    }                                                                                       \
    _Pragma("unroll") for (int m = mj + 1; m < 16; ++m) {                                    \
      x0[m] -= xj0 * lv[cur][m];                                                            \
      x1[m] -= xj1 * lv[cur][m];                                                            \
    }                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  }

#ifdef GPG_STAMP   // diagnostic build of tools/gemm_probe.hip only: per-wave cycle shares of the loop phases
__device__ unsigned long long* g_stamp_buf;
#define GPG_T(var) unsigned long long var; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0);
#else
#define GPG_T(var)
#endif

// ------------------------------------------------------------------------------------------------
// panel_solve_kernel: X <- X L^-T for `rows` rows against a factorised nb x nb diagonal block
// (nb a multiple of 64) in ONE launch.  A workgroup owns 64 rows; wave w owns rows 16w .. 16w+15.
// Left-looking over the 64-column blocks j of the panel:
//   (1) T_j = X_j - sum_{k<j} X_k L_jk^T   on MFMA (wave tile 16 x 64, K = 64 j; X_k is the workgroup's own
//       earlier output re-read through L1/L2, L_jk comes from L2; both staged through LDS, 16-deep chunks)
//   (2) X_j = T_j L_jj^-T by the quad-row substitution of trsm64_kernel (the accumulators are transposed
//       through a wave-private LDS tile into the 4-lanes-per-row layout)
// Replaces nb/64 trsm64 + nb/64 - 1 small-K gemm launches, whose ~15 us dependent-launch latency each
// (not their flops) set the cost of B_p.
// ------------------------------------------------------------------------------------------------
// Body shared with the dataflow kernels: solves the 64 rows starting at X (rows_left of them are real) with
// the whole workgroup.  U = staging / transposition buffer (4 * 16 * 80 doubles), Ls / sdinv = image of the
// current diagonal block.  Ends with a workgroup barrier.
__device__ __forceinline__ void panel_solve_rows64(const double* __restrict__ L, int ldl, const double* __restrict__ dinv,
                                                   double* X, int ldx, int rows, int nb, double* U, double (*Ls)[4][18],
                                                   double* sdinv) {
  constexpr int KB = 16, SA = 80;                     // SA: +128 B pad keeps ds_read_b64 conflict-free
  constexpr int BUF = KB * SA;                        // doubles per staging buffer
  double* const sA = U;
  double* const sB = U + 2 * BUF;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  // MFMA-side row of this lane, substitution-side row of this lane (both inside the wave's 16 rows)
  int rowc = 16 * w + l15;
  const bool c_ok = rowc < rows;
  rowc = c_ok ? rowc : rows - 1;
  const int q = tid & 3;
  int rowt = tid >> 2;
  const bool t_ok = rowt < rows;
  rowt = t_ok ? rowt : rows - 1;

  // staging: thread -> (row pair p, k) ; two double2 per operand per chunk
  const int sp = tid & 31, sk = tid >> 5;             // sk in 0..7, second load at sk + 8
  int rowa = 2 * sp;
  rowa = rowa + 1 < rows ? rowa : (rows >= 2 ? rows - 2 : 0);

  double cx[16], li[16];   // next block's X tile (MFMA layout) and L_jj (linear), prefetched
#define GPG_PS_PREFETCH(jb)                                                                   \
  {                                                                                          \
    const double* Cw = X + rowc + (size_t)(64 * (jb) + l4) * ldx;                             \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) cx[i] = Cw[(size_t)((i >> 2) * 16 + 4 * (i & 3)) * ldx]; \
    const double* Ljj = L + (size_t)(64 * (jb)) + (size_t)(64 * (jb)) * ldl;                   \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                          \
      const int t = tid + 256 * i;                                                           \
      li[i] = Ljj[(t & 63) + (size_t)(t >> 6) * ldl];                                         \
    }                                                                                        \
  }
  GPG_PS_PREFETCH(0)
#ifdef GPG_STAMP
  unsigned long long ps_pre = 0, ps_gemm = 0, ps_tr = 0, ps_sub = 0, ps_st = 0;
#endif
  for (int j = 0; j < nb / 64; ++j) {
    GPG_T(p0)
    // ---- accumulators start as X_j; LDS image of L_jj for the substitution (nobody reads Ls during the MFMA
    //      phase).  Both were fetched into registers one block ahead, behind the previous substitution. ----------
    d4 acc[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ni][r] = cx[4 * ni + r];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = tid + 256 * i, jj = t >> 6, k = t & 63;
      Ls[jj][k & 3][k >> 2] = li[i];
    }
    if (tid < 64) sdinv[tid] = dinv[64 * j + tid];
    // ---- (1) MFMA phase -----------------------------------------------------------------------------------
    const int nchunk = 4 * j;                          // K = 64 j in chunks of 16
    GPG_T(p1)
    if (nchunk > 0)
      wave_tile_gemm(acc, X + rowa + (size_t)sk * ldx, ldx, L + (size_t)(64 * j + 2 * sp) + (size_t)sk * ldl, ldl, nchunk, sA, sB,
                     w, l15, l4, sp, sk);
    __syncthreads();   // staging buffers free (they become Ts), L_jj image complete
    GPG_T(p2)
    // ---- (2) substitution phase: accumulators -> Ts[col][row] -> 4 lanes per row ---------------------------
    {
      double* Ts = U;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ts[(ni * 16 + 4 * r + l4) * SA + 16 * w + l15] = acc[ni][r];
    }
    __syncthreads();
    double x[16];
    {
      const double* Tr = U + q * SA + (tid >> 2);
#pragma unroll
      for (int m = 0; m < 16; ++m) x[m] = Tr[(4 * m) * SA];
    }
    if (j + 1 < nb / 64) GPG_PS_PREFETCH(j + 1)
    GPG_T(p3)
    GPG_QUAD_SUBST(x, Ls, sdinv, q)
    GPG_T(p4)
    if (t_ok) {
      double* Xr = X + rowt + (size_t)(64 * j + q) * ldx;
#pragma unroll
      for (int m = 0; m < 16; ++m) Xr[(size_t)(4 * m) * ldx] = x[m];
    }
    __syncthreads();   // X_j visible to the whole workgroup (vmcnt(0) + barrier), Ts / Ls free again
    GPG_T(p5)
#ifdef GPG_STAMP
    ps_pre += p1 - p0; ps_gemm += p2 - p1; ps_tr += p3 - p2; ps_sub += p4 - p3; ps_st += p5 - p4;
#endif
  }
#undef GPG_PS_PREFETCH
#ifdef GPG_STAMP
  if (lane == 0 && blockIdx.x < 4096 && g_stamp_buf != nullptr) {
    unsigned long long* o = g_stamp_buf + ((size_t)blockIdx.x * 4 + w) * 8;
    o[0] = ps_pre; o[1] = ps_gemm; o[2] = ps_tr; o[3] = ps_sub; o[4] = ps_st;
  }
#endif
}

__global__ void __launch_bounds__(256, 2)
panel_solve_kernel(const double* __restrict__ L, int ldl, const double* __restrict__ dinv, double* X, int ldx,
                   int rows, int nb) {
  __shared__ __attribute__((aligned(16))) double U[4 * 16 * 80];   // sA[2] | sB[2]; re-used as the transposition tile Ts[64][80]
  __shared__ __attribute__((aligned(16))) double Ls[64][4][18];
  __shared__ double sdinv[64];
  const int m0 = blockIdx.x * 64;
  panel_solve_rows64(L, ldl, dinv, X + m0, ldx, rows - m0, nb, U, Ls, sdinv);
}

// ------------------------------------------------------------------------------------------------
// tile_chol_kernel: dataflow (left-looking) Cholesky of the trailing block A[c0:, c0:] in 64 x 64 tiles,
// ONE launch.  Used where the blocked algorithm is latency-bound: the last few thousand columns of a large
// matrix and small matrices as a whole.  Workgroup b owns tile (i, j) = tasks[b] (column-major task order,
// rows i >= j; the right-hand-side rows below the matrix are ordinary tile rows):
//     acc  = A_ij - sum_{k<j} L_ik L_jk^T      MFMA, k-blocks consumed as soon as their flags are up
//     i==j : L_jj = chol(acc) by wave 0 (potrf64_wave), reciprocal pivots to dinv
//     i> j : L_ij = acc L_jj^-T by the quad-row substitution, once flag(j, j) is up
//     publish: __threadfence, then flag(i, j) = 1 (agent-scope release)
// A task only ever waits for tasks with a smaller index, and workgroups are dispatched in index order, so
// the oldest unfinished workgroup can always run to completion (no deadlock whatever the residency).  As
// a backstop every wait is bounded in time: on timeout the kernel raises the abort word, all workgroups drain, and
// the host falls back on the blocked schedule.  (That does happen when two such launches share the GPU, e.g.
// two processes on one device: each launch's waiting workgroups can hold the slots the other one's oldest
// pending workgroup needs.)  The serial chain per 64 columns is potrf -> substitution ->
// one 64-deep MFMA block (~20 us) instead of three dependent launches per step plus B_p and U_p.
// ------------------------------------------------------------------------------------------------
#define GPG_TILE_WAIT_TICKS 25000000ull   // bound of every dependency wait: 0.25 s of the 100 MHz s_memrealtime clock
// Whole-workgroup wait on a completion flag (thread 0 polls, result shared through `sh`); 0 = timed out / aborted.
__device__ __forceinline__ int wg_wait_flag(int* flag, int* abort_word, int* info, int* sh) {
  if (threadIdx.x == 0) {
    int ok = 1;
    const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      if (__builtin_amdgcn_s_memrealtime() - t_wait > GPG_TILE_WAIT_TICKS || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicMax(info, GPG_INFO_INTERNAL);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    *sh = ok;
  }
  __syncthreads();
  const int ok = *sh;
  __syncthreads();   // sh may be rewritten by the next wait
  if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok;
}

// X (128 rows x 128 columns, in place) <- X L^-T against a factorised 128 x 128 diagonal tile, whole workgroup.
// Each lane quad carries TWO matrix rows (r and r + 64) through the substitution, so the 128 rows cost two
// substitution sweeps instead of four and half the L-image traffic.  Column block 0 is read from memory
// straight in the quad layout; block 1 first takes its update X1 L21^T on MFMA (two 64-row passes through
// the LDS tile).  U: 4 * 16 * 80 doubles, Ls / sdinv: diagonal-block image.  Ends with a workgroup barrier.
__global__ void __launch_bounds__(256, 2)
tile_chol_kernel(double* A, int ld, int c0, int Mt, const int* __restrict__ tasks, int* flags, int* pieces, int* abort_word,
                 double* __restrict__ dinv, int* __restrict__ info, int N) {
  constexpr int KB = 16, SA = 80, BUF = KB * SA;
  __shared__ __attribute__((aligned(16))) double U[4 * BUF];      // staging sA[2] | sB[2]; later the tile Ts[64][SA]
  __shared__ __attribute__((aligned(16))) double Ls[64][4][18];   // L_jj image (i > j) / potrf scratch St[64][64] (i == j)
  __shared__ double sdinv[64];
  __shared__ int sh_kr;
  double* const sA = U;
  double* const sB = U + 2 * BUF;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int task = tasks[blockIdx.x];
  const int ti = task & 0xffff, tj = task >> 16;
  const size_t r0 = (size_t)c0 + 64 * (size_t)ti;        // first matrix row of the tile
  const size_t cj = (size_t)c0 + 64 * (size_t)tj;        // first matrix column of the tile
  const int q = tid & 3;
  const int sp = tid & 31, sk = tid >> 5;
  int* const frow_i = flags + (size_t)ti * Mt;           // flags of tile row i
  int* const frow_j = flags + (size_t)tj * Mt;

  // accumulators start as A_ij
  d4 acc[4];
  {
    const double* Cw = A + r0 + 16 * w + l15 + (cj + l4) * (size_t)ld;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ni][r] = Cw[(size_t)(ni * 16 + 4 * r) * ld];
  }

  // ---- (1) left-looking accumulation over the finished tile columns ------------------------------------------
  int kdone = 0;
  while (kdone < tj) {
    if (tid == 0) {
      int kr = kdone;
      const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        while (kr < tj && __hip_atomic_load(frow_i + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 &&
               __hip_atomic_load(frow_j + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
          ++kr;
        if (kr > kdone) break;
        if (__builtin_amdgcn_s_memrealtime() - t_wait > GPG_TILE_WAIT_TICKS || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicMax(info, GPG_INFO_INTERNAL);
          kr = -1;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
      sh_kr = kr;
    }
    __syncthreads();
    const int kr = sh_kr;
    if (kr < 0) return;                                  // abort: drain
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the producers' tiles are visible from here on
    const size_t ck = (size_t)c0 + 64 * (size_t)kdone;
    wave_tile_gemm(acc, A + r0 + 2 * sp + (ck + sk) * (size_t)ld, ld, A + cj + 2 * sp + (ck + sk) * (size_t)ld, ld,
                   4 * (kr - kdone), sA, sB, w, l15, l4, sp, sk);
    __syncthreads();                                     // staging buffers free again; sh_kr may be rewritten
    kdone = kr;
  }

  // ---- (2) accumulators -> LDS tile Ts[col][row] --------------------------------------------------------------
  {
    double* Ts = U;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ts[(ni * 16 + 4 * r + l4) * SA + 16 * w + l15] = acc[ni][r];
  }
  if (ti == tj) {
    __syncthreads();
    if (w == 0) {   // diagonal tile: factor it (one wave; entries above the diagonal are garbage nobody reads)
      double* blk = A + r0 + cj * (size_t)ld;
      const int bad = potrf64_wave(U, SA, reinterpret_cast<double(*)[64]>(&Ls[0][0][0]), blk, ld, dinv + cj, pieces + 4 * tj);
      if (bad && lane == 0 && (int)cj + bad - 1 < N) atomicCAS(info, 0, (int)cj + bad);
    }
  } else {
    // substitute against the diagonal tile of this column piece by piece, as its 16-column pieces are published
    __syncthreads();
    double x[16];
    {
      const double* Tr = U + q * SA + (tid >> 2);
#pragma unroll
      for (int m = 0; m < 16; ++m) x[m] = Tr[(4 * m) * SA];
    }
    const double* Ljj = A + cj + cj * (size_t)ld;
#define GPG_TC_PIECE(S)                                                                      \
    {                                                                                       \
      if (!wg_wait_flag(pieces + 4 * tj + (S), abort_word, info, &sh_kr)) return;            \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                        \
        const int t = tid + 256 * u, jj = 16 * (S) + (t >> 6), k = t & 63;                   \
        Ls[jj][k & 3][k >> 2] = Ljj[k + (size_t)jj * ld];                                    \
      }                                                                                     \
      if (tid < 16) sdinv[16 * (S) + tid] = dinv[cj + 16 * (S) + tid];                       \
      __syncthreads();                                                                      \
      GPG_QUAD_SUBST_PIECE(x, Ls, sdinv, q, S)                                               \
      /* pin x: otherwise the FMAs of a piece are deferred into the next ones and everything spills */ \
      _Pragma("unroll") for (int m = 0; m < 16; ++m) asm volatile("" : "+v"(x[m]));              \
    }
    GPG_TC_PIECE(0)
    GPG_TC_PIECE(1)
    GPG_TC_PIECE(2)
    GPG_TC_PIECE(3)
#undef GPG_TC_PIECE
    double* Xr = A + r0 + (tid >> 2) + (cj + q) * (size_t)ld;
#pragma unroll
    for (int m = 0; m < 16; ++m) Xr[(size_t)(4 * m) * ld] = x[m];
  }
  // ---- (3) publish ----------------------------------------------------------------------------------------------
  __threadfence();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(frow_i + tj, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
// gemm_nt_minus<BM, BN>:  C[M x Nc] -= A[M x K] * B[Nc x K]^T   (all column-major)
//   M multiple of 64, Nc multiple of BN, K multiple of 8.  lower != 0: C's origin lies on the matrix
//   diagonal and tiles entirely above it are skipped; tiles with m0 < skipM and n0 < skipN are skipped too.
// 4 waves as 2 x 2; each wave owns (BM/2) x (BN/2) of C as 16x16 MFMA blocks.  The MFMA "A" operand
// is fed from the C-column side and the "B" operand from the C-row side, so that lane&15 indexes C's
// row: every accumulator load/store instruction touches 4 columns x 128 contiguous bytes.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ void __launch_bounds__(256, 2)
gemm_nt_minus_kernel(double* __restrict__ C, int ldc, const double* __restrict__ A, int lda,
                     const double* __restrict__ B, int ldb, int M, int Nc, int K, int lower, int skipM, int skipN,
                     const int* __restrict__ tilemap, int ntiles) {
  constexpr int KB = 8;
  constexpr int SA = BM + 16, SB = BN + 16;   // row strides: +128 B keeps ds_read_b64 conflict-free
  constexpr int MI = BM / 32, NI = BN / 32;   // 16x16 blocks per wave
  constexpr int LA = BM * KB / 512, LB = BN * KB / 512;  // double2 loads per thread per chunk
  __shared__ __attribute__((aligned(16))) double sA[2][KB][SA];
  __shared__ __attribute__((aligned(16))) double sB[2][KB][SB];

  int m0, n0;
  if (tilemap) {
    // 1-D grid over a precomputed list of live tiles in super-tile-major order.  Workgroups b and b + 8
    // share an XCD (round-robin dispatch), so each XCD is handed one contiguous chunk of the list: the
    // tiles resident on an XCD at any time share panel slices, which then hit in that XCD's L2.
    const int nwg = ntiles, b = blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int t = tilemap[v];
    m0 = (t & 0xffff) * BM;
    n0 = (t >> 16) * BN;
  } else {
    m0 = blockIdx.x * BM;
    n0 = blockIdx.y * BN;
    if (lower && m0 + BM <= n0) return;
    if (m0 < skipM && n0 < skipN) return;   // region owned by the look-ahead stream
  }

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w & 1, wn = w >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const bool wave_active = (m0 + wm * (BM / 2)) < M;   // ragged last row tile (M % 128 == 64)

  // ---- accumulators start as the C tile ----------------------------------------------------------
  d4 acc[NI][MI] = {};
  double* Cw = C + (size_t)(m0 + wm * (BM / 2) + l15) + (size_t)(n0 + wn * (BN / 2) + l4) * ldc;
  if (wave_active) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ni][mi][r] = Cw[mi * 16 + (size_t)(ni * 16 + 4 * r) * ldc];
  }

  // ---- global -> register -> LDS staging of one k-chunk (per-thread constants hoisted) ------------------
  // thread -> (row pair p, k) of the chunk; LA / LB double2 loads per thread per chunk
  const double* gA[LA];
  const double* gB[LB];
  double* lA[LA];
  double* lB[LB];
#pragma unroll
  for (int q = 0; q < LA; ++q) {
    const int idx = tid + 256 * q, p = idx % (BM / 2), k = idx / (BM / 2);
    int row = m0 + 2 * p;
    row = row < M ? row : M - 2;   // clamp: rows past M are never used
    gA[q] = A + row + (size_t)k * lda;
    lA[q] = &sA[0][k][2 * p];
  }
#pragma unroll
  for (int q = 0; q < LB; ++q) {
    const int idx = tid + 256 * q, p = idx % (BN / 2), k = idx / (BN / 2);
    gB[q] = B + n0 + 2 * p + (size_t)k * ldb;
    lB[q] = &sB[0][k][2 * p];
  }
  const size_t stepA = (size_t)KB * lda, stepB = (size_t)KB * ldb;
  constexpr int bufA = KB * SA, bufB = KB * SB;   // doubles per LDS buffer

#define GPG_GLOAD()                                                                    \
  _Pragma("unroll") for (int q = 0; q < LA; ++q) { ra[q] = *reinterpret_cast<const double2*>(gA[q]); gA[q] += stepA; } \
  _Pragma("unroll") for (int q = 0; q < LB; ++q) { rb[q] = *reinterpret_cast<const double2*>(gB[q]); gB[q] += stepB; }
#define GPG_SSTORE(buf)                                                                \
  _Pragma("unroll") for (int q = 0; q < LA; ++q) {                                      \
    double2 v; v.x = -ra[q].x; v.y = -ra[q].y;   /* C -= A B^T  ==  C += (-A) B^T */   \
    *reinterpret_cast<double2*>(lA[q] + (buf) * bufA) = v;                               \
  }                                                                                    \
  _Pragma("unroll") for (int q = 0; q < LB; ++q) *reinterpret_cast<double2*>(lB[q] + (buf) * bufB) = rb[q];
#define GPG_COMPUTE(buf)                                                               \
  _Pragma("unroll") for (int kk = 0; kk < KB; kk += 4) {                                \
    double fm[MI], fn[NI];                                                             \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) fm[mi] = sA[buf][kk + l4][wm * (BM / 2) + mi * 16 + l15]; \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) fn[ni] = sB[buf][kk + l4][wn * (BN / 2) + ni * 16 + l15]; \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                   \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                 \
        acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn[ni], fm[mi], acc[ni][mi], 0, 0, 0); \
  }

  double2 ra[LA], rb[LB];
  const int nchunk = K / KB;
  GPG_GLOAD();
  GPG_SSTORE(0);
  __syncthreads();
  for (int ch = 0; ch + 1 < nchunk; ++ch) {
    const int buf = ch & 1;
    GPG_GLOAD();          // prefetch the next k-chunk into registers
    GPG_COMPUTE(buf);     // inactive waves of a ragged tile compute on valid LDS, never store
    GPG_SSTORE(buf ^ 1);
    __syncthreads();
  }
  {
    const int buf = (nchunk - 1) & 1;
    GPG_COMPUTE(buf);
  }
#undef GPG_GLOAD
#undef GPG_SSTORE
#undef GPG_COMPUTE

  if (wave_active) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cw[mi * 16 + (size_t)(ni * 16 + 4 * r) * ldc] = acc[ni][mi][r];
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_dma_kernel: the 128 x 128 tile update  C -= A B^T  with the panel streamed by LDS-DMA.
// Same tile / wave / MFMA mapping as gemm_nt_minus_kernel<128,128>, but the k-chunks (8 deep) are
// written straight into a 4-stage LDS ring with global_load_lds_dwordx4: one wave-instruction moves one
// 1 KiB k-row of the A or B slice (rows are contiguous in the column-major panel and padded by 128 B in
// LDS, so each row is exactly one lane-linear DMA).  Three chunks stay in flight across the single raw
// s_barrier per chunk (counted vmcnt, never __syncthreads(), which would drain the DMAs); no staging
// VGPRs, no ds_write.  The sign of the update is folded into the A fragment (one v_xor per fragment).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;


template <int S>
__global__ void __launch_bounds__(256, 2)
gemm_dma_kernel(double* __restrict__ C, int ldc, const double* __restrict__ A, int lda, const double* __restrict__ B,
                int ldb, int M, int Nc, int K, int lower, int skipM, int skipN, const int* __restrict__ tilemap,
                int ntiles) {
  constexpr int BM = 128, BN = 128, KB = 8;
  constexpr int ROW = 144;                 // doubles per LDS k-row (128 + 16 pad: conflict-free ds_read_b64)
  constexpr int STAGE = 2 * KB * ROW;      // A rows then B rows
  __shared__ __attribute__((aligned(16))) double smem[S * STAGE];

  int m0, n0;
  if (tilemap) {
    const int nwg = ntiles, b = blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int t = tilemap[v];
    m0 = (t & 0xffff) * BM;
    n0 = (t >> 16) * BN;
  } else {
    m0 = blockIdx.x * BM;
    n0 = blockIdx.y * BN;
    if (lower && m0 + BM <= n0) return;
    if (m0 < skipM && n0 < skipN) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w & 1, wn = w >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const bool wave_active = (m0 + wm * 64) < M;

#ifdef GPG_STAMP
  const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();
#endif
  d4 acc[4][4] = {};
  double* Cw = C + (size_t)(m0 + wm * 64 + l15) + (size_t)(n0 + wn * 64 + l4) * ldc;
#ifndef GPG_ABLATE_CLOAD
  if (wave_active) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ni][mi][r] = Cw[mi * 16 + (size_t)(ni * 16 + 4 * r) * ldc];
  }
#endif

  // per-lane DMA sources: this wave moves k-rows w and w + 4 of every chunk, for A and for B
  int rowa = m0 + 2 * lane;
  rowa = rowa < M ? rowa : M - 2;          // ragged last row tile: clamp (those rows are never stored)
  const double* ga = A + rowa + (size_t)w * lda;
  const double* gb = B + n0 + 2 * lane + (size_t)w * ldb;
  const size_t a4 = (size_t)4 * lda, b4 = (size_t)4 * ldb, aK = (size_t)KB * lda, bK = (size_t)KB * ldb;
  double* const sbase = smem;

#ifdef GPG_ABLATE_DMA
#define GPG_DMA_ISSUE(stage) {}
#else
#define GPG_DMA_ISSUE(stage)                                                                              \
  {                                                                                                       \
    double* sa = sbase + (stage) * STAGE + w * ROW;                                                        \
    __builtin_amdgcn_global_load_lds((glb_ptr_t)ga, (lds_ptr_t)sa, 16, 0, 0);                              \
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga + a4), (lds_ptr_t)(sa + 4 * ROW), 16, 0, 0);          \
    __builtin_amdgcn_global_load_lds((glb_ptr_t)gb, (lds_ptr_t)(sa + KB * ROW), 16, 0, 0);                 \
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(gb + b4), (lds_ptr_t)(sa + (KB + 4) * ROW), 16, 0, 0);    \
    ga += aK;                                                                                             \
    gb += bK;                                                                                             \
  }
#endif
#define GPG_DMA_COMPUTE(stage)                                                                            \
  {                                                                                                       \
    const double* pa = sbase + (stage) * STAGE + l4 * ROW + wm * 64 + l15;                                 \
    const double* pb = sbase + (stage) * STAGE + (KB + l4) * ROW + wn * 64 + l15;                          \
    _Pragma("unroll") for (int kk = 0; kk < KB; kk += 4) {                                                 \
      double fm[4], fn[4];                                                                                \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) fm[mi] = -pa[kk * ROW + mi * 16];                    \
      _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) fn[ni] = pb[kk * ROW + ni * 16];                     \
      _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                     \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                   \
          acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn[ni], fm[mi], acc[ni][mi], 0, 0, 0);        \
    }                                                                                                     \
  }

  const int nchunk = K / KB;
  // prologue: S-1 chunks in flight
#pragma unroll
  for (int st = 0; st < S - 1; ++st)
    if (st < nchunk) GPG_DMA_ISSUE(st)
  // steady state: chunk i+1 must have landed, the S-2 younger chunks (4 DMAs per wave each) may fly on
#define GPG_WAIT_STEADY()                                            \
  if (S == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       \
  else if (S == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  \
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (nchunk >= S - 1) { GPG_WAIT_STEADY() }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int stage = 0;
#ifdef GPG_STAMP
  unsigned long long t_issue = 0, t_comp = 0, t_wait = 0, t_bar = 0;
  const unsigned long long tc0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int i = 0; i < nchunk; ++i) {
    const bool more = (i + S - 1) < nchunk;
    GPG_T(s0)
    if (more) {
      int st = stage + S - 1;
      st = st >= S ? st - S : st;
      GPG_DMA_ISSUE(st)
    }
    GPG_T(s1)
    GPG_DMA_COMPUTE(stage)
    GPG_T(s2)
    // chunk i+1 must have landed before anybody reads it; chunks i+2, i+3 may stay in flight
    if (more) { GPG_WAIT_STEADY() }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GPG_T(s3)
    __builtin_amdgcn_s_barrier();
    GPG_T(s4)
#ifdef GPG_STAMP
    t_issue += s1 - s0; t_comp += s2 - s1; t_wait += s3 - s2; t_bar += s4 - s3;
#endif
    stage = stage + 1 == S ? 0 : stage + 1;
  }
#ifdef GPG_STAMP
  const unsigned long long tl_main_end = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && blockIdx.x < 4096 && g_stamp_buf != nullptr) {
    unsigned long long* o = g_stamp_buf + ((size_t)blockIdx.x * 4 + w) * 4;
    const unsigned long long tc1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
    o[0] = t_issue; o[1] = t_comp; o[2] = tc1 - tc0; o[3] = tr1 - tr0;
  }
#endif
#undef GPG_DMA_ISSUE
#undef GPG_DMA_COMPUTE
#undef GPG_WAIT_STEADY

  if (wave_active) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cw[mi * 16 + (size_t)(ni * 16 + 4 * r) * ldc] = acc[ni][mi][r];
  }
#ifdef GPG_STAMP
  // timeline record of wave 0 (diagnostic): hw id, start, main-loop start, main-loop end, stores issued
  if (lane == 0 && w == 0 && blockIdx.x < 4096 && g_stamp_buf != nullptr) {
    unsigned long long* o = g_stamp_buf + 4096 * 16 + (size_t)blockIdx.x * 8;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    o[0] = ((unsigned long long)xcc << 32) | hwid;
    o[1] = tl_start; o[2] = tr0; o[3] = tl_main_end; o[4] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// gemm_direct_kernel: 128 x 128 tile update C -= A B^T with NO LDS: every wave loads its MFMA operand
// fragments straight from global memory in fragment layout (lane&15 -> row of the slice: 128 contiguous
// bytes; lane>>4 -> k: four segments per instruction), PF k-steps ahead in registers.  The two waves that
// share a row slice hit in the vector L1.  No DMA issue stalls, no barriers: waves run independently.
// ------------------------------------------------------------------------------------------------
template <int PF>
__global__ void __launch_bounds__(256, 2)
gemm_direct_kernel(double* __restrict__ C, int ldc, const double* __restrict__ A, int lda, const double* __restrict__ B,
                   int ldb, int M, int Nc, int K, int lower, int skipM, int skipN, const int* __restrict__ tilemap,
                   int ntiles) {
  constexpr int BM = 128, BN = 128;
  int m0, n0;
  if (tilemap) {
    const int nwg = ntiles, b = blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int t = tilemap[v];
    m0 = (t & 0xffff) * BM;
    n0 = (t >> 16) * BN;
  } else {
    m0 = blockIdx.x * BM;
    n0 = blockIdx.y * BN;
    if (lower && m0 + BM <= n0) return;
    if (m0 < skipM && n0 < skipN) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w & 1, wn = w >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const bool wave_active = (m0 + wm * 64) < M;

  d4 acc[4][4];
  double* Cw = C + (size_t)(m0 + wm * 64 + l15) + (size_t)(n0 + wn * 64 + l4) * ldc;
  if (wave_active) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ni][mi][r] = Cw[mi * 16 + (size_t)(ni * 16 + 4 * r) * ldc];
  } else {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = d4{0, 0, 0, 0};
  }
  int rowa = m0 + wm * 64 + l15;
  rowa = wave_active ? rowa : m0 + l15;                    // inactive wave of a ragged tile: recompute valid rows, never store
  const double* pa = A + rowa + (size_t)l4 * lda;
  const double* pb = B + n0 + wn * 64 + l15 + (size_t)l4 * ldb;
  const size_t sa = (size_t)4 * lda, sb = (size_t)4 * ldb;

  double f[PF + 1][8];
#define GPG_DR_LOAD(set)                                                              \
  {                                                                                   \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) f[set][mi] = pa[mi * 16];         \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) f[set][4 + ni] = pb[ni * 16];     \
    pa += sa;                                                                         \
    pb += sb;                                                                         \
  }
#define GPG_DR_MFMA(set)                                                              \
  {                                                                                   \
    double fm[4];                                                                     \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) fm[mi] = -f[set][mi];             \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                   \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                 \
        acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[set][4 + ni], fm[mi], acc[ni][mi], 0, 0, 0); \
  }
  const int nstep = K / 4;   // K is a multiple of 4 (PF + 1): the ring index stays static under the unroll
#pragma unroll
  for (int s = 0; s < PF; ++s) GPG_DR_LOAD(s)
  for (int s0 = 0; s0 < nstep; s0 += PF + 1) {
#pragma unroll
    for (int u = 0; u <= PF; ++u) {
      if (s0 + u + PF < nstep) GPG_DR_LOAD((u + PF) % (PF + 1))
      __builtin_amdgcn_sched_barrier(0);
      GPG_DR_MFMA(u)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef GPG_DR_LOAD
#undef GPG_DR_MFMA
  if (wave_active) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cw[mi * 16 + (size_t)(ni * 16 + 4 * r) * ldc] = acc[ni][mi][r];
  }
}

// direct_tile_gemm_x2: the MFMA loop of gemm_direct_kernel as a device function, with 16-byte fragment loads:
// acc (wave tile 64 x 64) -= A[64 x 4 nstep] B[64 x 4 nstep]^T, operand fragments straight from global memory,
// PF k-steps ahead, no LDS, no barrier; nstep a positive multiple of PF + 1.  Lane lane&15 = t owns the two
// adjacent rows 2t, 2t+1 of each 32-row group of its 64-row slices, so one global_load_dwordx4 feeds two MFMA
// operand blocks: 4 load instructions per k-step instead of 8 (the texture-address unit handles ~4 lanes per
// clock whatever the access width; measured +1.5 %).  acc[2p + e][2g + m][r] of lane (t, l4) <-> tile row 32 g + 2 t + m, tile column
// 32 p + 2 (4 r + l4) + e.  pa / pb: slice + 2 t, k = l4.
template <int PF>
__device__ __forceinline__ void direct_tile_gemm_x2(d4 (&acc)[4][4], const double* pa, int lda, const double* pb, int ldb,
                                                    int nstep) {
  const size_t sa = (size_t)4 * lda, sb = (size_t)4 * ldb;
  double2 fa0[PF + 1], fa1[PF + 1], fb0[PF + 1], fb1[PF + 1];
#define GPG_DX_LOAD(set)                                                              \
  {                                                                                   \
    fa0[set] = *reinterpret_cast<const double2*>(pa);                                 \
    fa1[set] = *reinterpret_cast<const double2*>(pa + 32);                            \
    fb0[set] = *reinterpret_cast<const double2*>(pb);                                 \
    fb1[set] = *reinterpret_cast<const double2*>(pb + 32);                            \
    pa += sa;                                                                         \
    pb += sb;                                                                         \
  }
#define GPG_DX_MFMA(set)                                                              \
  {                                                                                   \
    const double fm[4] = {-fa0[set].x, -fa0[set].y, -fa1[set].x, -fa1[set].y};        \
    const double fn[4] = {fb0[set].x, fb0[set].y, fb1[set].x, fb1[set].y};            \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                   \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                 \
        acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn[ni], fm[mi], acc[ni][mi], 0, 0, 0); \
  }
#pragma unroll
  for (int s = 0; s < PF; ++s) GPG_DX_LOAD(s)
  for (int s0 = 0; s0 < nstep; s0 += PF + 1) {
#pragma unroll
    for (int u = 0; u <= PF; ++u) {
      if (s0 + u + PF < nstep) GPG_DX_LOAD((u + PF) % (PF + 1))
      __builtin_amdgcn_sched_barrier(0);
      GPG_DX_MFMA(u)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef GPG_DX_LOAD
#undef GPG_DX_MFMA
}

// ------------------------------------------------------------------------------------------------
// tile128_chol_kernel: the whole factorisation as ONE dataflow launch over 128 x 128 tiles (left-looking).
// Workgroup b owns tile (i, j) = tasks[b], column-major task order, rows i >= j (the right-hand-side rows
// below the matrix are one more tile row):
//     acc  = A_ij - sum_{k<j} L_ik L_jk^T   the direct-fragment MFMA loop of gemm_direct_kernel over every finished tile
//                                          column, consumed in runs as the flags come up.  The C tile is read
//                                          once and written once per factorisation (the right-looking update
//                                          streams it once per panel) and there is no launch chain at all.
//     i==j : potrf of the 128 x 128 tile inside the workgroup (potrf64, 64-row substitution, 64 x 64 MFMA
//            update, potrf64)
//     i> j : L_ij = acc L_jj^-T, two 64-row passes of panel_solve_rows64 against the 128-wide diagonal tile
//     publish: __threadfence, flag(i, j) = 1 (agent-scope release)
// Progress argument and bounded waits as in tile_chol_kernel.
// ------------------------------------------------------------------------------------------------
// The diagonal tile publishes its pieces as they are final -- L11 (flag_a, after the first potrf64), L21 (flag_c,
// after its 64-row solve), L22 (flag_b = the tile's completion flag): column block 0 of this tile is solved and the
// MFMA update of block 1 runs while the diagonal tile is still in its second potrf64.
__device__ __forceinline__ int tile_solve_rows128(const double* L, int ldl, const double* dinv, double* X, int ldx, double* U,
                                                  double (*Ls)[4][18], double* sdinv, int* flag_a, int* flag_c, int* flag_b,
                                                  int* abort_word, int* info, int* sh) {
  constexpr int SA = 80, BUF = 16 * SA;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int q = tid & 3, rr = tid >> 2;
  const int sp = tid & 31, sk = tid >> 5;
  double x0[16], x1[16], li[16];
  // ---- column block 0 ----------------------------------------------------------------------------------------
  {
    const double* Xr = X + rr + (size_t)q * ldx;     // own rows: in flight while the flag is polled
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      x0[m] = Xr[(size_t)(4 * m) * ldx];
      x1[m] = Xr[64 + (size_t)(4 * m) * ldx];
    }
    if (!wg_wait_flag(flag_a, abort_word, info, sh)) return 0;
    for (int t = tid; t < 64 * 64; t += 256) {
      const int jj = t >> 6, k = t & 63;
      Ls[jj][k & 3][k >> 2] = L[k + (size_t)jj * ldl];
    }
    if (tid < 64) sdinv[tid] = dinv[tid];
  }
  __syncthreads();
  GPG_QUAD_SUBST2(x0, x1, Ls, sdinv, q)
  {
    double* Xr = X + rr + (size_t)q * ldx;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      Xr[(size_t)(4 * m) * ldx] = x0[m];
      Xr[64 + (size_t)(4 * m) * ldx] = x1[m];
    }
  }
  if (!wg_wait_flag(flag_c, abort_word, info, sh)) return 0;   // barrier inside: X1 visible to the workgroup, Ls free
  // ---- column block 1: T2 -= X1 L21^T for the two row halves, each transposed through the LDS tile ---------------
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    d4 acc[4];
    const double* Cw = X + 64 * h + 16 * w + l15 + (size_t)(64 + l4) * ldx;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ni][r] = Cw[(size_t)(ni * 16 + 4 * r) * ldx];
    wave_tile_gemm(acc, X + 64 * h + 2 * sp + (size_t)sk * ldx, ldx, L + 64 + 2 * sp + (size_t)sk * ldl, ldl, 4, U, U + 2 * BUF,
                   w, l15, l4, sp, sk);
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) U[(ni * 16 + 4 * r + l4) * SA + 16 * w + l15] = acc[ni][r];
    __syncthreads();
    const double* Tr = U + q * SA + rr;
    if (h == 0) {
#pragma unroll
      for (int m = 0; m < 16; ++m) x0[m] = Tr[(4 * m) * SA];
    } else {
#pragma unroll
      for (int m = 0; m < 16; ++m) x1[m] = Tr[(4 * m) * SA];
    }
    __syncthreads();   // tile consumed before the next pass stages into U again
  }
  if (!wg_wait_flag(flag_b, abort_word, info, sh)) return 0;
  {   // image of L22
    const double* L22 = L + 64 + (size_t)64 * ldl;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = tid + 256 * i;
      li[i] = L22[(t & 63) + (size_t)(t >> 6) * ldl];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = tid + 256 * i, jj = t >> 6, k = t & 63;
      Ls[jj][k & 3][k >> 2] = li[i];
    }
    if (tid < 64) sdinv[tid] = dinv[64 + tid];
  }
  __syncthreads();
  GPG_QUAD_SUBST2(x0, x1, Ls, sdinv, q)
  {
    double* Xr = X + rr + (size_t)(64 + q) * ldx;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      Xr[(size_t)(4 * m) * ldx] = x0[m];
      Xr[64 + (size_t)(4 * m) * ldx] = x1[m];
    }
  }
  __syncthreads();
  return 1;
}

// Finalisation of a 128 x 128 tile that already sits updated in memory (kept out of line so that its register
// needs do not leak into the MFMA loop of the kernel).  Returns 0 if the wait for the diagonal tile timed out.
__shared__ __attribute__((aligned(16))) double t128_U[4 * 16 * 80];   // staging / transposition tile of the finalisation
__shared__ __attribute__((aligned(16))) double t128_Ls[64][4][18];    // diagonal-block image / potrf scratch
__shared__ double t128_sdinv[64];

__device__ __noinline__ int tile128_finalize(double* A, int ld, size_t r0, size_t cj, int is_diag, int* flag_jj, int* flag_a,
                                             int* flag_c, int* abort_word, double* dinv, int* info, int N) {
  constexpr int SA = 80;
  double* const U = t128_U;
  double (*const Ls)[4][18] = t128_Ls;
  double* const sdinv = t128_sdinv;
  __shared__ int sh_ok;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
#ifdef GPG_STAMP
  unsigned long long* fo = (g_stamp_buf != nullptr && blockIdx.x < 16384) ? g_stamp_buf + 16384 * 8 + (size_t)blockIdx.x * 8 : nullptr;
#define GPG_FS(k) if (tid == 0 && fo) fo[k] = __builtin_amdgcn_s_memrealtime();
#else
#define GPG_FS(k)
#endif
  GPG_FS(0)
  if (is_diag) {
    double* blk = A + cj + cj * (size_t)ld;
    double (*St)[64] = reinterpret_cast<double(*)[64]>(&Ls[0][0][0]);   // potrf scratch over the Ls region
    if (w == 0) {   // A11 was left in the LDS tile by this same wave (no barrier, no trip through memory)
      const int bad = potrf64_wave(U, SA, St, blk, ld, dinv + cj);
      if (bad && lane == 0 && (int)cj + bad - 1 < N) atomicCAS(info, 0, (int)cj + bad);
      __threadfence();   // L11 and its reciprocal pivots (all written by this wave) are published early
      if (lane == 0) __hip_atomic_store(flag_a, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();   // also drains the other waves' stores of A21 / A22
    GPG_FS(1)
    // L21 = A21 L11^-T
    panel_solve_rows64(blk, ld, dinv + cj, blk + 64, ld, 64, 64, U, Ls, sdinv);
    __threadfence();   // L21 is final (every thread stored part of it; the solve ended with a barrier)
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag_c, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    GPG_FS(2)
    // A22 -= L21 L21^T on MFMA, then factor it from the LDS tile
    const int sp = tid & 31, sk = tid >> 5;
    d4 a2[4];
    const double* C2 = blk + 64 + 16 * w + l15 + (size_t)(64 + l4) * ld;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) a2[ni][r] = C2[(size_t)(ni * 16 + 4 * r) * ld];
    const double* g21 = blk + 64 + 2 * sp + (size_t)sk * ld;
    wave_tile_gemm(a2, g21, ld, g21, ld, 4, U, U + 2 * 16 * SA, w, l15, l4, sp, sk);
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) U[(ni * 16 + 4 * r + l4) * SA + 16 * w + l15] = a2[ni][r];
    __syncthreads();
    GPG_FS(3)
    if (w == 0) {
      const int bad = potrf64_wave(U, SA, St, blk + 64 + (size_t)64 * ld, ld, dinv + cj + 64);
      if (bad && lane == 0 && (int)cj + 64 + bad - 1 < N) atomicCAS(info, 0, (int)cj + 64 + bad);
    }
    GPG_FS(4)
    return 1;
  }
  GPG_FS(1)
  const double* Ljj = A + cj + cj * (size_t)ld;
  double* X = A + r0 + cj * (size_t)ld;
  if (!tile_solve_rows128(Ljj, ld, dinv + cj, X, ld, U, Ls, sdinv, flag_a, flag_c, flag_jj, abort_word, info, &sh_ok)) return 0;
  GPG_FS(2)
  GPG_FS(3)
  return 1;
}

__global__ void __launch_bounds__(256, 2)
tile128_chol_kernel(double* A, int ld, int Mt, const int* __restrict__ tasks, int* flags, int* flag_a, int* abort_word,
                    double* __restrict__ dinv, int* __restrict__ info, int N) {
  __shared__ int sh_kr;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w & 1, wn = w >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int task = tasks[blockIdx.x];
  const int ti = task & 0xffff, tj = task >> 16;
  const size_t r0 = 128 * (size_t)ti, cj = 128 * (size_t)tj;
  int* const frow_i = flags + (size_t)ti * Mt;
  int* const frow_j = flags + (size_t)tj * Mt;

#ifdef GPG_STAMP
  const unsigned long long tk_start = __builtin_amdgcn_s_memrealtime();
  unsigned long long tk_spin = 0, tk_gemm = 0, tk_runs = 0;
#endif
  // accumulator layout of direct_tile_gemm_x2: acc[2p + e][2g + m][r] <-> row 32 g + 2 l15 + m, column 32 p + 2 (4 r + l4) + e
  d4 acc[4][4];
  double* Cw = A + r0 + wm * 64 + 2 * l15 + (cj + wn * 64 + 2 * l4) * (size_t)ld;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double2 v = *reinterpret_cast<const double2*>(Cw + 32 * g + (size_t)(32 * (ni >> 1) + 8 * r + (ni & 1)) * ld);
        acc[ni][2 * g][r] = v.x;
        acc[ni][2 * g + 1][r] = v.y;
      }

  // ---- (1) left-looking accumulation ----------------------------------------------------------------------------
  int kdone = 0;
  while (kdone < tj) {
    GPG_T(q0)
    if (tid == 0) {
      int kr = kdone;
      const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        while (kr < tj && __hip_atomic_load(frow_i + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 &&
               __hip_atomic_load(frow_j + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
          ++kr;
        if (kr > kdone) break;
        if (__builtin_amdgcn_s_memrealtime() - t_wait > GPG_TILE_WAIT_TICKS || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicMax(info, GPG_INFO_INTERNAL);
          kr = -1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      sh_kr = kr;
    }
    __syncthreads();
    const int kr = sh_kr;
    if (kr < 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    GPG_T(q1)
    const size_t ck = 128 * (size_t)kdone;
    direct_tile_gemm_x2<3>(acc, A + r0 + wm * 64 + 2 * l15 + (ck + l4) * (size_t)ld, ld,
                           A + cj + wn * 64 + 2 * l15 + (ck + l4) * (size_t)ld, ld, 32 * (kr - kdone));
    __syncthreads();   // sh_kr may be rewritten
    GPG_T(q2)
#ifdef GPG_STAMP
    tk_spin += q1 - q0; tk_gemm += q2 - q1; ++tk_runs;
#endif
    kdone = kr;
  }

#ifdef GPG_STAMP
  const unsigned long long tk_fin0 = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- (2) the updated tile goes back to memory; the finalisation works on it in place.  Diagonal tile: the
  //      top-left 64 x 64 block goes straight into the LDS tile its own wave factors next, the strictly upper
  //      block is dropped. ---------------------------------------------------------------------------------------
  if (ti == tj && w == 0) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          t128_U[(32 * (ni >> 1) + 8 * r + 2 * l4 + (ni & 1)) * 80 + 32 * (mi >> 1) + 2 * l15 + (mi & 1)] = acc[ni][mi][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else if (!(ti == tj && wm == 0)) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double2 v;
          v.x = acc[ni][2 * g][r];
          v.y = acc[ni][2 * g + 1][r];
          *reinterpret_cast<double2*>(Cw + 32 * g + (size_t)(32 * (ni >> 1) + 8 * r + (ni & 1)) * ld) = v;
        }
  }
  if (ti != tj) __syncthreads();
  if (tile128_finalize(A, ld, r0, cj, ti == tj, frow_j + tj, flag_a + tj, flag_a + Mt + tj, abort_word, dinv, info, N) == 0) return;
  // ---- (3) publish ----------------------------------------------------------------------------------------------
  __threadfence();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(frow_i + tj, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#ifdef GPG_STAMP
  if (tid == 0 && g_stamp_buf != nullptr && blockIdx.x < 16384) {
    unsigned long long* o = g_stamp_buf + (size_t)blockIdx.x * 8;
    o[0] = tk_start; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = tk_spin; o[3] = tk_gemm; o[4] = tk_runs;
    o[5] = tk_fin0; o[6] = (unsigned long long)task;
  }
#endif
}

template <int BM, int BN>
void launch_gemm(gpg_ctx* c, double* C, int ldc, const double* A, int lda, const double* B, int ldb, int M, int Nc,
                 int K, int lower, int skipM = 0, int skipN = 0) {
  if (M <= 0 || Nc <= 0 || K <= 0) return;
  dim3 grid((M + BM - 1) / BM, Nc / BN);
  if (BM == 128 && BN == 128 && c->gemm_impl == 1) {
    hipLaunchKernelGGL(gemm_dma_kernel<4>, grid, dim3(256), 0, c->stream, C, ldc, A, lda, B, ldb, M, Nc, K, lower, skipM,
                       skipN, (const int*)nullptr, 0);
    return;
  }
  hipLaunchKernelGGL((gemm_nt_minus_kernel<BM, BN>), grid, dim3(256), 0, c->stream, C, ldc, A, lda, B, ldb, M, Nc, K,
                     lower, skipM, skipN, (const int*)nullptr, 0);
}

// Live-tile list of a lower-trapezoid update (Mt x Nt tiles of 128, tiles above the diagonal and the
// skipT x skipT leading block dropped), ordered by 4 x 4 super-tiles.  Built once per shape and kept on
// the device for the lifetime of the context.
const TileMap& get_tilemap(gpg_ctx* c, int Mt, int Nt, int skipT) {
  const unsigned long long key = ((unsigned long long)Mt << 40) | ((unsigned long long)Nt << 16) | (unsigned)skipT;
  auto it = c->tilemaps.find(key);
  if (it != c->tilemaps.end()) return it->second;
  constexpr int S = 4;
  std::vector<int> list;
  list.reserve((size_t)Mt * Nt / 2 + Mt);
  for (int sj = 0; sj * S < Nt; ++sj)
    for (int si = sj; si * S < Mt; ++si)
      for (int ti = si * S; ti < (si + 1) * S && ti < Mt; ++ti)
        for (int tj = sj * S; tj < (sj + 1) * S && tj < Nt; ++tj) {
          if (tj > ti) continue;                       // above the diagonal
          if (ti < skipT && tj < skipT) continue;      // owned by the look-ahead stream
          list.push_back(ti | (tj << 16));
        }
  TileMap tm;
  tm.n = (int)list.size();
  tm.dev = nullptr;
  if (tm.n > 0) {
    (void)hipMalloc(&tm.dev, sizeof(int) * list.size());
    (void)hipMemcpy(tm.dev, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice);
  }
  return c->tilemaps.emplace(key, tm).first->second;
}

void launch_gemm_trailing(gpg_ctx* c, double* C, int ldc, const double* A, int lda, const double* B, int ldb, int M,
                          int Nc, int K, int skip) {
  const TileMap& tm = get_tilemap(c, (M + 127) / 128, Nc / 128, skip / 128);
  if (tm.n <= 0) return;
  if (c->gemm_impl == 1) {
    hipLaunchKernelGGL(gemm_dma_kernel<4>, dim3(tm.n), dim3(256), 0, c->stream, C, ldc, A, lda, B, ldb, M, Nc, K, 1, 0, 0,
                       (const int*)tm.dev, tm.n);
    return;
  }
  if (c->gemm_impl == 3) {
    hipLaunchKernelGGL(gemm_direct_kernel<3>, dim3(tm.n), dim3(256), 0, c->stream, C, ldc, A, lda, B, ldb, M, Nc, K, 1, 0, 0,
                       (const int*)tm.dev, tm.n);
    return;
  }
  hipLaunchKernelGGL((gemm_nt_minus_kernel<128, 128>), dim3(tm.n), dim3(256), 0, c->stream, C, ldc, A, lda, B, ldb, M, Nc,
                     K, 1, 0, 0, (const int*)tm.dev, tm.n);
}

// Column-major task list of the dataflow factorisation (Mt tile columns, Rt >= Mt tile rows), cached per shape.
const TileMap& get_tile_tasks(gpg_ctx* c, int Mt, int Rt) {
  const unsigned long long key = (1ull << 63) | ((unsigned long long)Mt << 20) | (unsigned)Rt;
  auto it = c->tilemaps.find(key);
  if (it != c->tilemaps.end()) return it->second;
  std::vector<int> list;
  for (int j = 0; j < Mt; ++j)
    for (int i = j; i < Rt; ++i) list.push_back(i | (j << 16));
  TileMap tm;
  tm.n = (int)list.size();
  tm.dev = nullptr;
  (void)hipMalloc(&tm.dev, sizeof(int) * list.size());
  (void)hipMemcpy(tm.dev, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice);
  return c->tilemaps.emplace(key, tm).first->second;
}

// Factor A[c0:, c0:] (and carry the rows below the matrix) with the dataflow kernel, on c->stream.
static void launch_tile_chol(gpg_ctx* c, int c0) {
  const int Mt = (c->Npad - c0) / 64, Rt = (c->ld - c0) / 64;
  if (Mt <= 0) return;
  const TileMap& tm = get_tile_tasks(c, Mt, Rt);
  const size_t nflag = (size_t)Mt * Rt + 1 + 4 * (size_t)Mt;   // tile flags, abort word, four piece flags per diagonal tile
  if (c->tile_flags_cap < nflag) {
    if (c->tile_flags) (void)hipFree(c->tile_flags);
    (void)hipMalloc(&c->tile_flags, sizeof(int) * nflag);
    c->tile_flags_cap = nflag;
  }
  (void)hipMemsetAsync(c->tile_flags, 0, sizeof(int) * nflag, c->stream);
  const double m = (double)(c->Npad - c0);
  gpg_prof_begin(c, GPG_PROF_GEMM_TRAIL, m * m * m / 3.0);
  int* abort_word = c->tile_flags + (size_t)Mt * Rt;
  hipLaunchKernelGGL(tile_chol_kernel, dim3(tm.n), dim3(256), 0, c->stream, c->A, c->ld, c0, Mt, (const int*)tm.dev,
                     c->tile_flags, abort_word + 1, abort_word, c->dinv, c->info, c->N);
  gpg_prof_end(c);
}

// The whole matrix with the 128-tile dataflow kernel, on c->stream.
static void launch_tile128_chol(gpg_ctx* c) {
  const int Mt = c->Npad / 128, Rt = c->ld / 128;
  const TileMap& tm = get_tile_tasks(c, Mt, Rt);
  const size_t nflag = (size_t)Mt * Rt + 1 + 2 * Mt;   // tile flags, abort word, two early flags per diagonal tile
  if (c->tile_flags_cap < nflag) {
    if (c->tile_flags) (void)hipFree(c->tile_flags);
    (void)hipMalloc(&c->tile_flags, sizeof(int) * nflag);
    c->tile_flags_cap = nflag;
  }
  (void)hipMemsetAsync(c->tile_flags, 0, sizeof(int) * nflag, c->stream);
  const double m = (double)c->Npad;
  gpg_prof_begin(c, GPG_PROF_GEMM_TRAIL, m * m * m / 3.0);
  int* abort_word = c->tile_flags + (size_t)Mt * Rt;
  hipLaunchKernelGGL(tile128_chol_kernel, dim3(tm.n), dim3(256), 0, c->stream, c->A, c->ld, Mt, (const int*)tm.dev,
                     c->tile_flags, abort_word + 1, abort_word, c->dinv, c->info, c->N);
  gpg_prof_end(c);
}

}  // namespace

// Factorise the Npad x Npad matrix held in c->A; rows [Npad, ld) are right-hand-side rows.
//
// Per panel p (columns [k0, k1), width nb_outer):
//   D_p  factor the nb x nb diagonal block               (potrf64 + small trsm/gemm; latency-bound, tiny)
//   B_p  solve the rows below it: 64-column trsm sweeps + MFMA updates inside the panel
//   U_p  trailing update C -= A_panel A_panel^T on fp64 MFMA
// Look-ahead: D_(p+1) only needs the next diagonal block updated, so a second (high-priority) stream
// applies panel p to that block and factors it while the main stream runs the big U_p (which skips
// that block).  The serial pivot chain of potrf is thereby hidden behind the trailing update.
//   main stream : [wait D_p] B_p, U_p (minus next diag block), [wait D_(p+1)] B_(p+1), ...
//   diag stream : [wait B_p] update next diag block with panel p, D_(p+1)
static void factor_diag_block(gpg_ctx* c, int k0, int k1) {
  const int ld = c->ld;
  double* A = c->A;
  for (int j0 = k0; j0 < k1; j0 += GPG_NBI) {
    const int j1 = j0 + GPG_NBI;
    gpg_prof_begin(c, GPG_PROF_POTRF, 64.0 * 64.0 * 64.0 / 3.0);
    hipLaunchKernelGGL(potrf64_kernel, dim3(1), dim3(64), 0, c->stream, A, ld, j0, c->N, c->info, c->dinv);
    gpg_prof_end(c);
    const int rows = k1 - j1;
    if (rows > 0) {
      gpg_prof_begin(c, GPG_PROF_POTRF, (double)rows * 64.0 * 64.0);
      hipLaunchKernelGGL(trsm64_kernel, dim3((rows + 63) / 64), dim3(256), 0, c->stream, A + (size_t)j0 + (size_t)j0 * ld,
                         ld, c->dinv + j0, A + (size_t)j1 + (size_t)j0 * ld, ld, rows);
      launch_gemm<64, 64>(c, A + (size_t)j1 + (size_t)j1 * ld, ld, A + (size_t)j1 + (size_t)j0 * ld, ld,
                          A + (size_t)j1 + (size_t)j0 * ld, ld, rows, rows, GPG_NBI, 1);
      gpg_prof_end(c);
    }
  }
}

static void launch_panel_solve(gpg_ctx* c, const double* Lpp, int ldl, const double* dinv, double* X, int ldx, int rows,
                               int nb) {
  if (rows <= 0 || nb <= 0) return;
  hipLaunchKernelGGL(panel_solve_kernel, dim3((rows + 63) / 64), dim3(256), 0, c->stream, Lpp, ldl, dinv, X, ldx, rows, nb);
}

static void solve_below_block(gpg_ctx* c, int k0, int k1) {
  const int ld = c->ld;
  double* A = c->A;
  const int rows = ld - k1;
  if (rows <= 0) return;
  if (c->panel_impl == 1) {   // one fused launch per panel
    gpg_prof_begin(c, GPG_PROF_TRSM, (double)rows * (double)(k1 - k0) * (double)(k1 - k0));
    launch_panel_solve(c, A + (size_t)k0 + (size_t)k0 * ld, ld, c->dinv + k0, A + (size_t)k1 + (size_t)k0 * ld, ld, rows,
                       k1 - k0);
    gpg_prof_end(c);
    return;
  }
  for (int j0 = k0; j0 < k1; j0 += GPG_NBI) {
    const int j1 = j0 + GPG_NBI;
    gpg_prof_begin(c, GPG_PROF_TRSM, (double)rows * 64.0 * 64.0);
    hipLaunchKernelGGL(trsm64_kernel, dim3((rows + 63) / 64), dim3(256), 0, c->stream, A + (size_t)j0 + (size_t)j0 * ld, ld,
                       c->dinv + j0, A + (size_t)k1 + (size_t)j0 * ld, ld, rows);
    gpg_prof_end(c);
    const int ncols = k1 - j1;
    if (ncols > 0) {   // X[:, j1:k1] -= X[:, j0:j1] L[j1:k1, j0:j1]^T
      gpg_prof_begin(c, GPG_PROF_GEMM_PANEL, 2.0 * (double)rows * ncols * 64.0);
      launch_gemm<128, 64>(c, A + (size_t)k1 + (size_t)j1 * ld, ld, A + (size_t)k1 + (size_t)j0 * ld, ld,
                           A + (size_t)j1 + (size_t)j0 * ld, ld, rows, ncols, GPG_NBI, 0);
      gpg_prof_end(c);
    }
  }
}

void gpg_cholesky(gpg_ctx* c) {
  const int ld = c->ld, Npad = c->Npad;
  double* A = c->A;
  hipStream_t sM = c->stream, sD = c->lookahead ? c->stream_upd : c->stream;
  const bool two = (sD != sM);
  // panel boundaries: wide panels (nb_big) while at least big_rows columns remain -- the trailing update then
  // streams the C tiles half as often -- and nb_outer afterwards, where the serial diagonal chain matters more
  std::vector<int> kb;
  for (int k = 0; k < Npad;) {
    kb.push_back(k);
    const int nbw = (c->nb_big > c->nb_outer && Npad - k >= c->big_rows) ? c->nb_big : c->nb_outer;
    k = (k + nbw < Npad) ? k + nbw : Npad;
  }
  kb.push_back(Npad);
  const int npanel = (int)kb.size() - 1;
  while ((int)c->ev_panel.size() < npanel + 1) {
    hipEvent_t e1, e2;
    (void)hipEventCreateWithFlags(&e1, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&e2, hipEventDisableTiming);
    c->ev_panel.push_back(e1);   // ev_panel[p]: diagonal block p factorised (diag stream)
    c->ev_upd.push_back(e2);     // ev_upd[p]  : rows below panel p solved (main stream)
  }
  // D_0 follows the assembly on the main stream
  c->stream = sM;
  if (c->tail_cols > 0 && Npad <= c->tail_cols) {   // small matrix: the dataflow kernel does all of it
    launch_tile_chol(c, 0);
    return;
  }
  if (c->chol_impl == 1) {
    launch_tile128_chol(c);
    return;
  }
  factor_diag_block(c, 0, kb[1]);
  for (int p = 0; p < npanel; ++p) {
    const int k0 = kb[p], k1 = kb[p + 1];                           // panel p = columns [k0, k1)
    c->stream = sM;
    if (two && p > 0) (void)hipStreamWaitEvent(sM, c->ev_panel[p], 0);
    solve_below_block(c, k0, k1);                                   // B_p (also carries the RHS rows)
    if (k1 >= Npad) break;
    if (c->tail_cols > 0 && Npad - k1 <= c->tail_cols) {
      // the rest is latency-bound: apply panel p to all of it, then hand over to the dataflow kernel
      const double nt = (double)(Npad - k1);
      gpg_prof_begin(c, GPG_PROF_GEMM_TRAIL, nt * (nt + 1.0) * (double)(k1 - k0));
      launch_gemm_trailing(c, A + (size_t)k1 + (size_t)k1 * ld, ld, A + (size_t)k0 * ld + k1, ld, A + (size_t)k0 * ld + k1, ld,
                           ld - k1, Npad - k1, k1 - k0, 0);
      gpg_prof_end(c);
      launch_tile_chol(c, k1);
      break;
    }
    const int k2 = kb[p + 2];                                       // next diagonal block = [k1, k2)
    const int K = k1 - k0;
    const double* Ap = A + (size_t)k0 * ld;
    const double w = (double)(k2 - k1), nt = (double)(Npad - k1);
    const double flops_all = nt * (nt + 1.0) * (double)K;           // lower triangle of the trailing block
    const double flops_diag = w * (w + 1.0) * (double)K;
    if (two) {
      (void)hipEventRecord(c->ev_upd[p], sM);
      // diag stream: next diagonal block -= its rows of panel p, then D_(p+1)
      c->stream = sD;
      (void)hipStreamWaitEvent(sD, c->ev_upd[p], 0);
      gpg_prof_begin(c, GPG_PROF_POTRF, flops_diag);
      launch_gemm<64, 64>(c, A + (size_t)k1 + (size_t)k1 * ld, ld, Ap + k1, ld, Ap + k1, ld, k2 - k1, k2 - k1, K, 1);
      gpg_prof_end(c);
      factor_diag_block(c, k1, k2);
      (void)hipEventRecord(c->ev_panel[p + 1], sD);
      // main stream: the rest of the trailing update
      c->stream = sM;
      gpg_prof_begin(c, GPG_PROF_GEMM_TRAIL, flops_all - flops_diag);
      launch_gemm_trailing(c, A + (size_t)k1 + (size_t)k1 * ld, ld, Ap + k1, ld, Ap + k1, ld, ld - k1, Npad - k1, K, k2 - k1);
      gpg_prof_end(c);
    } else {
      gpg_prof_begin(c, GPG_PROF_GEMM_TRAIL, flops_all);
      launch_gemm_trailing(c, A + (size_t)k1 + (size_t)k1 * ld, ld, Ap + k1, ld, Ap + k1, ld, ld - k1, Npad - k1, K, 0);
      gpg_prof_end(c);
      factor_diag_block(c, k1, k2);
    }
  }
  c->stream = sM;
}

// W (rows x Npad, leading dimension ldw, "RHS rows" layout) <- W L^-T using the factor in c->A.
void gpg_forward_rows(gpg_ctx* c, double* W, int ldw, int rows) {
  const int ld = c->ld, Npad = c->Npad, NB = c->nb_outer;
  const double* A = c->A;
  for (int k0 = 0; k0 < Npad; k0 += NB) {
    const int nbw = (Npad - k0) < NB ? (Npad - k0) : NB;
    const int k1 = k0 + nbw;
    if (c->panel_impl == 1) {
      launch_panel_solve(c, A + (size_t)k0 + (size_t)k0 * ld, ld, c->dinv + k0, W + (size_t)k0 * ldw, ldw, rows, nbw);
    } else {
      for (int j0 = k0; j0 < k1; j0 += GPG_NBI) {
        const int j1 = j0 + GPG_NBI;
        hipLaunchKernelGGL(trsm64_kernel, dim3((rows + 63) / 64), dim3(256), 0, c->stream,
                           A + (size_t)j0 + (size_t)j0 * ld, ld, c->dinv + j0, W + (size_t)j0 * ldw, ldw, rows);
        const int ncols = k1 - j1;
        if (ncols > 0)
          launch_gemm<128, 64>(c, W + (size_t)j1 * ldw, ldw, W + (size_t)j0 * ldw, ldw,
                               A + (size_t)j1 + (size_t)j0 * ld, ld, rows, ncols, GPG_NBI, 0);
      }
    }
    if (k1 < Npad)
      launch_gemm<128, 128>(c, W + (size_t)k1 * ldw, ldw, W + (size_t)k0 * ldw, ldw, A + (size_t)k1 + (size_t)k0 * ld,
                            ld, rows, Npad - k1, nbw, 0);
  }
}

// Minv <- -(L L^T)^-1 (lower triangle) from the factor in c->A:
//   (1) W = I L^-T by a forward sweep over the "RHS rows" layout.  W is upper triangular (W[r, c] = 0 for c < r),
//       so panel [k0, k1) only touches rows [0, k1): about N^3/3 flop, the cost of the factorisation.
//   (2) Minv[0:k1, 0:k1] -= W[0:k1, k0:k1] W[0:k1, k0:k1]^T panel by panel with the trailing-update kernel
//       (lower tiles only): another N^3/3.
// Replaces adj_ln_detK = cho_solve(chofac, eye(N)) of the reference (CalcLkd.py:174,234).
void gpg_inverse_from_factor(gpg_ctx* c, double* W, double* Minv) {
  const int ld = c->ld, Npad = c->Npad, NB = c->nb_outer, ldw = c->Npad;
  const double* A = c->A;
  gpg_launch_identity(c, W, ldw);
  (void)hipMemsetAsync(Minv, 0, sizeof(double) * (size_t)ldw * Npad, c->stream);
  for (int k0 = 0; k0 < Npad; k0 += NB) {
    const int k1 = (k0 + NB < Npad) ? k0 + NB : Npad;
    if (c->panel_impl == 1) {   // rows [j1, k1) of a column block are still zero when it is solved: harmless extra rows
      launch_panel_solve(c, A + (size_t)k0 + (size_t)k0 * ld, ld, c->dinv + k0, W + (size_t)k0 * ldw, ldw, k1, k1 - k0);
    } else {
      for (int j0 = k0; j0 < k1; j0 += GPG_NBI) {
        const int j1 = j0 + GPG_NBI;
        hipLaunchKernelGGL(trsm64_kernel, dim3(j1 / 64), dim3(256), 0, c->stream, A + (size_t)j0 + (size_t)j0 * ld, ld,
                           c->dinv + j0, W + (size_t)j0 * ldw, ldw, j1);
        const int ncols = k1 - j1;
        if (ncols > 0)
          launch_gemm<128, 64>(c, W + (size_t)j1 * ldw, ldw, W + (size_t)j0 * ldw, ldw, A + (size_t)j1 + (size_t)j0 * ld, ld,
                               j1, ncols, GPG_NBI, 0);
      }
    }
    if (k1 < Npad)
      launch_gemm<128, 128>(c, W + (size_t)k1 * ldw, ldw, W + (size_t)k0 * ldw, ldw, A + (size_t)k1 + (size_t)k0 * ld, ld,
                            k1, Npad - k1, k1 - k0, 0);
    launch_gemm_trailing(c, Minv, ldw, W + (size_t)k0 * ldw, ldw, W + (size_t)k0 * ldw, ldw, k1, k1, k1 - k0, 0);
  }
}
