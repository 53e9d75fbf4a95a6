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

__global__ void __launch_bounds__(64) potrf64_kernel(double* __restrict__ A, int ld, int j0, int N, int* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) double St[64][64];   // St[k][i] = L[i][k]
  const int i = threadIdx.x;
  double* blk = A + (size_t)j0 + (size_t)j0 * ld;
  int bad = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = blk[i + (size_t)(16 * s + c) * ld];
#pragma unroll 2
    for (int k = 0; k < 16 * s; ++k) {
      const double lik = St[k][i];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] -= lik * St[k][16 * s + c];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const double ajj = readlane_d(a[c], 16 * s + c);
      bad = (bad == 0 && !(ajj > 0.0)) ? 16 * s + c + 1 : bad;   // first non-positive / NaN pivot (LAPACK info)
      const double dj = sqrt(ajj);
      const double inv = 1.0 / dj;
      a[c] = (i == 16 * s + c) ? dj : a[c] * inv;
#pragma unroll
      for (int k2 = c + 1; k2 < 16; ++k2) a[k2] -= a[c] * readlane_d(a[c], 16 * s + k2);
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      St[16 * s + c][i] = a[c];
      if (i >= 16 * s + c) blk[i + (size_t)(16 * s + c) * ld] = a[c];
    }
    __syncthreads();
  }
  if (bad && i == 0 && j0 + bad - 1 < N) atomicCAS(info, 0, j0 + bad);
}

// ------------------------------------------------------------------------------------------------
// trsm64: X <- X L^-T for the `rows` rows below a factorised 64 x 64 diagonal block (X is rows x 64).
// Lane <-> row; L^T is staged in LDS and read with wave-uniform (broadcast) addresses.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) trsm64_kernel(const double* __restrict__ Lblk, int ldl, double* __restrict__ X,
                                                     int ldx, int rows) {
  __shared__ __attribute__((aligned(16))) double Lt[64][64];  // Lt[j][k] = L[k][j]
  __shared__ double invd[64];
  for (int t = threadIdx.x; t < 64 * 64; t += 256) {
    int j = t >> 6, k = t & 63;
    Lt[j][k] = Lblk[k + (size_t)j * ldl];
  }
  if (threadIdx.x < 64) invd[threadIdx.x] = 1.0 / Lblk[threadIdx.x + (size_t)threadIdx.x * ldl];
  __syncthreads();
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  double x[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) x[j] = X[r + (size_t)j * ldx];
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    x[j] *= invd[j];
#pragma unroll
    for (int k = j + 1; k < 64; ++k) x[k] -= x[j] * Lt[j][k];
  }
#pragma unroll
  for (int j = 0; j < 64; ++j) X[r + (size_t)j * ldx] = x[j];
}

// ------------------------------------------------------------------------------------------------
// gemm_nt_minus<BM, BN>:  C[M x Nc] -= A[M x K] * B[Nc x K]^T   (all column-major)
//   M multiple of 64, Nc multiple of BN, K multiple of 16.  lower != 0: C's origin lies on the matrix
//   diagonal and tiles entirely above it are skipped.
// 4 waves as 2 x 2; each wave owns (BM/2) x (BN/2) of C as 16x16 MFMA blocks.  The MFMA "A" operand
// is fed from the C-column side and the "B" operand from the C-row side, so that lane&15 indexes C's
// row: every accumulator load/store instruction touches 4 columns x 128 contiguous bytes.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ void __launch_bounds__(256, 2)
gemm_nt_minus_kernel(double* __restrict__ C, int ldc, const double* __restrict__ A, int lda,
                     const double* __restrict__ B, int ldb, int M, int Nc, int K, int lower) {
  constexpr int KB = 8;
  constexpr int SA = BM + 16, SB = BN + 16;   // row strides: +128 B keeps ds_read_b64 conflict-free
  constexpr int MI = BM / 32, NI = BN / 32;   // 16x16 blocks per wave
  constexpr int LA = BM * KB / 512, LB = BN * KB / 512;  // double2 loads per thread per chunk
  __shared__ __attribute__((aligned(16))) double sA[2][KB][SA];
  __shared__ __attribute__((aligned(16))) double sB[2][KB][SB];

  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  if (lower && m0 + BM <= n0) return;

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

template <int BM, int BN>
void launch_gemm(gpg_ctx* c, double* C, int ldc, const double* A, int lda, const double* B, int ldb, int M, int Nc,
                 int K, int lower) {
  if (M <= 0 || Nc <= 0 || K <= 0) return;
  dim3 grid((M + BM - 1) / BM, Nc / BN);
  hipLaunchKernelGGL((gemm_nt_minus_kernel<BM, BN>), grid, dim3(256), 0, c->stream, C, ldc, A, lda, B, ldb, M, Nc, K,
                     lower);
}

}  // namespace

// Factorise the Npad x Npad matrix held in c->A; rows [Npad, ld) are right-hand-side rows.
void gpg_cholesky(gpg_ctx* c) {
  const int ld = c->ld, Npad = c->Npad, NB = c->nb_outer;
  double* A = c->A;
  for (int k0 = 0; k0 < Npad; k0 += NB) {
    const int nbw = (Npad - k0) < NB ? (Npad - k0) : NB;
    const int k1 = k0 + nbw;
    for (int j0 = k0; j0 < k1; j0 += GPG_NBI) {
      const int j1 = j0 + GPG_NBI;
      gpg_prof_begin(c, GPG_PROF_POTRF, 64.0 * 64.0 * 64.0 / 3.0);
      hipLaunchKernelGGL(potrf64_kernel, dim3(1), dim3(64), 0, c->stream, A, ld, j0, c->N, c->info);
      gpg_prof_end(c);
      const int rows = ld - j1;
      if (rows > 0) {
        gpg_prof_begin(c, GPG_PROF_TRSM, (double)rows * 64.0 * 64.0);
        hipLaunchKernelGGL(trsm64_kernel, dim3((rows + 255) / 256), dim3(256), 0, c->stream,
                           A + (size_t)j0 + (size_t)j0 * ld, ld, A + (size_t)j1 + (size_t)j0 * ld, ld, rows);
        gpg_prof_end(c);
      }
      const int ncols = k1 - j1;
      if (ncols > 0) {
        gpg_prof_begin(c, GPG_PROF_GEMM_PANEL, 2.0 * (double)rows * ncols * 64.0);
        launch_gemm<128, 64>(c, A + (size_t)j1 + (size_t)j1 * ld, ld, A + (size_t)j1 + (size_t)j0 * ld, ld,
                             A + (size_t)j1 + (size_t)j0 * ld, ld, rows, ncols, GPG_NBI, 1);
        gpg_prof_end(c);
      }
    }
    if (k1 < Npad) {
      const int M = ld - k1, Nc = Npad - k1;
      // algorithmic flops of this launch: lower triangle of the Nt x Nt trailing block, K = nbw
      const double nt = (double)(Npad - k1);
      gpg_prof_begin(c, GPG_PROF_GEMM_TRAIL, nt * (nt + 1.0) * (double)nbw);
      launch_gemm<128, 128>(c, A + (size_t)k1 + (size_t)k1 * ld, ld, A + (size_t)k1 + (size_t)k0 * ld, ld,
                            A + (size_t)k1 + (size_t)k0 * ld, ld, M, Nc, nbw, 1);
      gpg_prof_end(c);
    }
  }
}

// W (rows x Npad, leading dimension ldw, "RHS rows" layout) <- W L^-T using the factor in c->A.
void gpg_forward_rows(gpg_ctx* c, double* W, int ldw, int rows) {
  const int ld = c->ld, Npad = c->Npad, NB = c->nb_outer;
  const double* A = c->A;
  for (int k0 = 0; k0 < Npad; k0 += NB) {
    const int nbw = (Npad - k0) < NB ? (Npad - k0) : NB;
    const int k1 = k0 + nbw;
    for (int j0 = k0; j0 < k1; j0 += GPG_NBI) {
      const int j1 = j0 + GPG_NBI;
      hipLaunchKernelGGL(trsm64_kernel, dim3((rows + 255) / 256), dim3(256), 0, c->stream,
                         A + (size_t)j0 + (size_t)j0 * ld, ld, W + (size_t)j0 * ldw, ldw, rows);
      const int ncols = k1 - j1;
      if (ncols > 0)
        launch_gemm<128, 64>(c, W + (size_t)j1 * ldw, ldw, W + (size_t)j0 * ldw, ldw,
                             A + (size_t)j1 + (size_t)j0 * ld, ld, rows, ncols, GPG_NBI, 0);
    }
    if (k1 < Npad)
      launch_gemm<128, 128>(c, W + (size_t)k1 * ldw, ldw, W + (size_t)k0 * ldw, ldw, A + (size_t)k1 + (size_t)k0 * ld,
                            ld, rows, Npad - k1, nbw, 0);
  }
}
