// Cholesky entry point (gpg_cholesky) and the BLOCKED right-looking schedule for gfx950, fp64, column-major lower
// triangle, in place.  Replaces scipy.linalg.cho_factor(Kcov_precon, lower=True) (reference Kernel.py:251; LAPACK
// dpotrf).  The default schedule is the dataflow one of cholesky_dataflow.hip (one launch per factorisation); what
// lives here is used by GPG_FACTOR_BLOCKED (A/B measurements, fallback when dataflow launches share a device) and by
// the forward sweeps of the prediction / gradient paths.
//
// Blocked schedule: panel (width nb_outer) = 64-wide inner steps of
//     potrf64   one wave, a matrix row per lane held in registers
//     trsm64    X <- X L_kk^-T, four lanes per matrix row (rows are contiguous in column-major -> coalesced)
//     gemm      update of the remaining panel columns        (B_p as one fused launch: panel_solve_kernel)
// followed by the trailing update  C -= A_panel A_panel^T  on v_mfma_f64_16x16x4_f64
// (64 cycles / instruction / SIMD measured = 78.6 TFLOP/s chip peak), 128x128 tiles, LDS-DMA ring or direct
// fragment loads, the C tile loaded straight into the accumulators.
// Right-hand-side rows stored below the matrix (rows Npad .. ld) ride along every trsm / gemm, which
// yields L^-1 B for free (no separate forward substitution on the likelihood path).
#include "chol_device.h"

namespace {

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
  // the quad-row substitution of chol_device.h (the L values of step j + 1 are read from LDS while step j computes)
  GPG_QUAD_SUBST(x, Ls, sdinv, q)
  if (active) {
#pragma unroll
    for (int m = 0; m < 16; ++m) Xr[(size_t)(4 * m) * ldx] = x[m];
  }
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
    if (!gpg_dev_alloc(c, &tm.dev, sizeof(int) * list.size())) {
      static const TileMap none{nullptr, 0};     // nothing to launch; the API call reports the failure
      return none;
    }
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
    gpg_launch_tile_chol(c, 0);
    return;
  }
  if (c->chol_impl == 1) {
    gpg_launch_tile128_chol(c);
    return;
  }
  c->last_factor_kernel = 0; c->last_factor_batch = 1;
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
      gpg_launch_tile_chol(c, k1);
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
void gpg_forward_rows(gpg_ctx* c, double* W, int ldw, int rows, int valid) {
  // a few row tiles (posterior evaluation): latency-bound -> one dataflow launch, unless dataflow is switched off
  if (gpg_dataflow_solves(c) && gpg_launch_rows_fwd(c, W, ldw, rows, valid)) return;
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
  // dataflow schedule: W = L^-T and Minv = -W W^T as one launch each (128 x 128 tiles); the blocked sweeps below remain
  // for the blocked factor mode (A/B runs, fallback after a timed-out wait)
  if (c->chol_impl != 0 || c->tail_cols != 0) {
    if (gpg_launch_tile128_inverse(c, W, Minv)) return;
  }
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
