// Device-side building blocks shared by the blocked (cholesky.hip) and the dataflow (cholesky_dataflow.hip)
// factorisations: potrf64 core, quad-row substitution, MFMA tile loops, flag wait.  gfx950 only.
#pragma once
#include "gpg_internal.h"

// Consumer side of a flag: what a workgroup reads after seeing a flag (a finished tile, its reciprocal pivots) is
// memory that neither its CU nor its XCD has read before in this launch -- nobody reads a tile before its flag, and
// tiles / pivot groups do not share cache lines -- so no stale copy can sit in the L1 / L2 it reads through and the
// agent-scope acquire's `buffer_inv sc1` (invalidate the XCD's whole L2, on every wait of every workgroup) is not
// needed: ordering against the polling load is enough.  -DGPG_HEAVY_ACQUIRE restores the full fence for A/B runs.
#ifdef GPG_HEAVY_ACQUIRE
#define GPG_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define GPG_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup")
#endif

// Producer side: write the L2's dirty lines back (`buffer_wbl2 sc1`) before the flag goes up; __threadfence() would add
// the invalidate of an acquire on top.
#ifdef GPG_HEAVY_RELEASE
#define GPG_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define GPG_ST(p, v) (*(p) = (v))
#else
// Everything another workgroup reads after a flag (factor tiles, reciprocal pivots, solved rows) is stored with
// GPG_ST: an agent-scope atomic store, i.e. `global_store_dwordx2 ... sc1`, written through the XCD's L2 to memory.
// The release is then only the wait for those stores (`s_waitcnt vmcnt(0)`) instead of `buffer_wbl2 sc1`, the
// write-back of every dirty line of the L2.
#define GPG_RELEASE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define GPG_ST(p, v) gpg_store_through((p), (v))
__device__ __forceinline__ void gpg_store_through(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
#endif
// Every thread that wrote part of a tile runs GPG_RELEASE() and the workgroup meets at a barrier before one thread
// raises the flag, so the flag store itself needs no second write-back.
#ifdef GPG_HEAVY_RELEASE
#define GPG_FLAG_UP(p) __hip_atomic_store(p, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
#else
#define GPG_FLAG_UP(p) __hip_atomic_store(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

typedef double d4 __attribute__((ext_vector_type(4)));

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's outstanding global stores (vmcnt(0) --
// microseconds for write-through stores), which the LDS hand-overs inside a finalisation do not need
#define GPG_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

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
#ifdef GPG_POTRF_STAMP   // tools/potrf_probe.hip: cycle counter at the two phase boundaries of every 16-column sub-block
__device__ unsigned long long* g_potrf_stamp;
#define GPG_PS(k) if ((threadIdx.x & 63) == 0 && g_potrf_stamp) g_potrf_stamp[k] = __builtin_readcyclecounter();
#else
#define GPG_PS(k)
#endif
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
    GPG_PS(2 * s)
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
      if (i >= 16 * s + c) GPG_ST(&blk[i + (size_t)(16 * s + c) * ld], a[c]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (i >= 16 * s && i < 16 * s + 16) GPG_ST(&dinv[i], myinv);
    if (piece_flags) {   // dataflow kernels: these 16 columns (and their reciprocal pivots) are final -- publish them
      GPG_RELEASE();
      if (i == 0) GPG_FLAG_UP(piece_flags + s);
    }
    GPG_PS(2 * s + 1)
  }
  return bad;
}

// potrf64_wg: the same factorisation called by ALL FOUR waves of a 256-thread workgroup.  Wave 0 runs the pivot phases exactly as in
// potrf64_wave; the MFMA update that precedes the pivots of sub-block s (64 x 16 panel -= L[:, :16 s] L[16 s : 16 s + 16, :16 s]^T) is
// spread over the waves -- wave R takes the 16-row tile R of the panel, tiles above the diagonal block (R < s) are skipped -- so its
// chain is 4 s dependent MFMAs instead of 16 s issued by one wave (potrf_probe: 2.1 / 3.2 / 4.6 k cycles of 33 k for s = 1 / 2 / 3
// in the one-wave version) -- and all but the last four of them run AHEAD, during the pivot chain of the previous sub-block (waves
// 1 .. 3 are idle then): what stays between two pivot phases is a rank-16 update.  Two workgroup barriers per sub-block; the return
// value is that of wave 0 (other waves return 0).
// tid_in: index of the calling thread inside its 256-thread team (default: threadIdx.x; the pair kernel runs two teams per workgroup,
// whose barriers -- __syncthreads, all 512 threads -- coincide because both teams execute the same sequence).
#ifndef GPG_POTRF_DEFER
#define GPG_POTRF_DEFER 3        // pivot of the next sub-block at which the previous piece is published (0: right behind its stores, as in rounds 1 / 2)
#endif
#if GPG_POTRF_DEFER
#define GPG_POTRF_BARRIER() GPG_LDS_BARRIER()   // the barriers inside order the LDS image only (src is an LDS tile at every call site); no wait for the stores
#else
#define GPG_POTRF_BARRIER() __syncthreads()
#endif
__device__ __forceinline__ int potrf64_wg(const double* src, int sld, double (*St)[64], double* __restrict__ blk, int ld,
                                          double* __restrict__ dinv, int* piece_flags = nullptr, int tid_in = -1) {
  const int tid_team = tid_in < 0 ? (int)threadIdx.x : tid_in;
  const int w = __builtin_amdgcn_readfirstlane(tid_team >> 6);   // wave-uniform by construction; said so, or the wave roles below become divergent branches
  const int i = tid_team & 63, l15 = i & 15, l4 = i >> 4;
  int bad = 0;
  double myinv = 0.0;
  d4 pre;                                                  // tile R = w of the NEXT sub-block's panel, updated with every sub-block before the current one
#pragma unroll
  for (int r = 0; r < 4; ++r) pre[r] = 0.0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    double a[16];
    if (s == 0) {
      if (w == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = src[i + (size_t)c * sld];
      }
    } else {
      if (w >= s) {                                        // what is left of this sub-block's update: the 16 columns just factored
#pragma unroll
        for (int kk = 4 * (s - 1); kk < 4 * s; ++kk) {
          const double fn = St[4 * kk + l4][16 * s + l15];
          const double fm = -St[4 * kk + l4][16 * w + l15];
          pre = __builtin_amdgcn_mfma_f64_16x16x4f64(fn, fm, pre, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) St[16 * s + 4 * r + l4][16 * w + l15] = pre[r];
      }
      GPG_POTRF_BARRIER();                                 // the updated panel is in St[16 s .. 16 s + 15][.]
      if (w == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = St[16 * s + c][i];
      }
    }
    GPG_PS(2 * s)
    if (w == 0) {
      double ajj = readlane_d(a[0], 16 * s);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
#if GPG_POTRF_DEFER
        // the PREVIOUS sub-block's columns went to memory ~1000 cycles ago: their acknowledgements are in (or nearly), the wait costs
        // nothing now -- right behind the stores it cost ~800 cycles of the pivot chain per sub-block (tools/potrf_probe: 28.4 k cycles
        // with the pieces published against 25.1 k without)
        if (s > 0 && c == GPG_POTRF_DEFER && piece_flags) {
          GPG_RELEASE();
          if (i == 0) GPG_FLAG_UP(piece_flags + s - 1);
        }
#endif
        bad = (bad == 0 && !(ajj > 0.0)) ? 16 * s + c + 1 : bad;
        const double y0 = __builtin_amdgcn_rsq(ajj);
        double g = ajj * y0, h = 0.5 * y0;
        double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        r = __builtin_fma(-h, g, 0.5);
        const double dj = __builtin_fma(g, r, g);
        const double inv = 2.0 * __builtin_fma(h, r, h);
        const double lc = a[c] * inv;
        if (c < 15) {
          const double t = __builtin_fma(-lc, lc, a[c + 1]);
          ajj = readlane_d(t, 16 * s + c + 1);
        }
        a[c] = (i == 16 * s + c) ? dj : lc;
        myinv = (i == 16 * s + c) ? inv : myinv;
#pragma unroll
        for (int k2 = c + 1; k2 < 16; ++k2) a[k2] -= a[c] * readlane_d(a[c], 16 * s + k2);
#pragma unroll
        for (int k2 = c + 1; k2 < 16; ++k2) asm volatile("" : "+v"(a[k2]));
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        St[16 * s + c][i] = a[c];
        if (i >= 16 * s + c) GPG_ST(&blk[i + (size_t)(16 * s + c) * ld], a[c]);
      }
      if (i >= 16 * s && i < 16 * s + 16) GPG_ST(&dinv[i], myinv);
      if (piece_flags && (!GPG_POTRF_DEFER || s == 3)) {   // pieces 0 .. 2: published from inside the next sub-block's pivot phase
        GPG_RELEASE();
        if (i == 0) GPG_FLAG_UP(piece_flags + s);
      }
    } else if (s < 3 && w >= s + 1) {
      // look-ahead, while wave 0 runs the pivot chain: the next sub-block's tile, updated with the columns that are already final
      // (sub-blocks 0 .. s-1); only the rank-16 part of sub-block s is left for after the barrier
#pragma unroll
      for (int r = 0; r < 4; ++r) pre[r] = src[(16 * w + l15) + (size_t)(16 * (s + 1) + 4 * r + l4) * sld];
#pragma unroll
      for (int kk = 0; kk < 4 * s; ++kk) {
        const double fn = St[4 * kk + l4][16 * (s + 1) + l15];
        const double fm = -St[4 * kk + l4][16 * w + l15];
        pre = __builtin_amdgcn_mfma_f64_16x16x4f64(fn, fm, pre, 0, 0, 0);
      }
    }
    GPG_PS(2 * s + 1)
    if (s < 3) GPG_POTRF_BARRIER();                        // columns 16 s .. 16 s + 15 of St are final for the next update
  }
  return bad;
}



// DPP quad_perm broadcast of lane Q of every lane quad (see trsm64_kernel / GPG_QUAD_SUBST)
template <int Q>
__device__ __forceinline__ double quad_bcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, Q * 0x55, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, Q * 0x55, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
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

// (Round 3 tried column steps on the UNSCALED row against a column-scaled image -- y_m -= y_j (L_mj / L_jj), reciprocal pivots applied
// once per piece: two dependent hops per column instead of five.  tools/subst_probe: the 64 column steps of a block cost 211 cycles
// each either way -- the steps are bound by their ~21 fp64 instructions at ~10 cycles, not by the chain -- and beside an MFMA wave they
// cost the same while the MFMA wave drops from 64 to 96 cycles per instruction.  +2 % for one chain-bound 64-tile matrix, -2 % for
// batched 128-tile launches (one more FMA per row and step), and the two kernels would no longer agree bit for bit: not adopted.
// Also tried: the per-lane selects moved onto the multipliers (x *= owner ? dv : 1.0; x -= xj * (below ? L : 0.0): exact, 12 instead
// of 16 non-FMA instructions per step) with the next step's L values read through a running LDS pointer (one v_add per step instead of
// one per read): 283 instead of 212 cycles per step in tools/subst_probe, nothing in the kernels.)
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
// GPG_QUAD_SUBST2 with a hook after the four steps of every column group mj (x0[mj], x1[mj] are final then): the dataflow kernels
// put the two stores of the finished columns there, so that the 32 stores of a block trickle out behind the column steps instead of
// being issued -- 4 us -- and acknowledged after them
#define GPG_QUAD_SUBST2_HOOK(x0, x1, Ls, sdinv, q, HOOK)                                      \
  {                                                                                         \
    double lv[2][16];                                                                       \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) lv[0][m] = Ls[0][q][m];                   \
    _Pragma("unroll") for (int mj = 0; mj < 16; ++mj) {                                      \
      GPG_QS2_STEP(x0, x1, Ls, sdinv, q, 0)                                                  \
      GPG_QS2_STEP(x0, x1, Ls, sdinv, q, 1)                                                  \
      GPG_QS2_STEP(x0, x1, Ls, sdinv, q, 2)                                                  \
      GPG_QS2_STEP(x0, x1, Ls, sdinv, q, 3)                                                  \
      HOOK(mj)                                                                              \
    }                                                                                       \
  }
// one 16-column piece of GPG_QUAD_SUBST2 (see GPG_QUAD_SUBST_PIECE)
#define GPG_QUAD_SUBST2_PIECE(x0, x1, Ls, sdinv, q, s)                                       \
  {                                                                                         \
    double lv[2][16];                                                                       \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) lv[0][m] = Ls[16 * (s)][q][m];            \
    _Pragma("unroll") for (int mj = 4 * (s); mj < 4 * (s) + 4; ++mj) {                       \
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

// Reverse quad-row substitution x <- x L^-1 (backward solve against L^T): columns are eliminated from 63 down to 0.
// LsT is the TRANSPOSED image LsT[j][q][m] = L[j][4m + q] (row j of the block), sdinv the reciprocal pivots.
#define GPG_QUAD_SUBST_REV(x, LsT, sdinv, q)                                                 \
  {                                                                                         \
    double lv[2][16];                                                                       \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) lv[1][m] = LsT[63][q][m];                 \
    _Pragma("unroll") for (int mj = 15; mj >= 0; --mj) {                                     \
      GPG_QS_STEP_REV(x, LsT, sdinv, q, 3)                                                   \
      GPG_QS_STEP_REV(x, LsT, sdinv, q, 2)                                                   \
      GPG_QS_STEP_REV(x, LsT, sdinv, q, 1)                                                   \
      GPG_QS_STEP_REV(x, LsT, sdinv, q, 0)                                                   \
    }                                                                                       \
  }
#define GPG_QS_STEP_REV(x, LsT, sdinv, q, QJ)                                                \
  {                                                                                         \
    constexpr int cur = QJ & 1, nxt = cur ^ 1;                                               \
    const int jc = 4 * mj + QJ;                                                             \
    const int jn = jc - 1 >= 0 ? jc - 1 : 0;                                                \
    const int m1n = jn >> 2;                                                                \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) if (m <= m1n) lv[nxt][m] = LsT[jn][q][m]; \
    const double xs = x[mj] * sdinv[jc];                                                    \
    x[mj] = (q == QJ) ? xs : x[mj];                                                         \
    const double xj = quad_bcast<QJ>(x[mj]);                                                \
    if (QJ > 0) {                                                                           \
      const double t = x[mj] - xj * lv[cur][mj];                                            \
      x[mj] = (q < QJ) ? t : x[mj];                                                         \
    }                                                                                       \
    _Pragma("unroll") for (int m = 0; m < mj; ++m) x[m] -= xj * lv[cur][m];                  \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  }

// wave_tile_gemm_nn: acc (wave tile 16 x 64) -= A[64 x K] B[K x 64] with B given "k-major": gb points at
// B[k = 0][n = 0], k runs along contiguous memory, n strides by ldb (a block of L used untransposed by the backward
// solve).  Same staging buffers / pipeline as wave_tile_gemm; only the B staging differs: 8 consecutive threads fetch
// the 16 k of one n (128 contiguous bytes) and scatter them into the k-major LDS rows.
__device__ __forceinline__ void wave_tile_gemm_nn(d4 (&acc)[4], const double* ga, int lda, const double* gb, int ldb, int nchunk,
                                                  double* sA, double* sB, int w, int l15, int l4, int sp, int sk) {
  constexpr int KB = 16, SA = 80, BUF = KB * SA;
  const size_t a8 = (size_t)8 * lda;
  const int tid = threadIdx.x;
  const int bk = 2 * (tid & 7), bn = tid >> 3;               // B staging: k pair, n (and n + 32)
  const double* gbt = gb + bk + (size_t)bn * ldb;
  const size_t bn32 = (size_t)32 * ldb;
  double2 ra0_a, ra0_b, rb0_a, rb0_b, ra1_a, ra1_b, rb1_a, rb1_b;
#define GPG_NN_GLOAD(set)                                        \
  ra##set##_a = *reinterpret_cast<const double2*>(ga);           \
  ra##set##_b = *reinterpret_cast<const double2*>(ga + a8);      \
  rb##set##_a = *reinterpret_cast<const double2*>(gbt);          \
  rb##set##_b = *reinterpret_cast<const double2*>(gbt + bn32);   \
  ga += 2 * a8;                                                  \
  gbt += KB;
#define GPG_NN_SSTORE(buf, set)                                                              \
  {                                                                                          \
    double2 v0, v1;                                                                          \
    v0.x = -ra##set##_a.x; v0.y = -ra##set##_a.y; v1.x = -ra##set##_b.x; v1.y = -ra##set##_b.y;  \
    *reinterpret_cast<double2*>(sA + (buf) * BUF + sk * SA + 2 * sp) = v0;                    \
    *reinterpret_cast<double2*>(sA + (buf) * BUF + (sk + 8) * SA + 2 * sp) = v1;              \
    sB[(buf) * BUF + bk * SA + bn] = rb##set##_a.x;                                           \
    sB[(buf) * BUF + (bk + 1) * SA + bn] = rb##set##_a.y;                                     \
    sB[(buf) * BUF + bk * SA + bn + 32] = rb##set##_b.x;                                      \
    sB[(buf) * BUF + (bk + 1) * SA + bn + 32] = rb##set##_b.y;                                \
  }
#define GPG_NN_COMPUTE(buf)                                                                  \
  _Pragma("unroll") for (int kk = 0; kk < KB; kk += 4) {                                      \
    const double fm = sA[(buf) * BUF + (kk + l4) * SA + 16 * w + l15];                        \
    double fn[4];                                                                            \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) fn[ni] = sB[(buf) * BUF + (kk + l4) * SA + ni * 16 + l15]; \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                          \
      acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn[ni], fm, acc[ni], 0, 0, 0);           \
  }
  GPG_NN_GLOAD(0);
  GPG_NN_GLOAD(1);
  GPG_NN_SSTORE(0, 0);
  __syncthreads();
  for (int ch = 0; ch < nchunk; ch += 2) {
    if (ch + 2 < nchunk) { GPG_NN_GLOAD(0); }
    GPG_NN_COMPUTE(0);
    GPG_NN_SSTORE(1, 1);
    __syncthreads();
    if (ch + 3 < nchunk) { GPG_NN_GLOAD(1); }
    GPG_NN_COMPUTE(1);
    if (ch + 2 < nchunk) { GPG_NN_SSTORE(0, 0); }
    __syncthreads();
  }
#undef GPG_NN_GLOAD
#undef GPG_NN_SSTORE
#undef GPG_NN_COMPUTE
}

#ifdef GPG_STAMP   // diagnostic build of tools/gemm_probe.hip only: per-wave cycle shares of the loop phases
__device__ unsigned long long* g_stamp_buf;
#define GPG_STAMP_MAX 131072   // workgroups with a timeline record
#define GPG_T(var) unsigned long long var; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0);
#define GPG_TR(var) unsigned long long var; __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0);   // 100 MHz ticks
#else
#define GPG_T(var)
#define GPG_TR(var)
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
                                                   double* sdinv, int tid_in = -1) {
  constexpr int KB = 16, SA = 80;                     // SA: +128 B pad keeps ds_read_b64 conflict-free
  constexpr int BUF = KB * SA;                        // doubles per staging buffer
  double* const sA = U;
  double* const sB = U + 2 * BUF;

  const int tid = tid_in < 0 ? (int)threadIdx.x : tid_in, lane = tid & 63;   // thread inside its 256-thread team
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
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
      for (int m = 0; m < 16; ++m) GPG_ST(&Xr[(size_t)(4 * m) * ldx], x[m]);
    }
    __syncthreads();   // X_j visible to the whole workgroup (vmcnt(0) + barrier), Ts / Ls free again
    GPG_T(p5)
#ifdef GPG_STAMP
    ps_pre += p1 - p0; ps_gemm += p2 - p1; ps_tr += p3 - p2; ps_sub += p4 - p3; ps_st += p5 - p4;
#endif
  }
#undef GPG_PS_PREFETCH
#if defined(GPG_STAMP) && defined(GPG_STAMP_PANEL)   // (the panel kernel's own record: would overwrite the dataflow kernels' task records)
  if (lane == 0 && blockIdx.x < 4096 && g_stamp_buf != nullptr) {
    unsigned long long* o = g_stamp_buf + ((size_t)blockIdx.x * 4 + w) * 8;
    o[0] = ps_pre; o[1] = ps_gemm; o[2] = ps_tr; o[3] = ps_sub; o[4] = ps_st;
  }
#endif
}


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
  if (ok) GPG_ACQUIRE();
  return ok;
}

// How far do the completion flags of NR flag rows reach from tile column kdone on (exclusive end kend)?  Called by ONE whole wave: lane l
// looks at tile column base + l of every row -- a handful of independent loads per 64 tile columns instead of NR dependent loads per
// tile column from one thread (round 3: with the flags all up, as they are for 85 % of the tasks of a batched launch, that serial scan
// was the whole "flag wait": ~50 tile columns x 2-4 L2 round trips = 25-65 us per task).  Returns the first column whose flags are not
// all up (kdone if none is up yet), identical in every lane.
template <int NR>
__device__ __forceinline__ int wave_scan_flags(int* const (&rows)[NR], int kdone, int kend) {
  const int lane = threadIdx.x & 63;
  for (int base = kdone; base < kend; base += 64) {
    const int k = base + lane;
    int up = 0;
    if (k < kend) {
      up = 1;
#pragma unroll
      for (int r = 0; r < NR; ++r) up &= __hip_atomic_load(rows[r] + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    }
    const unsigned long long have = __ballot(up);
    const int valid = kend - base < 64 ? kend - base : 64;
    const unsigned long long need = valid == 64 ? ~0ull : ((1ull << valid) - 1ull);
    const unsigned long long miss = ~have & need;
    if (miss) return base + (int)__builtin_ctzll(miss);
  }
  return kend;
}

// X (128 rows x 128 columns, in place) <- X L^-T against a factorised 128 x 128 diagonal tile, whole workgroup.
// Each lane quad carries TWO matrix rows (r and r + 64) through the substitution, so the 128 rows cost two
// substitution sweeps instead of four and half the L-image traffic.  Column block 0 is read from memory
// straight in the quad layout; block 1 first takes its update X1 L21^T on MFMA (two 64-row passes through
// the LDS tile).  U: 4 * 16 * 80 doubles, Ls / sdinv: diagonal-block image.  Ends with a workgroup barrier.

// direct_tile_gemm_acc: the MFMA loop of the dataflow kernels, operand fragments straight from global memory:
// acc (wave tile 64 x 64) += A[64 x 4 nstep] B[64 x 4 nstep]^T, PF k-steps ahead, no LDS; nstep a positive multiple of UN.
// Lane lane&15 = t owns the two adjacent rows 2t, 2t+1 of each 32-row group of its 64-row slices, so one global_load_dwordx4 feeds two
// MFMA operand blocks: 4 load instructions per k-step.  acc[2p + e][2g + m][r] of lane (t, l4) <-> tile row 32 g + 2 t + m, tile column
// 32 p + 2 (4 r + l4) + e.
// Round 3: the loop body is MFMA + loads + scalar arithmetic only.
//  * The SIGN lives outside: callers that need C - A B^T negate their accumulators once before and once after the loop instead of
//    negating the A fragments of every k-step (a v_xor + v_mov per fragment: 8 VALU instructions per 16 MFMAs).  -(fma(a, b, -c)) and
//    fma(-a, b, c) round identically (round-to-nearest is symmetric), so every result keeps its bits.
//  * The operand ADDRESSES are a wave-uniform base (ua / ub, advanced by scalar adds) plus a constant 32-bit lane offset (la / lb,
//    bytes): global_load ... v_off, s[base] instead of two 64-bit VALU pointer increments per k-step.
// VALU instructions of one wave take issue slots from the MFMAs of BOTH waves of its SIMD (tools/subst_probe: a wave of fp64 VALU work
// slows its SIMD partner's MFMA stream from 64 to 96 cycles per instruction).
#ifndef GPG_KSYNC
#define GPG_KSYNC 16       // k-steps between two workgroup barriers inside the MFMA loop (0: none; a multiple of PF + 1)
#endif
// the barriers of direct_tile_gemm_acc<PF, MI, KS> over nstep k-steps, for a wave that has no MFMA work in this run
template <int PF, int KS = GPG_KSYNC>
__device__ __forceinline__ void direct_tile_sync_only(int nstep) {
  if (KS > 0)
    for (int s0 = 0; s0 < nstep; s0 += (KS > 0 ? KS : 1)) __builtin_amdgcn_s_barrier();
}
// acc <- -acc (before and after a run of direct_tile_gemm_acc that is to SUBTRACT the product)
__device__ __forceinline__ void direct_tile_negate(d4 (&acc)[4][4]) {
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ni][mi][r] = -acc[ni][mi][r];
}
// MI = 4: all 64 rows of the wave tile; MI = 2: only its first 32-row group (acc[.][0], acc[.][1]) -- the right-hand-side tile row of
// the factorisation carries two real rows.
#ifndef GPG_UNROLL_KSTEPS
#define GPG_UNROLL_KSTEPS 32
#endif
typedef double gpg_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 gpg_dx_nt(const char* p) {
  const gpg_d2v v = __builtin_nontemporal_load(reinterpret_cast<const gpg_d2v*>(p));
  double2 r; r.x = v.x; r.y = v.y; return r;
}
__device__ __forceinline__ double2 gpg_dx_x2(const char* p, unsigned l0, unsigned l8) {   // two 8-byte loads (l8 = l0 + 8, opaque to the compiler)
  double2 r;
  r.x = *reinterpret_cast<const double*>(p + (size_t)l0);
  r.y = *reinterpret_cast<const double*>(p + (size_t)l8);
  return r;
}
template <int PF, int MI = 4, int KS = GPG_KSYNC, int UN = GPG_UNROLL_KSTEPS>
__device__ __forceinline__ void direct_tile_gemm_acc(d4 (&acc)[4][4], const double* ua_d, unsigned la, int lda, const double* ub_d,
                                                     unsigned lb, int ldb, int nstep) {
  const size_t sa = (size_t)4 * lda * sizeof(double), sb = (size_t)4 * ldb * sizeof(double);   // bytes per k-step
  const char* ua = reinterpret_cast<const char*>(ua_d);
  const char* ub = reinterpret_cast<const char*>(ub_d);
  double2 fa0[PF + 1], fa1[PF + 1], fb0[PF + 1], fb1[PF + 1];
#if defined(GPG_DX_X2)
  unsigned la8 = la + 8u, lb8 = lb + 8u;
  asm volatile("" : "+v"(la8), "+v"(lb8));
#endif
#if defined(GPG_DX_NT)        // diagnostic builds of tools/subst_probe.hip only: other flavours of the operand loads
#define GPG_DX_LD(u, l, x) gpg_dx_nt((u) + (size_t)(l) + (x))
#elif defined(GPG_DX_X2)
#define GPG_DX_LD(u, l, x) gpg_dx_x2((u) + (x), (l), (l##8))
#else
#define GPG_DX_LD(u, l, x) (reinterpret_cast<const double2*>((u) + (size_t)(l))[(x) / 16])
#endif
#define GPG_DX_LOAD(set)                                                              \
  {                                                                                   \
    fa0[set] = GPG_DX_LD(ua, la, 0);                                                  \
    if (MI == 4) fa1[set] = GPG_DX_LD(ua, la, 256);                                   \
    fb0[set] = GPG_DX_LD(ub, lb, 0);                                                  \
    fb1[set] = GPG_DX_LD(ub, lb, 256);                                                \
  }
#define GPG_DX_MFMA(set)                                                              \
  {                                                                                   \
    const double fm[4] = {fa0[set].x, fa0[set].y, MI == 4 ? fa1[set].x : 0.0, MI == 4 ? fa1[set].y : 0.0}; \
    const double fn[4] = {fb0[set].x, fb0[set].y, fb1[set].x, fb1[set].y};            \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                   \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                \
        acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn[ni], fm[mi], acc[ni][mi], 0, 0, 0); \
  }
  static_assert(((PF + 1) & PF) == 0 && UN % (PF + 1) == 0 && (KS == 0 || UN % KS == 0), "PF + 1 a power of two dividing UN; KS divides UN");
  // Software pipeline, UN k-steps (one 128-column tile column by default) of straight-line code per trip of the loop.  EVERY k-step
  // issues its loads, unconditionally: the operand pointers simply stop advancing at the last k-step of the run, so the PF stages
  // past the end re-read that k-step (vector-L1 hits, values never used).  With that the number of loads in flight is the same on
  // every path into every k-step and the compiler's s_waitcnt insertion is exact (vmcnt(4 PF): only the newest PF stages stay in
  // flight).  Rounds 1 and 2 ran `for (4 k-steps) { if (more) load; mfma }`: that loop's header merges predecessors with different
  // numbers of loads in flight, the pessimistic merge is vmcnt(0), and every wave drained its whole prefetch queue once per four
  // k-steps (found in the ISA in round 3).  The advance is scalar arithmetic (s_cselect + s_add).
#define GPG_DX_ADVANCE(li)                                                            \
  {                                                                                   \
    const size_t more = (li) + 1 < nstep ? ~(size_t)0 : (size_t)0;                    \
    ua += sa & more;                                                                  \
    ub += sb & more;                                                                  \
  }
#pragma unroll
  for (int s = 0; s < PF; ++s) { GPG_DX_LOAD(s) GPG_DX_ADVANCE(s) }
  for (int s0 = 0; s0 < nstep; s0 += UN) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      GPG_DX_LOAD((u + PF) % (PF + 1))
      GPG_DX_ADVANCE(s0 + u + PF)
      __builtin_amdgcn_sched_barrier(0);
      GPG_DX_MFMA(u % (PF + 1))
      __builtin_amdgcn_sched_barrier(0);
      // Keep the waves of the workgroup within KS k-steps of each other: the waves that read the same operand slice only meet in
      // the 32-KB vector L1 if they ask for it within a k-step or two (PMC, round 3, profiles/r03k_*: without this the L1 absorbs
      // about half of the second wave's reads; with it L2 requests fall by 17 %, memory traffic by 10 %, the launch gains 1.2 %).
      // Plain s_barrier: no counter is drained, the loads in flight stay in flight.  Every wave of the workgroup must come through
      // here the same number of times (waves without MFMA work in a run call direct_tile_sync_only).
      if (KS > 0 && u % (KS > 0 ? KS : 1) == (KS > 0 ? KS : 1) - 1) __builtin_amdgcn_s_barrier();
    }
  }
#undef GPG_DX_ADVANCE
#undef GPG_DX_LOAD
#undef GPG_DX_LD
#undef GPG_DX_MFMA
}


// the former interface (per-lane pointers, acc -= A B^T) for the diagnostic programs under tools/
template <int PF, int MI = 4, int KS = GPG_KSYNC>
__device__ __forceinline__ void direct_tile_gemm_x2(d4 (&acc)[4][4], const double* pa, int lda, const double* pb, int ldb, int nstep) {
  direct_tile_negate(acc);
  direct_tile_gemm_acc<PF, MI, KS, 16>(acc, pa, 0u, lda, pb, 0u, ldb, nstep);   // (the probes run 16 and 512 k-steps)
  direct_tile_negate(acc);
}

}  // namespace
