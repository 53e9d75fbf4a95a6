// Dataflow (left-looking, tile task graph) Cholesky for gfx950: the whole factorisation as ONE launch.
// Replaces scipy.linalg.cho_factor(Kcov_precon, lower=True) (reference Kernel.py:251; LAPACK dpotrf).
// tile_chol_kernel: 64 x 64 tiles (small matrices), tile128_chol_kernel: 128 x 128 tiles.
#include "chol_device.h"

// tools/tile_probe A/B only (WRONG results): every task reads its B operand from tile row 0 -- what would the launch gain if the row
// panel L_j,0:j cost no memory traffic at all (upper bound of any operand-sharing scheme)?  -DGPG_FAKE_B
#ifdef GPG_FAKE_B
#define GPG_FAKE_B_ROW(cj) ((size_t)0 * (cj))
#else
#define GPG_FAKE_B_ROW(cj) (cj)
#endif
#ifndef GPG_MFMA_PF
#define GPG_MFMA_PF 1      // k-steps of operand fragments in flight ahead of the MFMAs (16 VGPRs each); PF + 1 a power of two.
                           // Round 3, with exact waits in the loop (direct_tile_gemm_acc): 1 is as fast as 3 on ten cfg3 matrices per launch
                           // (299.9 / 299.2 ms) and faster on short contractions (4608 columns x 16: 10.85 / 11.02 ms; 9216 x 1: 7.7 / 8.1) --
                           // the second wave of the SIMD covers the latency, and a shallower queue keeps the shared slices in the 32-KB L1
#endif

namespace {

// One ticket counter, or (A/B builds, -DGPG_TICKET_QUEUES) eight of them: workgroup b draws from queue b % 8, whose tasks
// are the list positions q, q + 8, ... -- the static task -> XCD deal of a one-task-per-workgroup launch, kept persistent.
#ifdef GPG_TICKET_QUEUES
#define GPG_TICKET_FETCH(ticket) ((int)(blockIdx.x & 7) + 8 * __hip_atomic_fetch_add((ticket) + (blockIdx.x & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
#else
#define GPG_TICKET_FETCH(ticket) __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

// Next task of a persistent workgroup: one relaxed agent-scope fetch-add by thread 0, shared through LDS.  The two
// barriers also separate the LDS use of consecutive tasks.
__device__ __forceinline__ int next_ticket(int* ticket, int* sh, int round) {
#ifdef GPG_NO_PERSIST   // A/B builds of tools/tile_probe.hip only: one task per workgroup, taken in dispatch order (the round-1 schedule)
  (void)ticket; (void)sh;
  return round == 0 ? (int)blockIdx.x : 0x7fffffff;
#else
#ifdef GPG_TICKET_ONESHOT   // A/B: ticket order, but one task per workgroup (grid = number of tasks; the hardware refills the slots)
  if (round > 0) return 0x7fffffff;
#endif
  (void)round;
  __syncthreads();
  if (threadIdx.x == 0) *sh = GPG_TICKET_FETCH(ticket);
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(*sh);   // the same word for every lane: said so, or every address derived from the task is per-lane arithmetic
#endif
}

// Wave priority by phase.  The two workgroups of a compute unit share every SIMD, and between equal priorities the
// OLDER wave wins the issue arbitration.  With one-task workgroups that order follows the task order by itself (the
// workgroup nearer its end is the older one); persistent workgroups keep their launch age for the whole factorisation,
// so half of the latency-bound finalisations (dependent fp64 VALU chains: pivots, substitutions) would queue behind the
// partner's MFMA stream whatever their place on the critical path.  The finalisation therefore raises its priority
// explicitly, the accumulation of a diagonal tile (the head of its tile column) runs above ordinary accumulation.
#ifndef GPG_NO_SETPRIO
#define GPG_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define GPG_PRIO(n)
#endif
// Priority by distance from the diagonal, dd = i - j of the tile: the head of a tile column (diagonal tile, the next few
// rows) is what the following columns' heads wait for.  Timelines (tools/timeline_compare.py) show tile (j+1, j) finishing
// 12 us after diag(j) with one-task workgroups (where the older = earlier task wins the arbitration) but 124 us after it
// with persistent ones at equal priorities.
#ifndef GPG_PRIO_NEAR
#define GPG_PRIO_NEAR 3
#endif
#define GPG_PRIO_ACC(dd) { if ((dd) == 0) { GPG_PRIO(2); } else if ((dd) <= GPG_PRIO_NEAR) { GPG_PRIO(1); } }
#define GPG_PRIO_FIN(dd) { if ((dd) <= GPG_PRIO_NEAR) { GPG_PRIO(3); } else { GPG_PRIO(2); } }

// Ticket of the NEXT task, fetched by the publish step of the current one (GPG_PUBLISH_AND_NEXT): the fetch-add travels
// with the drain of the tile's write-through stores, so a workgroup that finishes a task knows its next one without a
// further round trip.  That matters on the critical path of a factorisation: the diagonal tile of the next column
// usually gets its workgroup only when some task ends (all slots are busy), and whatever that workgroup spends finding
// out what to do next is added to every one of the Mt hops of the dependency chain (measured: 141 tile columns x ~6 us).
// The ticket is held without being worked on only for the duration of that drain.
__shared__ int g_next_ticket;
#if defined(GPG_NO_PERSIST) || defined(GPG_TICKET_ONESHOT)
#define GPG_PUBLISH_AND_NEXT(ticket, flag_ptr)                                             \
  GPG_RELEASE();                                                                            \
  __syncthreads();                                                                          \
  if (threadIdx.x == 0) { GPG_FLAG_UP(flag_ptr); g_next_ticket = 0x7fffffff; }             \
  GPG_PRIO(0);
#else
#define GPG_PUBLISH_AND_NEXT(ticket, flag_ptr)                                             \
  {                                                                                         \
    int nxt_ = 0;                                                                           \
    if (threadIdx.x == 0) nxt_ = GPG_TICKET_FETCH(ticket);                                  \
    GPG_RELEASE();                                                                          \
    if (threadIdx.x == 0) g_next_ticket = nxt_;                                             \
    __syncthreads();                                                                        \
    if (threadIdx.x == 0) GPG_FLAG_UP(flag_ptr);                                            \
    GPG_PRIO(0);                                                                            \
  }
#endif

// The task body of a persistent kernel is compiled as a function of its own (noinline) that fetches the kernel's
// arguments from the kernarg segment by scalar loads: inlined into the task loop, the arguments stay live in ~25 SGPRs
// across the back edge and the compiler spills the broadcast operands of the pivot chain (potrf64_wave) through
// v_writelane / v_readlane instead -- 400 extra pairs in tile_chol_kernel, 8 % on a chain-bound factorisation.  The
// kernel's formal parameter is only there to give the segment its layout.
#define GPG_KERNARGS(T, ap)                                                                                   \
  const __attribute__((address_space(4))) T* ap =                                                             \
      (const __attribute__((address_space(4))) T*)__builtin_amdgcn_kernarg_segment_ptr();                      \
  asm volatile("" : "+s"(ap))
// In a (noinline) device function the kernarg segment pointer is NOT available through the builtin (it reads as null
// there): the kernel passes it as an ordinary argument, which arrives in VGPRs; readfirstlane makes it scalar again so
// that the loads through it are s_load.
#define GPG_KERNARGS_FROM(T, ap, bits)                                                                        \
  const unsigned long long ap##_u =                                                                           \
      ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((bits) >> 32)) << 32) |             \
      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((bits) & 0xffffffffull));            \
  const __attribute__((address_space(4))) T* ap = (const __attribute__((address_space(4))) T*)ap##_u

// ------------------------------------------------------------------------------------------------
// tile_chol_kernel: dataflow (left-looking) Cholesky of the trailing block A[c0:, c0:] in 64 x 64 tiles,
// ONE launch.  Used where the blocked algorithm is latency-bound: the last few thousand columns of a large
// matrix and small matrices as a whole.  Workgroup b owns tile (i, j) = tasks[b] (column-major task order,
// rows i >= j; the right-hand-side rows below the matrix are ordinary tile rows):
//     acc  = A_ij - sum_{k<j} L_ik L_jk^T      MFMA, k-blocks consumed as soon as their flags are up
//     i==j : L_jj = chol(acc) by wave 0 (potrf64_wave), reciprocal pivots to dinv
//     i> j : L_ij = acc L_jj^-T by the quad-row substitution, once flag(j, j) is up
//     publish: __threadfence, then flag(i, j) = 1 (agent-scope release)
// Scheduling: PERSISTENT workgroups.  The launch has at most as many workgroups as the device holds at once; each
// takes its next task from an atomic ticket counter, in list order, until the list is exhausted.  A task only ever
// waits for tasks with a smaller ticket.  Progress: let T be the smallest unfinished ticket.  If T has been taken,
// its workgroup is resident (it took the ticket while running) and everything T waits for has a smaller ticket, i.e.
// is finished -- T completes.  If T has not been taken, every resident workgroup holds a smaller ticket, all of those
// are finished, so one of them asks for its next ticket and gets T.  Nothing here depends on the order in which the
// hardware dispatches workgroups, on how many of them are resident, or on having the device alone: another process's
// launch on the same GPU can only delay this one, never block it.  Every wait still is bounded in time as a backstop:
// on timeout the kernel raises the abort word, all workgroups drain, and the host falls back on the blocked schedule.
// The serial chain per 64 columns is potrf -> substitution -> one 64-deep MFMA block (~20 us) instead of three
// dependent launches per step plus B_p and U_p.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool
tile_chol_task(int tix, double* A, int ld, int c0, int Mt, const int* __restrict__ tasks, int* flags, int* pieces, int* abort_word,
               int* ticket, double* __restrict__ dinv, int* __restrict__ info, int N, const int* __restrict__ batch_of, size_t a_stride,
               int d_stride, int f_stride, int fuse) {
  constexpr int KB = 16, SA = 80, BUF = KB * SA;
  __shared__ __attribute__((aligned(16))) double U[4 * BUF];      // staging sA[2] | sB[2]; later the tile Ts[64][SA]
  __shared__ __attribute__((aligned(16))) double Ls[64][4][18];   // L_jj image (i > j) / potrf scratch St[64][64] (i == j)
  __shared__ double sdinv[64];
  __shared__ int sh_kr;
  double* const sA = U;
  double* const sB = U + 2 * BUF;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int task = tasks[tix];
  const int ti = task & 0xffff, tj = task >> 16;
  if (batch_of) {   // batched launch: several independent matrices (restart rows) share the grid
    const int b = batch_of[tix];
    A += (size_t)b * a_stride;
    dinv += (size_t)b * d_stride;
    info += b;
    flags += (size_t)b * f_stride;
    pieces += (size_t)b * f_stride;
  }
  const size_t r0 = (size_t)c0 + 64 * (size_t)ti;        // first matrix row of the tile
  const size_t cj = (size_t)c0 + 64 * (size_t)tj;        // first matrix column of the tile
  const int q = tid & 3;
  const int sp = tid & 31, sk = tid >> 5;
  // Fused diagonal task (fuse != 0, tj > 0): this workgroup ALSO owns the sub-diagonal tile (tj, tj-1) -- the task list has no task of
  // its own for it.  It carries that tile's accumulators (acc1) next to its own, substitutes them against the pieces of the previous
  // diagonal tile as they are published, and applies the result to its own accumulators from LDS: the chain diagonal tile -> tile
  // (j+1, j) -> store / flag / poll / reload -> diagonal tile (j+1, j+1) loses its two trips through memory (~20.7 -> ~17 us per column).
  const bool fused = fuse != 0 && ti == tj && tj > 0;
  const int kend = fused ? tj - 1 : tj;                  // tile columns the MFMA loop covers (the fused task's last one comes from LDS)
  const size_t cjm = cj - 64;                            // first matrix column of tile column tj - 1 (fused only)
  int* const frow_i = flags + (size_t)ti * Mt;           // flags of tile row i
  int* const frow_j = flags + (size_t)(fused ? tj - 1 : tj) * Mt;

  // accumulators start as A_ij
  d4 acc[4];
  {
    const double* Cw = A + r0 + 16 * w + l15 + (cj + l4) * (size_t)ld;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ni][r] = Cw[(size_t)(ni * 16 + 4 * r) * ld];
  }
  d4 acc1[4];                                            // fused: A(tj, tj-1), nobody else writes that tile
  if (fused) {
    const double* Cw = A + r0 + 16 * w + l15 + (cjm + l4) * (size_t)ld;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc1[ni][r] = Cw[(size_t)(ni * 16 + 4 * r) * ld];
  }

  // ---- (1) left-looking accumulation over the finished tile columns ------------------------------------------
#ifdef GPG_STAMP
  const unsigned long long tk_start = __builtin_amdgcn_s_memrealtime();
  unsigned long long tk_spin = 0, tk_gemm = 0, tk_runs = 0;
#endif
  GPG_PRIO_ACC(ti - tj)
  __shared__ int sh_kr1;
  int kdone = 0, kd1 = 0;                                // tile columns applied to acc / to the fused task's acc1
  while (kdone < kend || (fused && kd1 < kend)) {
    GPG_TR(q0)
    if (tid == 0) {
      int kr = kdone, kr1 = kd1;
      const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        if (fused) {   // own accumulators: tile row tj only; acc1: tile rows tj and tj - 1 -- the two frontiers advance independently
          while (kr < kend && __hip_atomic_load(frow_i + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ++kr;
          while (kr1 < kr && __hip_atomic_load(frow_j + kr1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ++kr1;
          if (kr > kdone || kr1 > kd1) break;
        } else {
          while (kr < kend && __hip_atomic_load(frow_i + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 &&
                 __hip_atomic_load(frow_j + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
            ++kr;
          if (kr > kdone) break;
        }
        if (__builtin_amdgcn_s_memrealtime() - t_wait > GPG_TILE_WAIT_TICKS || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicMax(info, GPG_INFO_INTERNAL);
          kr = -1;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
      sh_kr = kr;
      sh_kr1 = kr1;
    }
    __syncthreads();
    const int kr = sh_kr, kr1 = sh_kr1;
    if (kr < 0) return false;                            // abort: drain
    GPG_ACQUIRE();   // the producers' tiles are visible from here on
    GPG_TR(q1)
    if (kr > kdone) {
      const size_t ck = (size_t)c0 + 64 * (size_t)kdone;
      wave_tile_gemm(acc, A + r0 + 2 * sp + (ck + sk) * (size_t)ld, ld, A + cj + 2 * sp + (ck + sk) * (size_t)ld, ld,
                     4 * (kr - kdone), sA, sB, w, l15, l4, sp, sk);
    }
    __syncthreads();                                     // staging buffers free again; sh_kr may be rewritten
    if (fused && kr1 > kd1) {                            // the sub-diagonal tile's share: rows tj against rows tj - 1
      const size_t ck = (size_t)c0 + 64 * (size_t)kd1;
      wave_tile_gemm(acc1, A + r0 + 2 * sp + (ck + sk) * (size_t)ld, ld, A + cjm + 2 * sp + (ck + sk) * (size_t)ld, ld,
                     4 * (kr1 - kd1), sA, sB, w, l15, l4, sp, sk);
      __syncthreads();
    }
    GPG_TR(q2)
#ifdef GPG_STAMP
    tk_spin += q1 - q0; tk_gemm += q2 - q1; ++tk_runs;
#endif
    kdone = kr;
    kd1 = kr1;
  }
#ifdef GPG_STAMP
  const unsigned long long tk_fin0 = __builtin_amdgcn_s_memrealtime();
#endif

  GPG_PRIO_FIN(ti - tj)
  // one 16-column piece of the diagonal tile of tile column PC (matrix column PCJ): wait, image, quad-row substitution of x
#define GPG_TC_PIECE(S, PC, PCJ)                                                             \
    {                                                                                       \
      if (!wg_wait_flag(pieces + 4 * (PC) + (S), abort_word, info, &sh_kr)) return false;    \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                        \
        const int t = tid + 256 * u, jj = 16 * (S) + (t >> 6), k = t & 63;                   \
        Ls[jj][k & 3][k >> 2] = (A + (PCJ) + (PCJ) * (size_t)ld)[k + (size_t)jj * ld];       \
      }                                                                                     \
      if (tid < 16) sdinv[16 * (S) + tid] = dinv[(PCJ) + 16 * (S) + tid];                    \
      __syncthreads();                                                                      \
      GPG_QUAD_SUBST_PIECE(x, Ls, sdinv, q, S)                                               \
      /* pin x: otherwise the FMAs of a piece are deferred into the next ones and everything spills */ \
      _Pragma("unroll") for (int m = 0; m < 16; ++m) asm volatile("" : "+v"(x[m]));              \
    }
  if (fused) {
    // ---- (1') the sub-diagonal tile: acc1 -> LDS -> quad rows, substitution against L(tj-1, tj-1) piece by piece, then out to memory
    //      (asynchronously) and, through the LDS tile, into this task's own accumulators: acc -= X X^T ------------------------------
    {
      double* Ts = U;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ts[(ni * 16 + 4 * r + l4) * SA + 16 * w + l15] = acc1[ni][r];
    }
    __syncthreads();
    double x[16];
    {
      const double* Tr = U + q * SA + (tid >> 2);
#pragma unroll
      for (int m = 0; m < 16; ++m) x[m] = Tr[(4 * m) * SA];
    }
    GPG_TC_PIECE(0, tj - 1, cjm)
    GPG_TR(tq_p0)
    GPG_TC_PIECE(1, tj - 1, cjm)
    GPG_TC_PIECE(2, tj - 1, cjm)
    GPG_TR(tq_p2)
#ifdef GPG_STAMP
    if (!wg_wait_flag(pieces + 4 * (tj - 1) + 3, abort_word, info, &sh_kr)) return false;   // (diagnostic: when piece 3 is SEEN; the wait inside the piece macro then falls through)
#endif
    GPG_TR(tq_seen3)
    GPG_TC_PIECE(3, tj - 1, cjm)
    GPG_TR(tq_sub3)
    {
      double* Xr = A + r0 + (tid >> 2) + (cjm + q) * (size_t)ld;
      double* Tw = U + q * SA + (tid >> 2);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        GPG_ST(&Xr[(size_t)(4 * m) * ld], x[m]);
        Tw[(4 * m) * SA] = x[m];                          // every thread rewrites exactly the entries it read: no barrier needed before
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 64; kk += 4) {
      const double fm = -U[(kk + l4) * SA + 16 * w + l15];
      double fn[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) fn[ni] = U[(kk + l4) * SA + ni * 16 + l15];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn[ni], fm, acc[ni], 0, 0, 0);
    }
    GPG_RELEASE();                                        // the stores of X have had the whole update to complete
    __syncthreads();                                      // ... and the LDS tile is free again
    if (tid == 0) GPG_FLAG_UP(frow_i + (tj - 1));          // tile (tj, tj-1) is final for everybody else
    GPG_TR(tq_upd)
#ifdef GPG_STAMP   // tools/timeline_chain64.py: the chain of the fused diagonal tasks, split at these stamps
    if (tid == 0 && g_stamp_buf != nullptr && tix < GPG_STAMP_MAX) {
      unsigned long long* pz = g_stamp_buf + (size_t)GPG_STAMP_MAX * 16 + (size_t)tix * 16;
      pz[0] = tq_p0; pz[1] = tq_p2; pz[2] = tq_seen3; pz[3] = tq_sub3; pz[4] = tq_upd;
    }
#endif
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
    {   // diagonal tile: factor it (pivots on wave 0, the MFMA updates on all four; entries above the diagonal are garbage nobody reads)
      double* blk = A + r0 + cj * (size_t)ld;
#ifdef GPG_STAMP
      if (tid == 0 && g_stamp_buf != nullptr && tix < GPG_STAMP_MAX) (g_stamp_buf + (size_t)GPG_STAMP_MAX * 16 + (size_t)tix * 16)[5] = __builtin_amdgcn_s_memrealtime();
#endif
      const int bad = potrf64_wg(U, SA, reinterpret_cast<double(*)[64]>(&Ls[0][0][0]), blk, ld, dinv + cj, pieces + 4 * tj);
#ifdef GPG_STAMP
      if (tid == 0 && g_stamp_buf != nullptr && tix < GPG_STAMP_MAX) (g_stamp_buf + (size_t)GPG_STAMP_MAX * 16 + (size_t)tix * 16)[6] = __builtin_amdgcn_s_memrealtime();
#endif
      if (w == 0 && bad && lane == 0 && (int)cj + bad - 1 < N) atomicCAS(info, 0, (int)cj + bad);
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
    GPG_TC_PIECE(0, tj, cj)
    GPG_TC_PIECE(1, tj, cj)
    GPG_TC_PIECE(2, tj, cj)
    GPG_TC_PIECE(3, tj, cj)
#undef GPG_TC_PIECE
    double* Xr = A + r0 + (tid >> 2) + (cj + q) * (size_t)ld;
#pragma unroll
    for (int m = 0; m < 16; ++m) GPG_ST(&Xr[(size_t)(4 * m) * ld], x[m]);
  }
  // ---- (3) publish (and fetch the next ticket) --------------------------------------------------------------------
#ifdef GPG_STAMP
  const unsigned long long tk_pub0 = __builtin_amdgcn_s_memrealtime();
#endif
  GPG_PUBLISH_AND_NEXT(ticket, frow_i + tj)
#ifdef GPG_STAMP
  if (tid == 0 && g_stamp_buf != nullptr && tix < GPG_STAMP_MAX) {
    unsigned long long* o = g_stamp_buf + (size_t)tix * 8;
    o[0] = tk_start; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = tk_spin; o[3] = tk_gemm; o[4] = tk_runs;
    o[5] = tk_fin0; o[6] = (unsigned long long)task; o[7] = (unsigned long long)blockIdx.x;
    unsigned long long* fo = g_stamp_buf + (size_t)GPG_STAMP_MAX * 8 + (size_t)tix * 8;
    fo[0] = tk_fin0; fo[1] = tk_pub0; fo[2] = fo[3] = fo[4] = fo[5] = fo[6] = fo[7] = 0;
  }
#endif
  return true;
}

struct TileCholArgs {
  double* A; int ld, c0, Mt; const int* tasks; int ntask; int* flags; int* pieces; int* abort_word; int* ticket; double* dinv;
  int* info; int N; const int* batch_of; size_t a_stride; int d_stride, f_stride, fuse;
};

__device__ __noinline__ int tile_chol_task_call(int tix_v, unsigned long long kernarg_bits) {
  GPG_KERNARGS_FROM(TileCholArgs, ap, kernarg_bits);
  const int tix = __builtin_amdgcn_readfirstlane(tix_v);
  return tile_chol_task(tix, ap->A, ap->ld, ap->c0, ap->Mt, ap->tasks, ap->flags, ap->pieces, ap->abort_word, ap->ticket, ap->dinv,
                        ap->info, ap->N, ap->batch_of, ap->a_stride, ap->d_stride, ap->f_stride, ap->fuse) ? 1 : 0;
}

__global__ void __launch_bounds__(256, 2) tile_chol_kernel(TileCholArgs) {
  int tix;
  {
    GPG_KERNARGS(TileCholArgs, ap);
    tix = next_ticket(ap->ticket, &g_next_ticket, 0);       // the first ticket; every later one comes with a task's publish step
  }
  for (;;) {
    GPG_KERNARGS(TileCholArgs, ap);
    if (tix >= ap->ntask) return;
    if (!__builtin_amdgcn_readfirstlane(tile_chol_task_call(tix, (unsigned long long)ap))) return;   // aborted: every workgroup drains
    tix = __builtin_amdgcn_readfirstlane(g_next_ticket);                                    // written before the barrier of the publish step
  }
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
// Persistent workgroups, ticket order, progress argument and bounded waits as in tile_chol_kernel.
// ------------------------------------------------------------------------------------------------
// The diagonal tile publishes its pieces as they are final -- L11 in four 16-column pieces (pa[0..3], while its first
// potrf64 is still running), L21 (flag_c, after its 64-row solve), L22 in four pieces (pb[0..3]): every column block of
// this tile is substituted piece by piece, and the MFMA update of block 1 runs while the diagonal tile is still in its
// second potrf64.
#ifdef GPG_STAMP
__shared__ unsigned long long* t128_fo;      // finalisation record of the running task (thread 0 writes and reads it)
__shared__ unsigned long long t128_wait_acc; // ticks thread 0 spent polling piece flags in this task's finalisation (wg_wait_flag2)
#define GPG_FS2(k) if (threadIdx.x == 0 && t128_fo) t128_fo[k] = __builtin_amdgcn_s_memrealtime();
// piece-level record (third region of the stamp buffer, 16 words per task): per piece S of column block 0 -- after the flag wait,
// after the image is in LDS (barrier), after the 16 column steps
#define GPG_FS3(k) if (threadIdx.x == 0 && t128_fo) (g_stamp_buf + (size_t)GPG_STAMP_MAX * 16 + 2 * (t128_fo - (g_stamp_buf + (size_t)GPG_STAMP_MAX * 8)))[k] = __builtin_amdgcn_s_memrealtime();
#else
#define GPG_FS2(k)
#define GPG_FS3(k)
#endif
__device__ __forceinline__ int tile_solve_rows128(const double* L, int ldl, const double* dinv, double* X, int ldx, double* U,
                                                  double (*Ls)[4][18], double* sdinv, int* pa, int* flag_c, int* pb,
                                                  int* abort_word, int* info, int* sh) {
  constexpr int SA = 80, BUF = 16 * SA;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int q = tid & 3, rr = tid >> 2;
  const int sp = tid & 31, sk = tid >> 5;
  double x0[16], x1[16];
  // ---- column block 0 ----------------------------------------------------------------------------------------
  {
    const double* Xr = X + rr + (size_t)q * ldx;     // own rows: in flight while the flag is polled
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      x0[m] = Xr[(size_t)(4 * m) * ldx];
      x1[m] = Xr[64 + (size_t)(4 * m) * ldx];
    }
  }
  // L11 arrives in four 16-column pieces (pa[0..3]) while the diagonal tile is still in its first potrf64
#define GPG_T128_PIECE(FL, LP, DOFF, S)                                                      \
  {                                                                                         \
    if (!wg_wait_flag((FL) + (S), abort_word, info, sh)) return 0;                           \
    if ((DOFF) == 0) { GPG_FS3(3 * (S)) }                                                    \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                          \
      const int t = tid + 256 * u, jj = 16 * (S) + (t >> 6), k = t & 63;                     \
      Ls[jj][k & 3][k >> 2] = (LP)[k + (size_t)jj * ldl];                                    \
    }                                                                                       \
    if (tid < 16) sdinv[16 * (S) + tid] = dinv[(DOFF) + 16 * (S) + tid];                     \
    __syncthreads();                                                                        \
    if ((DOFF) == 0) { GPG_FS3(3 * (S) + 1) }                                                \
    GPG_QUAD_SUBST2_PIECE(x0, x1, Ls, sdinv, q, S)                                           \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) { asm volatile("" : "+v"(x0[m])); asm volatile("" : "+v"(x1[m])); } \
    if ((DOFF) == 0) { GPG_FS3(3 * (S) + 2) }                                                \
  }
  GPG_T128_PIECE(pa, L, 0, 0)
  GPG_T128_PIECE(pa, L, 0, 1)
  GPG_T128_PIECE(pa, L, 0, 2)
  GPG_T128_PIECE(pa, L, 0, 3)
  GPG_FS2(5)
  {
    double* Xr = X + rr + (size_t)q * ldx;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      GPG_ST(&Xr[(size_t)(4 * m) * ldx], x0[m]);
      GPG_ST(&Xr[64 + (size_t)(4 * m) * ldx], x1[m]);
    }
  }
  if (!wg_wait_flag(flag_c, abort_word, info, sh)) return 0;   // barrier inside: X1 visible to the workgroup, Ls free
  GPG_FS2(6)
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
  GPG_FS2(7)
  {   // L22 in four pieces (pb[0..3]) while the diagonal tile is in its second potrf64
    const double* L22 = L + 64 + (size_t)64 * ldl;
    GPG_T128_PIECE(pb, L22, 64, 0)
    GPG_T128_PIECE(pb, L22, 64, 1)
    GPG_T128_PIECE(pb, L22, 64, 2)
    GPG_T128_PIECE(pb, L22, 64, 3)
  }
#undef GPG_T128_PIECE
  {
    double* Xr = X + rr + (size_t)(64 + q) * ldx;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      GPG_ST(&Xr[(size_t)(4 * m) * ldx], x0[m]);
      GPG_ST(&Xr[64 + (size_t)(4 * m) * ldx], x1[m]);
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
__shared__ int t128_sh_ok;
__shared__ int t128_sh2[2];           // verdict on the NEXT 16-column piece (fin128_offdiag; two words used alternately: a barrier lies between two uses of one)

// LDS of pair128_chol_kernel (512 threads = two teams; defined here because fin128_offdiag serves both 128-tile kernels)
__shared__ __attribute__((aligned(16))) double p128_U[2][4 * 16 * 80];    // per team: staging / transposition tile of the finalisation
__shared__ __attribute__((aligned(16))) double p128_Ls[2][64][4][18];     // per team: diagonal-block image / potrf scratch
__shared__ double p128_sdinv[2][64];
__shared__ int p128_sh2[2];
__shared__ int p128_sh_ok;            // result of the joint flag waits: ONE word for both teams (a function-local __shared__ would be one per template instance)

// Both flags up?  Thread 0 of the workgroup polls; 0 = timed out / aborted (both matrices are marked).
__device__ __forceinline__ int wg_wait_flag2(int* fa, int* fb, int* abort_word, int* info_a, int* info_b, int* sh) {
  if (threadIdx.x == 0) {
    int ok = 1;
    const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(fa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 ||
           __hip_atomic_load(fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      if (__builtin_amdgcn_s_memrealtime() - t_wait > GPG_TILE_WAIT_TICKS || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicMax(info_a, GPG_INFO_INTERNAL);
        atomicMax(info_b, GPG_INFO_INTERNAL);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    *sh = ok;
#ifdef GPG_STAMP
    t128_wait_acc += __builtin_amdgcn_s_memrealtime() - t_wait;
#endif
  }
  __syncthreads();
  const int ok = *sh;
  __syncthreads();
  if (ok) GPG_ACQUIRE();
  return ok;
}

// ------------------------------------------------------------------------------------------------
// fin128_offdiag (round 3): finalisation X <- X L_jj^-T of an off-diagonal 128 x 128 tile of the factorisation, for both 128-tile
// kernels (SET 0 / 1: the two teams of pair128_chol_kernel, SET 2: tile128_chol_kernel; one instance each, because a team's LDS arrays
// must be addressed statically).  What the timeline of round 3 showed (tools/timeline_pieces.py, 64 matrices of 2560 columns): of
// ~100 us per task only ~40 are the 128 column steps; 34 us are the tile going to memory and coming back in the quad layout (store,
// s_waitcnt, barrier, 32 loads), 42 us are eight times (flag poll, two barriers, a dependent load of 16 columns of L_jj, a barrier).
//  * Column block 0 reaches this function THROUGH LDS: wave (wm, 0) of the team has written its 64 x 64 accumulator block as
//    [column][row] -- rows 0..63 into U (row stride 80), rows 64..127 into the region of the L image viewed as [64][72]
//    (fin128_handoff_block0) -- and every thread takes its two rows in the quad layout from there.  Column block 1 (waves (wm, 1))
//    went to memory as before and is read back for its MFMA update much later.
//  * When the diagonal tile is COMPLETE on entry (its tile flag; the rule in batched launches) there is one check instead of nine
//    waits: the image of L11 is loaded in one piece, the 64 column steps run without a barrier, the raw L22 is fetched into registers
//    while block 1 takes its update.  Otherwise the piecewise path of rounds 1 / 2 runs (16 columns at a time behind the diagonal
//    task's early flags).  Both paths do the same arithmetic in the same order.
// done0 / done1: tile flags of the diagonal tiles of the two teams' matrices (the same word twice for SET 2), every other flag
// argument likewise.  Returns 0 if a wait timed out.
// ------------------------------------------------------------------------------------------------
template <int SET> __device__ __forceinline__ double* fin128_U() { if constexpr (SET == 2) return t128_U; else return p128_U[SET]; }
template <int SET> __device__ __forceinline__ double (*fin128_Ls())[4][18] { if constexpr (SET == 2) return t128_Ls; else return p128_Ls[SET]; }
template <int SET> __device__ __forceinline__ double* fin128_sdinv() { if constexpr (SET == 2) return t128_sdinv; else return p128_sdinv[SET]; }
template <int SET> __device__ __forceinline__ int* fin128_sh() { if constexpr (SET == 2) return &t128_sh_ok; else return &p128_sh_ok; }
template <int SET> __device__ __forceinline__ int* fin128_sh2() { if constexpr (SET == 2) return t128_sh2; else return p128_sh2; }
// the caller's side of the LDS hand-over (inlined into the task: the accumulators are in its registers)
template <int SET>
__device__ __forceinline__ void fin128_handoff_block0(const d4 (&acc)[4][4], int wm, int l15, int l4) {
  double* const U = fin128_U<SET>();
  double* const L2 = &fin128_Ls<SET>()[0][0][0];
#define GPG_F128_HAND(dst, stride)                                                           \
  _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                           \
    _Pragma("unroll") for (int g = 0; g < 2; ++g)                                            \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                        \
        double2 v;                                                                          \
        v.x = acc[ni][2 * g][r];                                                            \
        v.y = acc[ni][2 * g + 1][r];                                                        \
        *reinterpret_cast<double2*>((dst) + (32 * (ni >> 1) + 8 * r + 2 * l4 + (ni & 1)) * (stride) + 32 * g + 2 * l15) = v; \
      }
  if (wm == 0) { GPG_F128_HAND(U, 80) } else { GPG_F128_HAND(L2, 72) }
#undef GPG_F128_HAND
}
#ifdef GPG_STAMP
#define GPG_FS4(k) if (SET != 1 && threadIdx.x == 0 && t128_fo) t128_fo[k] = __builtin_amdgcn_s_memrealtime();
#else
#define GPG_FS4(k)
#endif
// fin128_onepiece: X <- X L^-T for a 128 x 128 tile whose column block 0 has been handed over in LDS (fin128_handoff_block0) and whose
// column block 1 sits in memory, against a COMPLETE diagonal tile L (reciprocal pivots dinv) -- the one-piece path of fin128_offdiag, also
// the finalisation of tile128_trinv_kernel (whose diagonal tiles are always complete).  The caller has synchronised the workgroup after
// the hand-over (and the stores of block 1); on return the stores of X are in flight: the caller's publish step waits for them.
// One-piece path (the diagonal tile was complete on entry).  Round 3, second half: the update of column block 1 no longer goes
// through memory.  Before: X1 stored, s_waitcnt vmcnt(0) + barrier (5.5 us of store acknowledgements), then wave_tile_gemm staging
// X1 and L21 from memory through LDS with three barriers per 16-deep chunk (14.8 us) -- for 128 MFMAs per wave.  Now L21 is brought
// in with L11 (one round trip) and parked k-major in the staging tile, the A operand -X1 comes straight out of the substitution's
// registers by a lane permutation (quad layout: lane 4 r + q holds row r, columns 4 m + q; MFMA layout: lane 16 k + r holds row r,
// column 4 s + k: source lane 4 l15 + l4, register m = s), the X1 store is left in flight until the task publishes, and the
// transposition back into the quad layout is wave-private (wave w owns rows 16 w .. 16 w + 15 in both layouts).  Two barriers
// instead of fourteen.  Same MFMA operands in the same order as GPG_F128_UPDATE (fn = L21 fragment, fm = -X1 fragment, accumulator
// preloaded with T2, k ascending): bit-identical with the piecewise path.
template <int SET>
__device__ __forceinline__ void fin128_onepiece(double* X, int ldx, const double* L, int ldl, const double* dinv) {
  constexpr int SA = 80, S2 = 72;
  double* const U = fin128_U<SET>();
  double (*const Ls)[4][18] = fin128_Ls<SET>();
  double* const sdinv = fin128_sdinv<SET>();
  int tid_raw = (int)threadIdx.x;
  asm volatile("" : "+v"(tid_raw));
  const int tid = SET == 2 ? tid_raw : (tid_raw & 255), lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int q = tid & 3, rr = tid >> 2;
  (void)l15; (void)l4;
  double x0[16], x1[16];
#define GPG_F128_TAKE_X()                                                                    \
  {                                                                                         \
    int tq = tid;                                                                           \
    asm volatile("" : "+v"(tq));                                                            \
    const double* T0 = U + (tq & 3) * SA + (tq >> 2);                                        \
    const double* T1 = &fin128_Ls<SET>()[0][0][0] + (tq & 3) * S2 + (tq >> 2);               \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                         \
      x0[m] = T0[(4 * m) * SA];                                                             \
      x1[m] = T1[(4 * m) * S2];                                                             \
    }                                                                                       \
  }
  GPG_ACQUIRE();
  GPG_F128_TAKE_X()
  double li[16], lj[16];                                     // raw 64 x 64 blocks of L, 16 entries per thread
#define GPG_F128_RAW(dst, LP)                                                                \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                         \
    const int t = tid + 256 * i;                                                          \
    dst[i] = (LP)[(t & 63) + (size_t)(t >> 6) * ldl];                                      \
  }
#define GPG_F128_IMAGE(DOFF)                                                                 \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                         \
    const int t = tid + 256 * i, jj = t >> 6, k = t & 63;                                  \
    Ls[jj][k & 3][k >> 2] = li[i];                                                        \
  }                                                                                       \
  if (tid < 64) sdinv[tid] = dinv[(DOFF) + tid];
  GPG_F128_RAW(li, L)
  GPG_F128_RAW(lj, L + 64)                                   // L21[n][k] = L[64 + n + k ldl]
  GPG_LDS_BARRIER();                                         // every thread has taken x0 / x1 out of the staging tile and the image region
  GPG_F128_IMAGE(0)
#pragma unroll
  for (int i = 0; i < 16; ++i) {                             // L21 k-major: U[k][n], row stride SA
    const int t = tid + 256 * i;
    U[(t >> 6) * SA + (t & 63)] = lj[i];
  }
  GPG_F128_RAW(li, L + 64 + (size_t)64 * ldl)                // L22: in flight through the column steps of block 0
  GPG_LDS_BARRIER();
  GPG_FS4(2)
  {
    double* const Xs = X + rr + (size_t)q * ldx;             // the finished columns of X1 leave behind the steps (no wait before the task publishes)
#define GPG_F128_HOOK(mj) GPG_ST(&Xs[(size_t)(4 * (mj)) * ldx], x0[mj]); GPG_ST(&Xs[64 + (size_t)(4 * (mj)) * ldx], x1[mj]);
    GPG_QUAD_SUBST2_HOOK(x0, x1, Ls, sdinv, q, GPG_F128_HOOK)
  }
  GPG_FS4(3)
  // (addresses used from here on are re-derived from an opaque copy of the thread index: computed ahead of the column steps -- they are
  // pure functions of it -- they were kept in ~50 registers across the steps and spilled around them)
  int tid2 = tid;
  asm volatile("" : "+v"(tid2));
#define GPG_F128_RELANE(t)  const int lane_ = (t) & 63, l15 = lane_ & 15, l4 = lane_ >> 4, q = (t) & 3, rr = (t) >> 2, tid = (t); (void)l15; (void)l4; (void)q; (void)rr; (void)tid;
  d4 c0[4], c1[4];                                           // T2 in the MFMA layout: rows 16 w + l15 (c0) and 64 + 16 w + l15 (c1), column 64 + 16 ni + 4 r + l4
  {
    GPG_F128_RELANE(tid2)
    const double* Cw = X + 16 * w + l15 + (size_t)(64 + l4) * ldx;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c0[ni][r] = Cw[(size_t)(ni * 16 + 4 * r) * ldx];
        c1[ni][r] = Cw[64 + (size_t)(ni * 16 + 4 * r) * ldx];
      }
  }
  double a0[16], a1[16];
  {
    GPG_F128_RELANE(tid2)
    const int src = 4 * (4 * l15 + l4);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int lo0 = __builtin_amdgcn_ds_bpermute(src, __double2loint(x0[m])), hi0 = __builtin_amdgcn_ds_bpermute(src, __double2hiint(x0[m]));
      const int lo1 = __builtin_amdgcn_ds_bpermute(src, __double2loint(x1[m])), hi1 = __builtin_amdgcn_ds_bpermute(src, __double2hiint(x1[m]));
      a0[m] = -__hiloint2double(hi0, lo0);
      a1[m] = -__hiloint2double(hi1, lo1);
    }
  }
  GPG_FS3(6)
  GPG_LDS_BARRIER();                                         // (1) every wave is through the column steps of block 0: the image region is free
  GPG_FS4(4)
  {
    GPG_F128_RELANE(tid2)
    GPG_F128_IMAGE(64)
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
      double fn[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) fn[ni] = U[(4 * s4 + l4) * SA + ni * 16 + l15];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        c0[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn[ni], a0[s4], c0[ni], 0, 0, 0);
        c1[ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn[ni], a1[s4], c1[ni], 0, 0, 0);
      }
    }
  }
  GPG_LDS_BARRIER();                                         // (2) L21 is no longer read from the staging tile; the image of L22 is complete
  {
    GPG_F128_RELANE(tid2)
    const double* Tr = U + q * SA + rr;                      // rows 16 w .. 16 w + 15 are written and read by wave w only: no barrier
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) U[(ni * 16 + 4 * r + l4) * SA + 16 * w + l15] = c0[ni][r];
#pragma unroll
    for (int m = 0; m < 16; ++m) x0[m] = Tr[(4 * m) * SA];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) U[(ni * 16 + 4 * r + l4) * SA + 16 * w + l15] = c1[ni][r];
#pragma unroll
    for (int m = 0; m < 16; ++m) x1[m] = Tr[(4 * m) * SA];
  }
  GPG_FS4(5)
  {
    GPG_F128_RELANE(tid2)
    double* const Xs = X + rr + (size_t)(64 + q) * ldx;
    GPG_QUAD_SUBST2_HOOK(x0, x1, Ls, sdinv, q, GPG_F128_HOOK)
#undef GPG_F128_HOOK
  }
  GPG_FS4(6)
  GPG_FS3(0)
  GPG_FS3(1)
#undef GPG_F128_RELANE
#undef GPG_F128_RAW
#undef GPG_F128_IMAGE
#undef GPG_F128_TAKE_X
}

#ifndef GPG_FIN_INLINE
#define GPG_FIN_INLINE 1
#endif
#if GPG_FIN_INLINE
#define GPG_FIN_FN __device__ __forceinline__
#else
#define GPG_FIN_FN __device__ __noinline__
#endif
// Called with the ticket and the kernel's argument segment only: everything else is re-derived here by scalar loads.  What a task
// keeps alive across its MFMA loop is spilled, and every spill reload after the loop is a memory round trip under full load (the
// timeline of the first version of this function showed ~35 us of them in front of the call).
template <int SET>
GPG_FIN_FN int fin128_offdiag(int tix_v, unsigned long long kernarg_bits) {
  constexpr int SA = 80, S2 = 72, BUF = 16 * SA;
  GPG_KERNARGS_FROM(TileCholArgs, ap, kernarg_bits);
  const int tix = __builtin_amdgcn_readfirstlane(tix_v);
  const int Mt = ap->Mt, ld = ap->ld;
  int task, b0, b1;
  if constexpr (SET == 2) {
    task = ap->tasks[tix];
    b0 = b1 = ap->batch_of ? ap->batch_of[tix] : 0;
  } else {
    task = ap->tasks[2 * tix + SET];
    b0 = ap->batch_of[2 * tix];
    b1 = ap->batch_of[2 * tix + 1];
  }
  const int ti = task & 0xffff, tj = task >> 16, b = SET == 1 ? b1 : b0;
  double* const A = ap->A + (size_t)b * ap->a_stride;
  double* const dinv_m = ap->dinv + (size_t)b * ap->d_stride;
  int* const info0 = ap->info + b0;
  int* const info1 = ap->info + b1;
  int* const fl0 = ap->flags + (size_t)b0 * ap->f_stride;
  int* const fl1 = ap->flags + (size_t)b1 * ap->f_stride;
  int* const ea0 = ap->pieces + (size_t)b0 * ap->f_stride;
  int* const ea1 = ap->pieces + (size_t)b1 * ap->f_stride;
  int* const pa0 = ea0 + 4 * tj; int* const pa1 = ea1 + 4 * tj;
  int* const pb0 = ea0 + 4 * Mt + 4 * tj; int* const pb1 = ea1 + 4 * Mt + 4 * tj;
  int* const fc0 = ea0 + 8 * Mt + tj; int* const fc1 = ea1 + 8 * Mt + tj;
  int* const done0 = fl0 + (size_t)tj * Mt + tj; int* const done1 = fl1 + (size_t)tj * Mt + tj;
  int* const abort_word = ap->abort_word;
  const size_t r0 = 128 * (size_t)ti, cj = 128 * (size_t)tj;
  double* const U = fin128_U<SET>();
  double (*const Ls)[4][18] = fin128_Ls<SET>();
  double* const sdinv = fin128_sdinv<SET>();
  int* const sh = fin128_sh<SET>();
  int* const sh2 = fin128_sh2<SET>();
  int tid_raw = (int)threadIdx.x;
  asm volatile("" : "+v"(tid_raw));                            // (inlined: nothing derived from the lane index may be hoisted above the task's MFMA loop)
  const int tid = SET == 2 ? tid_raw : (tid_raw & 255), lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int q = tid & 3, rr = tid >> 2;
  const int sp = tid & 31, sk = tid >> 5;
  const double* const L = A + cj + cj * (size_t)ld;            // the diagonal tile of this team's matrix
  const double* const dinv = dinv_m + cj;
  double* const X = A + r0 + cj * (size_t)ld;
  const int ldl = ld, ldx = ld;
  GPG_FS4(0)
  if (threadIdx.x == 0)
    *sh = __hip_atomic_load(done0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 &&
          __hip_atomic_load(done1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  __syncthreads();                                             // the hand-over is in LDS, the verdict on the diagonal tile in *sh
  const int fast = *sh;
  GPG_FS4(1)
#ifdef GPG_STAMP
  if (SET != 1 && threadIdx.x == 0 && t128_fo) t128_fo[7] = fast ? 2 : 1;
#endif
  double x0[16], x1[16];
  // (taken out of the hand-over tiles separately on each path, through an opaque copy of the thread index: loaded once ahead of the
  // branch, the initial values stayed live -- in scratch -- through the whole one-piece path)
#define GPG_F128_TAKE_X()                                                                    \
  {                                                                                         \
    int tq = tid;                                                                           \
    asm volatile("" : "+v"(tq));                                                            \
    const double* T0 = U + (tq & 3) * SA + (tq >> 2);                                        \
    const double* T1 = &fin128_Ls<SET>()[0][0][0] + (tq & 3) * S2 + (tq >> 2);               \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                         \
      x0[m] = T0[(4 * m) * SA];                                                             \
      x1[m] = T1[(4 * m) * S2];                                                             \
    }                                                                                       \
  }
  // column block 1 after its update: T2 -= X1 L21^T for the two row halves, each transposed through the LDS tile into x0 / x1
#define GPG_F128_UPDATE()                                                                    \
  _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                          \
    d4 acc[4];                                                                              \
    const double* Cw = X + 64 * hh + 16 * w + l15 + (size_t)(64 + l4) * ldx;                 \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                         \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[ni][r] = Cw[(size_t)(ni * 16 + 4 * r) * ldx]; \
    wave_tile_gemm(acc, X + 64 * hh + 2 * sp + (size_t)sk * ldx, ldx, L + 64 + 2 * sp + (size_t)sk * ldl, ldl, 4, U, U + 2 * BUF, \
                   w, l15, l4, sp, sk);                                                     \
    __syncthreads();                                                                        \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                         \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) U[(ni * 16 + 4 * r + l4) * SA + 16 * w + l15] = acc[ni][r]; \
    __syncthreads();                                                                        \
    const double* Tr = U + q * SA + rr;                                                     \
    if (hh == 0) { _Pragma("unroll") for (int m = 0; m < 16; ++m) x0[m] = Tr[(4 * m) * SA]; } \
    else { _Pragma("unroll") for (int m = 0; m < 16; ++m) x1[m] = Tr[(4 * m) * SA]; }        \
    __syncthreads();                                                                        \
  }
#define GPG_F128_STORE(COL0)                                                                 \
  {                                                                                         \
    double* Xr = X + rr + (size_t)((COL0) + q) * ldx;                                        \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                         \
      GPG_ST(&Xr[(size_t)(4 * m) * ldx], x0[m]);                                             \
      GPG_ST(&Xr[64 + (size_t)(4 * m) * ldx], x1[m]);                                        \
    }                                                                                       \
  }
  if (fast) {
    fin128_onepiece<SET>(X, ldx, L, ldl, dinv);
    return 1;                                                  // (the caller's publish step waits for the stores and synchronises the workgroup)
  }
  GPG_F128_TAKE_X()
  __syncthreads();                                             // x1 is out of the image region, *sh may be rewritten
  // piecewise: L11 / L22 arrive in four 16-column pieces each while the diagonal task is still at work
  // A piece whose flag is already up while the previous piece is being substituted is fetched into registers behind that
  // substitution (pf = 1): its turn then starts with the LDS write instead of a flag poll, two barriers and a dependent load.
  double lp4[4];
  int pf = 0;
#define GPG_F128_PIECE(FL, LP, DOFF, S)                                                      \
  {                                                                                         \
    if (!pf) {                                                                              \
      if (!wg_wait_flag2(FL##0 + (S), FL##1 + (S), abort_word, info0, info1, sh)) return 0;  \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                        \
        const int t = tid + 256 * u, jj = 16 * (S) + (t >> 6), k = t & 63;                   \
        lp4[u] = (LP)[k + (size_t)jj * ldl];                                                 \
      }                                                                                     \
    }                                                                                       \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                          \
      const int t = tid + 256 * u, jj = 16 * (S) + (t >> 6), k = t & 63;                     \
      Ls[jj][k & 3][k >> 2] = lp4[u];                                                        \
    }                                                                                       \
    if (tid < 16) sdinv[16 * (S) + tid] = dinv[(DOFF) + 16 * (S) + tid];                     \
    if ((S) < 3 && threadIdx.x == 0)                                                         \
      sh2[(S) & 1] = __hip_atomic_load(FL##0 + (S) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 && \
             __hip_atomic_load(FL##1 + (S) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;  \
    __syncthreads();                                                                        \
    pf = (S) < 3 ? sh2[(S) & 1] : 0;                                                        \
    if (pf) {                                                                               \
      GPG_ACQUIRE();                                                                        \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                        \
        const int t = tid + 256 * u, jj = 16 * ((S) + 1) + (t >> 6), k = t & 63;             \
        lp4[u] = (LP)[k + (size_t)jj * ldl];                                                 \
      }                                                                                     \
    }                                                                                       \
    GPG_QUAD_SUBST2_PIECE(x0, x1, Ls, sdinv, q, S)                                           \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) { asm volatile("" : "+v"(x0[m])); asm volatile("" : "+v"(x1[m])); } \
  }
  GPG_F128_PIECE(pa, L, 0, 0)
  GPG_F128_PIECE(pa, L, 0, 1)
  GPG_F128_PIECE(pa, L, 0, 2)
  GPG_F128_PIECE(pa, L, 0, 3)
  GPG_FS4(2)
  GPG_F128_STORE(0)
  if (!wg_wait_flag2(fc0, fc1, abort_word, info0, info1, sh)) return 0;   // barrier inside: X1 in memory, L21 final
  GPG_FS4(3)
  GPG_F128_UPDATE()
  GPG_FS4(4)
  {
    const double* L22 = L + 64 + (size_t)64 * ldl;
    GPG_F128_PIECE(pb, L22, 64, 0)
    GPG_F128_PIECE(pb, L22, 64, 1)
    GPG_F128_PIECE(pb, L22, 64, 2)
    GPG_F128_PIECE(pb, L22, 64, 3)
  }
  GPG_FS4(5)
  GPG_F128_STORE(64)
  __syncthreads();
  GPG_FS4(6)
  return 1;
#undef GPG_F128_PIECE
#undef GPG_F128_UPDATE
#undef GPG_F128_STORE
#undef GPG_F128_TAKE_X
}


// fin128_diag: finalisation of a DIAGONAL 128 x 128 tile (both 128-tile kernels; SET as in fin128_offdiag).  On entry the top-left
// 64 x 64 block sits in the team's LDS tile (written by wave 0 of the team), A21 / A22 in memory:
//     L11 = chol(A11) (potrf64_wg; published in four 16-column pieces, early flags pa), L21 = A21 L11^-T (flag_c),
//     A22 -= L21 L21^T on MFMA, L22 = chol(A22) (pieces pb).
// Arguments as for fin128_offdiag: the ticket and the kernel's argument segment.
template <int SET>
GPG_FIN_FN int fin128_diag(int tix_v, unsigned long long kernarg_bits) {
  constexpr int SA = 80;
  GPG_KERNARGS_FROM(TileCholArgs, ap, kernarg_bits);
  const int tix = __builtin_amdgcn_readfirstlane(tix_v);
  const int Mt = ap->Mt, ld = ap->ld, N = ap->N;
  int task, b;
  if constexpr (SET == 2) {
    task = ap->tasks[tix];
    b = ap->batch_of ? ap->batch_of[tix] : 0;
  } else {
    task = ap->tasks[2 * tix + SET];
    b = ap->batch_of[2 * tix + SET];
  }
  const int tj = task >> 16;
  double* const A = ap->A + (size_t)b * ap->a_stride;
  double* const dinv = ap->dinv + (size_t)b * ap->d_stride;
  int* const info = ap->info + b;
  int* const early = ap->pieces + (size_t)b * ap->f_stride;
  int* const pa = early + 4 * tj;
  int* const pb = early + 4 * Mt + 4 * tj;
  int* const flag_c = early + 8 * Mt + tj;
  const size_t cj = 128 * (size_t)tj;
  double* const U = fin128_U<SET>();
  double (*const Ls)[4][18] = fin128_Ls<SET>();
  double* const sdinv = fin128_sdinv<SET>();
  int tid_raw = (int)threadIdx.x;
  asm volatile("" : "+v"(tid_raw));
  const int tid = SET == 2 ? tid_raw : (tid_raw & 255), lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  double* blk = A + cj + cj * (size_t)ld;
  double (*St)[64] = reinterpret_cast<double(*)[64]>(&Ls[0][0][0]);   // potrf scratch over the image region
  __syncthreads();   // A11 was left in the LDS tile by wave 0; the other waves' stores of A21 / A22 are drained below
  GPG_FS4(0)
  {
    const int bad = potrf64_wg(U, SA, St, blk, ld, dinv + cj, pa, tid);   // L11 published in 16-column pieces pa[0..3]
    if (w == 0 && bad && lane == 0 && (int)cj + bad - 1 < N) atomicCAS(info, 0, (int)cj + bad);
  }
  __syncthreads();   // also drains the other waves' stores of A21 / A22
  GPG_FS4(1)
  panel_solve_rows64(blk, ld, dinv + cj, blk + 64, ld, 64, 64, U, Ls, sdinv, tid);   // L21 = A21 L11^-T
  GPG_RELEASE();     // L21 is final (every thread stored part of it; the solve ended with a barrier)
  __syncthreads();
  if (tid == 0) GPG_FLAG_UP(flag_c);
  GPG_FS4(2)
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
  GPG_FS4(3)
  {
    const int bad = potrf64_wg(U, SA, St, blk + 64 + (size_t)64 * ld, ld, dinv + cj + 64, pb, tid);   // L22: pb[0..3]
    if (w == 0 && bad && lane == 0 && (int)cj + 64 + bad - 1 < N) atomicCAS(info, 0, (int)cj + 64 + bad);
  }
  GPG_FS4(4)
  return 1;
}

// ONE call site per task: with a call in each branch of `diagonal ? ... : ...` (and an early return after each) the compiler lays the two
// branches out one after the other, the accumulators of the second stay live across the call of the first, and all 240 live VGPRs
// are saved to scratch and reloaded around it (seen in the ISA, round 3).
template <int SET>
GPG_FIN_FN int fin128_task(int tix_v, unsigned long long kernarg_bits, int is_diag) {
  return __builtin_amdgcn_readfirstlane(is_diag) ? fin128_diag<SET>(tix_v, kernarg_bits) : fin128_offdiag<SET>(tix_v, kernarg_bits);
}

__device__ __forceinline__ bool
tile128_chol_task(int tix, double* A, int ld, int Mt, const int* __restrict__ tasks, int* flags, int* early /* pa | pb | flag_c */, int* abort_word,
                  int* ticket, double* __restrict__ dinv, int* __restrict__ info, int N, const int* __restrict__ batch_of, size_t a_stride,
                  int d_stride, int f_stride, unsigned long long kbits) {
  __shared__ int sh_kr;
  int tid_task = threadIdx.x;
  asm volatile("" : "+v"(tid_task));   // per task: nothing derived from the lane index is kept alive from one task to the next (across its MFMA loop and finalisation)
  const int tid = tid_task, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int task = tasks[tix];
  const int ti = task & 0xffff, tj = task >> 16;
  if (batch_of) {   // batched launch: several independent matrices (restart rows) share the grid
    const int b = batch_of[tix];
    A += (size_t)b * a_stride;
    dinv += (size_t)b * d_stride;
    info += b;
    flags += (size_t)b * f_stride;
    early += (size_t)b * f_stride;
  }
  const size_t r0 = 128 * (size_t)ti, cj = 128 * (size_t)tj;
  int* const frow_i = flags + (size_t)ti * Mt;
  int* const frow_j = flags + (size_t)tj * Mt;

#ifdef GPG_STAMP
  const unsigned long long tk_start = __builtin_amdgcn_s_memrealtime();
  unsigned long long tk_spin = 0, tk_gemm = 0, tk_runs = 0;
#endif
  // accumulator layout of direct_tile_gemm_acc: acc[2p + e][2g + m][r] <-> row 32 g + 2 l15 + m, column 32 p + 2 (4 r + l4) + e
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
  GPG_PRIO_ACC(ti - tj)
  direct_tile_negate(acc);                                   // the loop ADDS the products to -A_ij (direct_tile_gemm_acc); sign restored below
  const unsigned lane_off = (unsigned)(2 * l15 + l4 * ld) * 8u;   // bytes: this lane's rows 2 t, 2 t + 1 of column l4 of a k-step
  int kdone = 0;
  while (kdone < tj) {
    GPG_TR(q0)
    if (w == 0) {                                           // wave 0 scans the two flag rows, 64 tile columns per pass
      int* const frows[2] = {frow_i, frow_j};
      int kr;
      const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        kr = wave_scan_flags<2>(frows, kdone, tj);
        if (kr > kdone) break;
        if (__builtin_amdgcn_s_memrealtime() - t_wait > GPG_TILE_WAIT_TICKS || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          if (lane == 0) {
            __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicMax(info, GPG_INFO_INTERNAL);
          }
          kr = -1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      if (lane == 0) sh_kr = kr;
    }
    __syncthreads();
    const int kr = __builtin_amdgcn_readfirstlane(sh_kr);
    if (kr < 0) return false;
    GPG_ACQUIRE();
    GPG_TR(q1)
    const size_t ck = 128 * (size_t)kdone;
    if (ti < Mt) {
      if (!(ti == tj && wm == 0 && wn == 1))   // diagonal tile: the block above the diagonal is never stored
        direct_tile_gemm_acc<GPG_MFMA_PF>(acc, A + r0 + wm * 64 + ck * (size_t)ld, lane_off, ld,
                                A + GPG_FAKE_B_ROW(cj) + wn * 64 + ck * (size_t)ld, lane_off, ld, 32 * (kr - kdone));
      else direct_tile_sync_only<GPG_MFMA_PF>(32 * (kr - kdone));
    } else if (wm == 0)   // right-hand-side tile row (prep_rows_kernel): rows 0 and 1 are real, the other 126 zero -- an eighth of the MFMAs
      direct_tile_gemm_acc<GPG_MFMA_PF, 2>(acc, A + r0 + ck * (size_t)ld, lane_off, ld,
                                 A + cj + wn * 64 + ck * (size_t)ld, lane_off, ld, 32 * (kr - kdone));
    else direct_tile_sync_only<GPG_MFMA_PF>(32 * (kr - kdone));
    __syncthreads();   // sh_kr may be rewritten
    GPG_TR(q2)
#ifdef GPG_STAMP
    tk_spin += q1 - q0; tk_gemm += q2 - q1; ++tk_runs;
#endif
    kdone = kr;
  }

#ifdef GPG_STAMP
  const unsigned long long tk_fin0 = __builtin_amdgcn_s_memrealtime();
#endif
  GPG_PRIO_FIN(ti - tj)
  direct_tile_negate(acc);
  // (the store addresses are formed HERE: derived from Cw before the loop, the compiler keeps all 32 of them alive across the MFMA
  // loop, spilled, and reloads them one memory round trip at a time)
  double* Cs = Cw;
  asm volatile("" : "+v"(Cs));
  // (likewise the lane indices: every LDS / memory address below is a pure function of the lane, i.e. invariant over the kernel's
  // task loop -- hoisted out of it, ~130 of them stay alive through every MFMA loop and every call, in scratch)
  int l15s = l15, l4s = l4;
  asm volatile("" : "+v"(l15s), "+v"(l4s));
  // ---- (2) Diagonal tile: the top-left 64 x 64 block goes straight into the LDS tile its own wave factors next, the strictly
  //      upper block is dropped, the rest goes to memory and is finalised in place (fin128_diag).  Tile below the diagonal:
  //      column block 0 is handed to fin128_offdiag through LDS, column block 1 goes to memory. -----------------------------------
  if (ti == tj) {
    if (w == 0) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            t128_U[(32 * (ni >> 1) + 8 * r + 2 * l4s + (ni & 1)) * 80 + 32 * (mi >> 1) + 2 * l15s + (mi & 1)] = acc[ni][mi][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (wm != 0) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double2 v;
            v.x = acc[ni][2 * g][r];
            v.y = acc[ni][2 * g + 1][r];
            *reinterpret_cast<double2*>(Cs + 32 * g + (size_t)(32 * (ni >> 1) + 8 * r + (ni & 1)) * ld) = v;
          }
    }
#ifdef GPG_STAMP
    if (tid == 0) t128_fo = nullptr;
#endif
  } else {
    if (wn == 0) fin128_handoff_block0<2>(acc, wm, l15s, l4s);
    else {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double2 v;
            v.x = acc[ni][2 * g][r];
            v.y = acc[ni][2 * g + 1][r];
            *reinterpret_cast<double2*>(Cs + 32 * g + (size_t)(32 * (ni >> 1) + 8 * r + (ni & 1)) * ld) = v;
          }
    }
#ifdef GPG_STAMP
    if (tid == 0) t128_fo = (g_stamp_buf != nullptr && tix < GPG_STAMP_MAX) ? g_stamp_buf + (size_t)GPG_STAMP_MAX * 8 + (size_t)tix * 8 : nullptr;
#endif
  }
  if (fin128_task<2>(tix, kbits, ti == tj) == 0) return false;
  // ---- (3) publish (and fetch the next ticket) --------------------------------------------------------------------
  GPG_PUBLISH_AND_NEXT(ticket, frow_i + tj)
#ifdef GPG_STAMP
  if (tid == 0 && g_stamp_buf != nullptr && tix < GPG_STAMP_MAX) {
    unsigned long long* o = g_stamp_buf + (size_t)tix * 8;
    o[0] = tk_start; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = tk_spin; o[3] = tk_gemm; o[4] = tk_runs;
    o[5] = tk_fin0; o[6] = (unsigned long long)task; o[7] = (unsigned long long)blockIdx.x;
  }
#endif
  return true;
}

__global__ void __launch_bounds__(256, 2) tile128_chol_kernel(TileCholArgs) {   // c0 unused, pieces = early flags
  int tix;
  {
    GPG_KERNARGS(TileCholArgs, ap);
    tix = next_ticket(ap->ticket, &g_next_ticket, 0);
  }
  for (;;) {
    GPG_KERNARGS(TileCholArgs, ap);
    if (tix >= ap->ntask) return;
    // inlined here (its MFMA loop wants every VGPR: a separate function would save / restore ~100 callee-saved registers
    // through scratch per task); re-reading the arguments per iteration keeps them out of the loop-carried SGPRs
    if (!tile128_chol_task(tix, ap->A, ap->ld, ap->Mt, ap->tasks, ap->flags, ap->pieces, ap->abort_word, ap->ticket, ap->dinv, ap->info,
                           ap->N, ap->batch_of, ap->a_stride, ap->d_stride, ap->f_stride, (unsigned long long)ap))
      return;                                             // aborted: every workgroup drains
    tix = __builtin_amdgcn_readfirstlane(g_next_ticket);
  }
}



// ------------------------------------------------------------------------------------------------
// pair128_chol_kernel (round 3): the same dataflow factorisation for BATCHED launches, 512 threads = two 256-thread teams per
// workgroup, ONE workgroup per compute unit.  A task is a PAIR of 128 x 128 tiles of one tile column j that go through the same
// control flow: team h owns tile (i_h, j) of matrix b_h.  Ordinary pairs are two consecutive rows of one matrix: both teams then
// read the SAME row panel L_j,0:j as their B operand, k-step by k-step at the same pace -- eight waves that start a run together and
// execute the same instruction stream stay together by themselves: a workgroup barrier in the loop (GPG_PAIR_KSYNC k-steps apart;
// 0 = none, the default) only adds its skew: 283.4 ms and 3.90e8 KiB FETCH_SIZE without, 289.0 ms and 4.08e8 with one every 8 k-steps
// -- so it crosses the L2 -> L1 path once per pair: the operand stream of a 256 x 128 footprint, 25 % fewer bytes than two independent
// 128 x 128 tiles.  Diagonal tiles pair with the diagonal tile of the NEXT matrix of the batch
// (same j, same barrier sequence, different data); what is left over in a tile column pairs with the next matrix's leftover,
// and a tile without any partner is simply given to both teams (identical instruction streams on identical data: the duplicate
// stores write identical values).  Every barrier is a full-workgroup barrier: the two teams never diverge in control flow, and
// every decision that depends on flags (how far the next MFMA run goes, which 16-column piece of L_jj is final) is taken by
// thread 0 for both tiles at once.  Flags, early flags, ticket order and the progress argument are those of tile128_chol_kernel:
// a pair waits only for tiles of earlier tile columns and for the diagonal tiles of its own column, which precede it in the list.
// What it gives up: with one workgroup per CU the finalisation of a pair is not overlapped with another workgroup's MFMA loop.
// ------------------------------------------------------------------------------------------------
#ifndef GPG_PAIR_KSYNC
#define GPG_PAIR_KSYNC 0
#endif
#ifndef GPG_PAIR_PF
#define GPG_PAIR_PF GPG_MFMA_PF   // k-steps of operand prefetch in the pair kernel's MFMA loop (its eight waves run in lock step: nobody covers a late load)
#endif
__device__ __forceinline__ bool
pair128_chol_task(int tix, double* Abase, int ld, int Mt, const int* __restrict__ tasks, int* flags_base, int* early_base, int* abort_word,
                  int* ticket, double* __restrict__ dinv_base, int* __restrict__ info_base, int N, const int* __restrict__ batch_of,
                  size_t a_stride, int d_stride, int f_stride, unsigned long long kbits) {
  __shared__ int sh_kr;
  int tid_task = threadIdx.x;
  asm volatile("" : "+v"(tid_task));   // (see tile128_chol_task)
  const int tid = tid_task, h = __builtin_amdgcn_readfirstlane(tid >> 8), t = tid & 255, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  // both teams' assignments (uniform values; [h] selects this thread's)
  const int task0 = tasks[2 * tix], task1 = tasks[2 * tix + 1];
  const int b0 = batch_of[2 * tix], b1 = batch_of[2 * tix + 1];
  const int tj = task0 >> 16;                                   // common tile column; both tiles diagonal or both below the diagonal
  const int ti0 = task0 & 0xffff, ti1 = task1 & 0xffff;
  const int ti = h ? ti1 : ti0, b = h ? b1 : b0;
  double* const A = Abase + (size_t)b * a_stride;
  double* const dinv = dinv_base + (size_t)b * d_stride;
  int* const fl0 = flags_base + (size_t)b0 * f_stride;
  int* const fl1 = flags_base + (size_t)b1 * f_stride;
  int* const ea0 = early_base + (size_t)b0 * f_stride;
  int* const ea1 = early_base + (size_t)b1 * f_stride;
  int* const info0 = info_base + b0;
  int* const info1 = info_base + b1;
  const size_t r0 = 128 * (size_t)ti, cj = 128 * (size_t)tj;

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

#ifdef GPG_STAMP
  const unsigned long long tk_start = __builtin_amdgcn_s_memrealtime();
  unsigned long long tk_spin = 0, tk_gemm = 0, tk_runs = 0;
#endif
  // ---- (1) left-looking accumulation: runs as far as the flags of BOTH tiles' rows (and of tile row j of both matrices) allow ----
  direct_tile_negate(acc);                                   // the loop adds the products to -A_ij; sign restored below
  const unsigned lane_off = (unsigned)(2 * l15 + l4 * ld) * 8u;
  int kdone = 0;
  while (kdone < tj) {
    GPG_TR(q0)
    if (tid < 64) {                                         // wave 0 of team 0 scans the four flag rows, 64 tile columns per pass
      int* const frows[4] = {fl0 + (size_t)ti0 * Mt, fl0 + (size_t)tj * Mt, fl1 + (size_t)ti1 * Mt, fl1 + (size_t)tj * Mt};
      int kr;
      const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        kr = wave_scan_flags<4>(frows, kdone, tj);
        if (kr > kdone) break;
        if (__builtin_amdgcn_s_memrealtime() - t_wait > GPG_TILE_WAIT_TICKS || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          if (tid == 0) {
            __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicMax(info0, GPG_INFO_INTERNAL);
            atomicMax(info1, GPG_INFO_INTERNAL);
          }
          kr = -1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      if (tid == 0) sh_kr = kr;
    }
    __syncthreads();
    const int kr = __builtin_amdgcn_readfirstlane(sh_kr);
    if (kr < 0) return false;
    GPG_ACQUIRE();
    GPG_TR(q1)
    const size_t ck = 128 * (size_t)kdone;
    const int nstep = 32 * (kr - kdone);
    if (ti < Mt) {
      if (!(ti == tj && wm == 0 && wn == 1))
        direct_tile_gemm_acc<GPG_PAIR_PF, 4, GPG_PAIR_KSYNC>(acc, A + r0 + wm * 64 + ck * (size_t)ld, lane_off, ld,
                                                            A + cj + wn * 64 + ck * (size_t)ld, lane_off, ld, nstep);
      else direct_tile_sync_only<GPG_PAIR_PF, GPG_PAIR_KSYNC>(nstep);
    } else if (wm == 0)
      direct_tile_gemm_acc<GPG_PAIR_PF, 2, GPG_PAIR_KSYNC>(acc, A + r0 + ck * (size_t)ld, lane_off, ld,
                                                          A + cj + wn * 64 + ck * (size_t)ld, lane_off, ld, nstep);
    else direct_tile_sync_only<GPG_PAIR_PF, GPG_PAIR_KSYNC>(nstep);
    __syncthreads();
    GPG_TR(q2)
#ifdef GPG_STAMP
    tk_spin += q1 - q0; tk_gemm += q2 - q1; ++tk_runs;
#endif
    kdone = kr;
  }
#ifdef GPG_STAMP
  const unsigned long long tk_fin0 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- (2) diagonal tiles: top-left block into the team's LDS tile, the rest to memory; tiles below the diagonal: column block 0
  //      into LDS for fin128_offdiag, column block 1 to memory -----------------------------------------------------------------------
  direct_tile_negate(acc);
  double* Cs = Cw;                                            // (see tile128_chol_task)
  asm volatile("" : "+v"(Cs));
  int l15s = l15, l4s = l4;
  asm volatile("" : "+v"(l15s), "+v"(l4s));
  if (ti0 == tj) {
    if (w == 0) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            (h ? p128_U[1] : p128_U[0])[(32 * (ni >> 1) + 8 * r + 2 * l4s + (ni & 1)) * 80 + 32 * (mi >> 1) + 2 * l15s + (mi & 1)] = acc[ni][mi][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (wm != 0) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double2 v;
            v.x = acc[ni][2 * g][r];
            v.y = acc[ni][2 * g + 1][r];
            *reinterpret_cast<double2*>(Cs + 32 * g + (size_t)(32 * (ni >> 1) + 8 * r + (ni & 1)) * ld) = v;
          }
    }
  } else {
    if (wn == 0) {
      if (h) fin128_handoff_block0<1>(acc, wm, l15s, l4s); else fin128_handoff_block0<0>(acc, wm, l15s, l4s);
    } else {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double2 v;
            v.x = acc[ni][2 * g][r];
            v.y = acc[ni][2 * g + 1][r];
            *reinterpret_cast<double2*>(Cs + 32 * g + (size_t)(32 * (ni >> 1) + 8 * r + (ni & 1)) * ld) = v;
          }
    }
  }
#ifdef GPG_STAMP
  if (tid == 0) {
    t128_fo = (g_stamp_buf != nullptr && tix < GPG_STAMP_MAX) ? g_stamp_buf + (size_t)GPG_STAMP_MAX * 8 + (size_t)tix * 8 : nullptr;
    t128_wait_acc = 0;
    if (t128_fo) for (int k = 0; k < 8; ++k) t128_fo[k] = 0;
  }
  GPG_FS3(5)
#endif
  if ((h ? fin128_task<1>(tix, kbits, ti0 == tj) : fin128_task<0>(tix, kbits, ti0 == tj)) == 0) return false;
#ifdef GPG_STAMP
  if (tid == 0 && t128_fo) (g_stamp_buf + (size_t)GPG_STAMP_MAX * 16 + (size_t)tix * 16)[15] = t128_wait_acc;
  GPG_FS3(2)
#endif
  // ---- (3) publish both tiles (and fetch the next ticket) --------------------------------------------------------------------------
  {
    int nxt_ = 0;
    if (tid == 0) nxt_ = GPG_TICKET_FETCH(ticket);
    GPG_RELEASE();
    GPG_FS3(3)
    if (tid == 0) g_next_ticket = nxt_;
    __syncthreads();
    GPG_FS3(4)
    if (tid == 0) {
      GPG_FLAG_UP(fl0 + (size_t)ti0 * Mt + tj);
      GPG_FLAG_UP(fl1 + (size_t)ti1 * Mt + tj);
    }
  }
#ifdef GPG_STAMP
  if (tid == 0 && g_stamp_buf != nullptr && tix < GPG_STAMP_MAX) {
    unsigned long long* o = g_stamp_buf + (size_t)tix * 8;
    o[0] = tk_start; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = tk_spin; o[3] = tk_gemm; o[4] = tk_runs;
    o[5] = tk_fin0; o[6] = (unsigned long long)task0; o[7] = (unsigned long long)blockIdx.x;
  }
#endif
  return true;
}

__global__ void __launch_bounds__(512, 1) pair128_chol_kernel(TileCholArgs) {   // tasks: two ints per task, batch_of: two per task
  int tix;
  {
    GPG_KERNARGS(TileCholArgs, ap);
    tix = next_ticket(ap->ticket, &g_next_ticket, 0);
  }
  for (;;) {
    GPG_KERNARGS(TileCholArgs, ap);
    if (tix >= ap->ntask) return;
    if (!pair128_chol_task(tix, ap->A, ap->ld, ap->Mt, ap->tasks, ap->flags, ap->pieces, ap->abort_word, ap->ticket, ap->dinv, ap->info,
                           ap->N, ap->batch_of, ap->a_stride, ap->d_stride, ap->f_stride, (unsigned long long)ap))
      return;
    tix = __builtin_amdgcn_readfirstlane(g_next_ticket);
  }
}


// ------------------------------------------------------------------------------------------------
// tile128_trinv_kernel: W = L^-T (upper triangular, Npad x Npad, leading dimension ldw) from the finished factor L in A,
// as ONE dataflow launch over 128 x 128 tiles -- the first N^3/3 sweep of the explicit inverse that the adjoint
// likelihood gradient contracts with (reference: adj_ln_detK = cho_solve(chofac, eye(N)), CalcLkd.py:174,234).
// W starts as the identity.  Workgroup task (j, i), j <= i (row tile j, column tile i of W; list order: tile column by
// tile column, longest accumulation first):
//     acc  = W_ji - sum_{k=j}^{i-1} W_jk L_ik^T    the MFMA loop of the factorisation (A operand: W row tile j, B operand:
//                                                  L row tile i, both read in fragment layout straight from memory),
//                                                  consumed in runs as the flags of W's row tile j come up
//     W_ji = acc L_ii^-T                           the 128-row substitution of the factorisation's off-diagonal tiles
//                                                  against the (final) diagonal tile of L
//     publish flag(j, i)
// Persistent workgroups and ticket order as in tile_chol_kernel: (j, i) waits only for (j, k), k < i -- earlier tile
// columns, i.e. smaller tickets.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool
tile128_trinv_task(int tix, const double* __restrict__ A, int ld, const double* __restrict__ dinv, double* W, int ldw, int Mt,
                   const int* __restrict__ tasks, int* flags, int* ones, int* abort_word, int* ticket, int* info,
                   const int* __restrict__ batch_of, size_t a_stride, size_t w_stride, int d_stride, int f_stride, int* lflags, int lf_mt) {
  __shared__ int sh_kr;
  __shared__ int sh_ok;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int task = tasks[tix];
  const int tj = task & 0xffff, ti = task >> 16;          // row tile j of W, column tile i
  // Overlapped with a 64-tile factorisation (gpg_overlap_inverse_*): column tile ti of W needs the 128 rows 128 ti .. of L up to the
  // diagonal; they are final once the factorisation's diagonal tile (2 ti + 1, 2 ti + 1) is (its task consumed the rest of its row, and the
  // tile row above went the same way before it).
  // (lf_mt == 0: the factorisation runs on the same 128 x 128 tiling: its diagonal tile (ti, ti).)  The abort word is looked at FIRST, on
  // every task: the host raises it when the factorisation it overlaps turns out to have failed (gpg_overlap_inverse_cancel), and the launch
  // then drains within one task instead of inverting a broken factor.
  if (lflags) {
    int* const fl = lf_mt > 0 ? lflags + (size_t)(2 * ti + 1) * lf_mt + (2 * ti + 1) : lflags + (size_t)ti * Mt + ti;
    if (tid == 0) sh_ok = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
    __syncthreads();
    const int go = sh_ok;
    __syncthreads();
    if (!go || !wg_wait_flag(fl, abort_word, info, &sh_ok)) return false;
  }
  if (batch_of) {   // batched launch: the factors of several matrices (restart rows) are inverted by one launch
    const int b = batch_of[tix];
    A += (size_t)b * a_stride;
    dinv += (size_t)b * d_stride;
    W += (size_t)b * w_stride;
    flags += (size_t)b * f_stride;
    info += b;
  }
  const size_t r0 = 128 * (size_t)tj, ci = 128 * (size_t)ti;
  int* const frow = flags + (size_t)tj * Mt;

  d4 acc[4][4];
  double* Cw = W + r0 + wm * 64 + 2 * l15 + (ci + wn * 64 + 2 * l4) * (size_t)ldw;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double2 v = *reinterpret_cast<const double2*>(Cw + 32 * g + (size_t)(32 * (ni >> 1) + 8 * r + (ni & 1)) * ldw);
        acc[ni][2 * g][r] = v.x;
        acc[ni][2 * g + 1][r] = v.y;
      }
  direct_tile_negate(acc);                                   // the loop adds the products to -W_ji (direct_tile_gemm_acc); sign restored below
  const unsigned lane_off_w = (unsigned)(2 * l15 + l4 * ldw) * 8u, lane_off_l = (unsigned)(2 * l15 + l4 * ld) * 8u;
  int kdone = tj;
  while (kdone < ti) {
    if (w == 0) {                                           // wave 0 scans the flag row, 64 tile columns per pass
      int* const frows[1] = {frow};
      int kr;
      const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        kr = wave_scan_flags<1>(frows, kdone, ti);
        if (kr > kdone) break;
        if (__builtin_amdgcn_s_memrealtime() - t_wait > GPG_TILE_WAIT_TICKS ||
            __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          if (lane == 0) {
            __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicMax(info, GPG_INFO_INTERNAL);
          }
          kr = -1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      if (lane == 0) sh_kr = kr;
    }
    __syncthreads();
    const int kr = __builtin_amdgcn_readfirstlane(sh_kr);
    if (kr < 0) return false;
    GPG_ACQUIRE();
    const size_t ck = 128 * (size_t)kdone;
    direct_tile_gemm_acc<GPG_MFMA_PF>(acc, W + r0 + wm * 64 + ck * (size_t)ldw, lane_off_w, ldw,
                            A + ci + wn * 64 + ck * (size_t)ld, lane_off_l, ld, 32 * (kr - kdone));
    __syncthreads();   // sh_kr may be rewritten
    kdone = kr;
  }
  direct_tile_negate(acc);
#ifndef GPG_TRINV_OLDFIN
  // (r03) the factorisation's one-piece finalisation: column block 0 handed over through LDS, column block 1 through memory; the diagonal
  // tile of L is final, so there is nothing to wait for
  {
    double* Cs = Cw;                                          // (store addresses formed here, see tile128_chol_task)
    asm volatile("" : "+v"(Cs));
    int l15s = l15, l4s = l4;
    asm volatile("" : "+v"(l15s), "+v"(l4s));
    if (wn == 0) fin128_handoff_block0<2>(acc, wm, l15s, l4s);
    else {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double2 v;
            v.x = acc[ni][2 * g][r];
            v.y = acc[ni][2 * g + 1][r];
            *reinterpret_cast<double2*>(Cs + 32 * g + (size_t)(32 * (ni >> 1) + 8 * r + (ni & 1)) * ldw) = v;
          }
    }
  }
  __syncthreads();
  GPG_PRIO(2);
#ifdef GPG_STAMP
  if (tid == 0) t128_fo = nullptr;
#endif
  fin128_onepiece<2>(W + r0 + ci * (size_t)ldw, ldw, A + ci + ci * (size_t)ld, ld, dinv + ci);
#else
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double2 v;
        v.x = acc[ni][2 * g][r];
        v.y = acc[ni][2 * g + 1][r];
        *reinterpret_cast<double2*>(Cw + 32 * g + (size_t)(32 * (ni >> 1) + 8 * r + (ni & 1)) * ldw) = v;
      }
  __syncthreads();
  GPG_PRIO(2);
#ifdef GPG_STAMP
  if (tid == 0) t128_fo = nullptr;
#endif
  // the diagonal tile of L is final: its piece flags are a constant array of ones
  if (!tile_solve_rows128(A + ci + ci * (size_t)ld, ld, dinv + ci, W + r0 + ci * (size_t)ldw, ldw, t128_U, t128_Ls, t128_sdinv, ones,
                          ones + 4, ones, abort_word, info, &sh_ok))
    return false;
#endif
  GPG_PUBLISH_AND_NEXT(ticket, frow + ti)
  return true;
}

struct TrinvArgs {
  const double* A; int ld; const double* dinv; double* W; int ldw, Mt; const int* tasks; int ntask; int* flags; int* ones;
  int* abort_word; int* ticket; int* info; const int* batch_of; size_t a_stride, w_stride; int d_stride, f_stride;
  int* lflags; int lf_mt;   // non-null: the (64-tile, lf_mt tile columns) factorisation of this ONE matrix may still be running
};

__global__ void __launch_bounds__(256, 2) tile128_trinv_kernel(TrinvArgs) {
  int tix;
  {
    GPG_KERNARGS(TrinvArgs, ap);
    tix = next_ticket(ap->ticket, &g_next_ticket, 0);
  }
  for (;;) {
    GPG_KERNARGS(TrinvArgs, ap);
    if (tix >= ap->ntask) return;
    if (!tile128_trinv_task(tix, ap->A, ap->ld, ap->dinv, ap->W, ap->ldw, ap->Mt, ap->tasks, ap->flags, ap->ones, ap->abort_word,
                            ap->ticket, ap->info, ap->batch_of, ap->a_stride, ap->w_stride, ap->d_stride, ap->f_stride, ap->lflags, ap->lf_mt))
      return;
    tix = __builtin_amdgcn_readfirstlane(g_next_ticket);
  }
}

// Frontier of a contraction over tile columns k of W that may still be in the making (W = L^-T running on the other stream): thread 0
// advances kr from kdone while BOTH flag rows show column kr finished, waits (bounded) until at least one column is; result through sh.
// Returns the new frontier, or -1 (timed out / aborted).  Whole workgroup; ends with an acquire.
__device__ __forceinline__ int wg_wait_two_rows(int* fa, int* fb, int kdone, int kend, int* abort_word, int* info, int* sh) {
  if (threadIdx.x == 0) {
    int kr = kdone;
    const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
    for (;;) {
      while (kr < kend && __hip_atomic_load(fa + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 &&
             __hip_atomic_load(fb + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
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
    *sh = kr;
  }
  __syncthreads();
  const int kr = *sh;
  __syncthreads();
  if (kr >= 0) GPG_ACQUIRE();
  return kr;
}

// ------------------------------------------------------------------------------------------------
// tile128_wwt_kernel: M = - W W^T, lower triangle (second N^3/3 sweep of the explicit inverse: -(L L^T)^-1 = -L^-T L^-1),
// W = L^-T upper triangular: tile (a, b), a >= b:  M_ab = - sum_{k >= a} W_ak W_bk^T -- independent tiles, no flags;
// persistent workgroups take them from the ticket counter, longest contraction (smallest a) first.  The same
// direct-fragment MFMA loop, accumulators start at zero.
// ------------------------------------------------------------------------------------------------
// Generalised for the Frobenius-norm condition number (gpg_cond_fro): M = - Sa Sb^T with two different full (symmetric) operands
// and the contraction over ALL tile columns when full_k != 0 (Sb0 == nullptr: Sb = Sa, the W W^T case).
__global__ void __launch_bounds__(256, 2)
tile128_wwt_kernel(const double* __restrict__ W0, int ldw, double* __restrict__ M0, int ldm, int Mt, const int* __restrict__ tasks,
                   int ntask, int* ticket, const int* __restrict__ batch_of, size_t w_stride, const double* __restrict__ Sb0, int full_k,
                   int* wflags /* non-null (one matrix, W W^T): tile flags [row * Mt + column] of a W still in the making */, int* abort_word,
                   int* info) {
  __shared__ int sh_tix;
  __shared__ int sh_kr;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w & 1, wn = w >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  for (int round = 0;; ++round) {
    const int tix = next_ticket(ticket, &sh_tix, round);
    if (tix >= ntask) return;
    const int task = tasks[tix];
    const int ta = task & 0xffff, tb = task >> 16;        // a >= b
    const size_t boff = batch_of ? (size_t)batch_of[tix] * w_stride : 0;
    const double* W = W0 + boff;
    const double* Wb = Sb0 ? Sb0 + boff : W;
    double* M = M0 + boff;
    const size_t r0 = 128 * (size_t)ta, c0 = 128 * (size_t)tb, ck = full_k ? 0 : 128 * (size_t)ta;
    d4 acc[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ni][mi][r] = 0.0;
    const unsigned lane_off = (unsigned)(2 * l15 + l4 * ldw) * 8u;
    if (!wflags) {
      direct_tile_gemm_acc<GPG_MFMA_PF>(acc, W + r0 + wm * 64 + ck * (size_t)ldw, lane_off, ldw,
                              Wb + c0 + wn * 64 + ck * (size_t)ldw, lane_off, ldw, 32 * (full_k ? Mt : Mt - ta));
    } else {   // the same sum in the same order, in runs of the tile columns of W that are finished
      int kdone = ta;
      while (kdone < Mt) {
        const int kr = wg_wait_two_rows(wflags + (size_t)ta * Mt, wflags + (size_t)tb * Mt, kdone, Mt, abort_word, info, &sh_kr);
        if (kr < 0) return;
        const size_t cc = 128 * (size_t)kdone;
        direct_tile_gemm_acc<GPG_MFMA_PF>(acc, W + r0 + wm * 64 + cc * (size_t)ldw, lane_off, ldw,
                                W + c0 + wn * 64 + cc * (size_t)ldw, lane_off, ldw, 32 * (kr - kdone));
        kdone = kr;
      }
    }
    direct_tile_negate(acc);                               // M = - Sa Sb^T: the loop accumulated + Sa Sb^T
    double* Cw = M + r0 + wm * 64 + 2 * l15 + (c0 + wn * 64 + 2 * l4) * (size_t)ldm;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double2 v;
          v.x = acc[ni][2 * g][r];
          v.y = acc[ni][2 * g + 1][r];
          *reinterpret_cast<double2*>(Cw + 32 * g + (size_t)(32 * (ni >> 1) + 8 * r + (ni & 1)) * ldm) = v;
        }
  }
}

// ------------------------------------------------------------------------------------------------
// rows_fwd_kernel: W <- W L^-T for right-hand-side rows held OUTSIDE the matrix (rows layout, leading dimension ldw,
// 64-row tiles) against the finished factor in A, as ONE dataflow launch: workgroup (rt, j) owns the 64 x 64 block of
// row tile rt and column block j, accumulates  W_j - sum_{k<j} X_k L_jk^T  on MFMA as the X_k of its own row tile are
// published (the L tiles are final), substitutes against L_jj and raises flag(rt, j).  The blocked version
// (gpg_forward_rows: panel solve + GEMM launch per 512 columns) is bound by ~70 dependent launches of 100+ us each;
// here the chain per 64 columns is one 64-deep MFMA block, one substitution and one flag hop.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool
rows_fwd_task(int tix, const double* __restrict__ A, int ld, const double* __restrict__ dinv, double* W, int ldw, int Mt, int nrt,
              int valid, int* flags, int* abort_word, int* info) {
  constexpr int KB = 16, SA = 80, BUF = KB * SA;
  __shared__ __attribute__((aligned(16))) double U[4 * BUF];
  __shared__ __attribute__((aligned(16))) double Ls[64][4][18];
  __shared__ double sdinv[64];
  __shared__ int sh_kr;
  double* const sA = U;
  double* const sB = U + 2 * BUF;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int tj = tix / nrt, rt = tix - tj * nrt;                    // column-block-major task order
  const size_t r0 = 64 * (size_t)rt, cj = 64 * (size_t)tj;
  const int q = tid & 3;
  const int sp = tid & 31, sk = tid >> 5;
  int* const frow = flags + (size_t)rt * Mt;

  d4 acc[4];
  {
    const double* Cw = W + r0 + 16 * w + l15 + (cj + l4) * (size_t)ldw;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ni][r] = Cw[(size_t)(ni * 16 + 4 * r) * ldw];
  }
  {   // L_jj is final: its image is fetched before the first wait
    const double* Ljj = A + cj + cj * (size_t)ld;
    for (int t = tid; t < 64 * 64; t += 256) {
      const int jj = t >> 6, k = t & 63;
      Ls[jj][k & 3][k >> 2] = Ljj[k + (size_t)jj * ld];
    }
    if (tid < 64) sdinv[tid] = dinv[cj + tid];
  }
  int kdone = 0;
  while (kdone < tj) {
    if (tid == 0) {
      int kr = kdone;
      const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        while (kr < tj && __hip_atomic_load(frow + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ++kr;
        if (kr > kdone) break;
        if (__builtin_amdgcn_s_memrealtime() - t_wait > GPG_TILE_WAIT_TICKS ||
            __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicMax(info, GPG_INFO_INTERNAL);
          kr = -1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      sh_kr = kr;
    }
    __syncthreads();
    const int kr = __builtin_amdgcn_readfirstlane(sh_kr);
    if (kr < 0) return false;
    GPG_ACQUIRE();
    const size_t ck = 64 * (size_t)kdone;
    wave_tile_gemm(acc, W + r0 + 2 * sp + (ck + sk) * (size_t)ldw, ldw, A + cj + 2 * sp + (ck + sk) * (size_t)ld, ld,
                   4 * (kr - kdone), sA, sB, w, l15, l4, sp, sk);
    __syncthreads();
    kdone = kr;
  }
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
  if (64 * rt + 16 * w < valid) {   // rows >= valid are zero (and stay zero): their waves skip the substitution
    GPG_QUAD_SUBST(x, Ls, sdinv, q)
  }
  double* Xr = W + r0 + (tid >> 2) + (cj + q) * (size_t)ldw;
#pragma unroll
  for (int m = 0; m < 16; ++m) GPG_ST(&Xr[(size_t)(4 * m) * ldw], x[m]);
  GPG_RELEASE();
  __syncthreads();
  if (tid == 0) GPG_FLAG_UP(frow + tj);
  return true;
}

// The same product on 64 x 64 tiles (wave_tile_gemm, the MFMA loop of tile_chol_kernel): for ONE small matrix the 128-tile list has fewer
// tasks than the device has workgroup slots (210 at 2560 columns) and its longest contraction is the whole launch; four times as many
// tiles of a quarter of the work fill the slots and halve the longest task.  tasks[t] = a | b << 16, a >= b, longest first.
__global__ void __launch_bounds__(256, 2)
tile64_wwt_kernel(const double* __restrict__ W, int ldw, double* __restrict__ M, int ldm, int Mt, const int* __restrict__ tasks, int ntask,
                  int* ticket, int* wflags /* non-null: tile flags [row * Mt + column] of a W still in the making */, int* abort_word, int* info) {
  constexpr int KB = 16, SA = 80, BUF = KB * SA;
  __shared__ __attribute__((aligned(16))) double U[4 * BUF];
  __shared__ int sh_tix;
  __shared__ int sh_kr;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4, sp = tid & 31, sk = tid >> 5;
  for (int round = 0;; ++round) {
    const int tix = next_ticket(ticket, &sh_tix, round);
    if (tix >= ntask) return;
    const int task = tasks[tix];
    const int ta = task & 0xffff, tb = task >> 16;        // a >= b
    const size_t r0 = 64 * (size_t)ta, c0 = 64 * (size_t)tb, ck = 64 * (size_t)ta;   // W is upper triangular: columns >= 64 a
    d4 acc[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ni][r] = 0.0;
    if (!wflags) {
      wave_tile_gemm(acc, W + r0 + 2 * sp + (ck + sk) * (size_t)ldw, ldw, W + c0 + 2 * sp + (ck + sk) * (size_t)ldw, ldw, 4 * (Mt - ta), U,
                     U + 2 * BUF, w, l15, l4, sp, sk);
    } else {   // the same sum in the same order, in runs of the tile columns of W that are finished
      int kdone = ta;
      while (kdone < Mt) {
        const int kr = wg_wait_two_rows(wflags + (size_t)ta * Mt, wflags + (size_t)tb * Mt, kdone, Mt, abort_word, info, &sh_kr);
        if (kr < 0) return;
        const size_t cc = 64 * (size_t)kdone;
        wave_tile_gemm(acc, W + r0 + 2 * sp + (cc + sk) * (size_t)ldw, ldw, W + c0 + 2 * sp + (cc + sk) * (size_t)ldw, ldw, 4 * (kr - kdone), U,
                       U + 2 * BUF, w, l15, l4, sp, sk);
        __syncthreads();                                  // staging buffers free for the next run
        kdone = kr;
      }
    }
    double* Cw = M + r0 + 16 * w + l15 + (c0 + l4) * (size_t)ldm;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cw[(size_t)(ni * 16 + 4 * r) * ldm] = acc[ni][r];
    __syncthreads();                                      // staging buffers and sh_tix free for the next round
  }
}
__global__ void __launch_bounds__(256, 2)
rows_fwd_kernel(const double* __restrict__ A, int ld, const double* __restrict__ dinv, double* W, int ldw, int Mt, int nrt,
                int valid, int* flags, int* abort_word, int* ticket, int* info) {
  __shared__ int sh_tix;
  const int ntask = Mt * nrt;
  for (int round = 0;; ++round) {
    const int tix = next_ticket(ticket, &sh_tix, round);
    if (tix >= ntask) return;
    if (!rows_fwd_task(tix, A, ld, dinv, W, ldw, Mt, nrt, valid, flags, abort_word, info)) return;
  }
}

// ------------------------------------------------------------------------------------------------
// tile64_trinv_kernel: W = L^-T with 64 x 64 tiles, for small matrices (the 128-tile version's dependency chain -- one
// 128-row substitution per tile column -- is what a 2560-column inverse spends its time on).  Task (rt, tj), rt <= tj (row
// tile rt, column tile tj of the upper triangular W, which starts as the identity; list order: tile column by tile
// column): acc = W_(rt,tj) - sum_{k=rt}^{tj-1} W_(rt,k) L_(tj,k)^T on MFMA as the tiles of W's row rt are published, then the
// quad-row substitution against L_(tj,tj): the body of rows_fwd_kernel restricted to the non-zero tiles, batched.
// ------------------------------------------------------------------------------------------------
struct Trinv64Args {
  const double* A; int ld; const double* dinv; double* W; int ldw, Mt; const int* tasks; int ntask; int* flags; int* abort_word;
  int* ticket; int* info; const int* batch_of; size_t a_stride, w_stride; int d_stride, f_stride;
  int* lflags;   // non-null: the factorisation may still be running; its tile flags [ti * Mt + tj] (64 x 64 tiles; matrix b at + b lf_stride)
  int lf_stride;
};

__device__ __forceinline__ bool
tile64_trinv_task(int tix, const double* __restrict__ A, int ld, const double* __restrict__ dinv, double* W, int ldw, int Mt,
                  const int* __restrict__ tasks, int* flags, int* abort_word, int* ticket, int* info,
                  const int* __restrict__ batch_of, size_t a_stride, size_t w_stride, int d_stride, int f_stride, int* lflags, int lf_stride) {
  constexpr int KB = 16, SA = 80, BUF = KB * SA;
  __shared__ __attribute__((aligned(16))) double U[4 * BUF];
  __shared__ __attribute__((aligned(16))) double Ls[64][4][18];
  __shared__ double sdinv[64];
  __shared__ int sh_kr;
  double* const sA = U;
  double* const sB = U + 2 * BUF;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int task = tasks[tix];
  const int rt = task & 0xffff, tj = task >> 16;
  if (batch_of) {
    const int b = batch_of[tix];
    A += (size_t)b * a_stride;
    dinv += (size_t)b * d_stride;
    W += (size_t)b * w_stride;
    flags += (size_t)b * f_stride;
    info += b;
    if (lflags) lflags += (size_t)b * lf_stride;
  }
  const size_t r0 = 64 * (size_t)rt, cj = 64 * (size_t)tj;
  const int q = tid & 3;
  const int sp = tid & 31, sk = tid >> 5;
  int* const frow = flags + (size_t)rt * Mt;

  d4 acc[4];
  {
    const double* Cw = W + r0 + 16 * w + l15 + (cj + l4) * (size_t)ldw;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ni][r] = Cw[(size_t)(ni * 16 + 4 * r) * ldw];
  }
  // Overlapped with the factorisation (lflags): column tile tj of W needs row tile tj of L, L_(tj,k) for k <= tj -- all of it is final once
  // the diagonal tile (tj, tj) is (its task consumed the others), so ONE wait on that flag covers the whole task.
  if (lflags && !wg_wait_flag(lflags + (size_t)tj * Mt + tj, abort_word, info, &sh_kr)) return false;
  {   // L_jj is final: its image is fetched before the first wait
    const double* Ljj = A + cj + cj * (size_t)ld;
    for (int t = tid; t < 64 * 64; t += 256) {
      const int jj = t >> 6, k = t & 63;
      Ls[jj][k & 3][k >> 2] = Ljj[k + (size_t)jj * ld];
    }
    if (tid < 64) sdinv[tid] = dinv[cj + tid];
  }
  int kdone = rt;
  while (kdone < tj) {
    if (tid == 0) {
      int kr = kdone;
      const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        while (kr < tj && __hip_atomic_load(frow + kr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ++kr;
        if (kr > kdone) break;
        if (__builtin_amdgcn_s_memrealtime() - t_wait > GPG_TILE_WAIT_TICKS ||
            __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicMax(info, GPG_INFO_INTERNAL);
          kr = -1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      sh_kr = kr;
    }
    __syncthreads();
    const int kr = __builtin_amdgcn_readfirstlane(sh_kr);
    if (kr < 0) return false;
    GPG_ACQUIRE();
    const size_t ck = 64 * (size_t)kdone;
    wave_tile_gemm(acc, W + r0 + 2 * sp + (ck + sk) * (size_t)ldw, ldw, A + cj + 2 * sp + (ck + sk) * (size_t)ld, ld,
                   4 * (kr - kdone), sA, sB, w, l15, l4, sp, sk);
    __syncthreads();
    kdone = kr;
  }
  GPG_PRIO(2);
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
  GPG_QUAD_SUBST(x, Ls, sdinv, q)
  double* Xr = W + r0 + (tid >> 2) + (cj + q) * (size_t)ldw;
#pragma unroll
  for (int m = 0; m < 16; ++m) GPG_ST(&Xr[(size_t)(4 * m) * ldw], x[m]);
  GPG_PUBLISH_AND_NEXT(ticket, frow + tj)
  return true;
}

__global__ void __launch_bounds__(256, 2) tile64_trinv_kernel(Trinv64Args) {
  int tix;
  {
    GPG_KERNARGS(Trinv64Args, ap);
    tix = next_ticket(ap->ticket, &g_next_ticket, 0);
  }
  for (;;) {
    GPG_KERNARGS(Trinv64Args, ap);
    if (tix >= ap->ntask) return;
    if (!tile64_trinv_task(tix, ap->A, ap->ld, ap->dinv, ap->W, ap->ldw, ap->Mt, ap->tasks, ap->flags, ap->abort_word, ap->ticket,
                           ap->info, ap->batch_of, ap->a_stride, ap->w_stride, ap->d_stride, ap->f_stride, ap->lflags, ap->lf_stride))
      return;
    tix = __builtin_amdgcn_readfirstlane(g_next_ticket);
  }
}

// ------------------------------------------------------------------------------------------------
// rows_bwd_kernel: Z <- Z L^-1 (every row solved against L^T) for right-hand-side rows in the rows layout, ONE
// dataflow launch, column blocks from the last to the first: workgroup (rt, j) accumulates
// Z_j - sum_{k>j} Z_k L_kj on MFMA (L_kj used untransposed: k-major staging) as the Z_k of its row tile are
// published, then runs the reverse substitution against L_jj.  Replaces 2 x Npad/64 dependent launches
// (gpg_backward_rows).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool
rows_bwd_task(int tix, const double* __restrict__ A, int ld, const double* __restrict__ dinv, double* Z, int ldz, int Mt, int nrt,
              int valid, int* flags, int* abort_word, int* info) {
  constexpr int KB = 16, SA = 80, BUF = KB * SA;
  __shared__ __attribute__((aligned(16))) double U[4 * BUF];
  __shared__ __attribute__((aligned(16))) double Ls[64][4][18];   // transposed image: Ls[j][q][m] = L_jj[j][4m + q]
  __shared__ double sdinv[64];
  __shared__ int sh_kr;
  double* const sA = U;
  double* const sB = U + 2 * BUF;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int jr = tix / nrt, rt = tix - jr * nrt;
  const int tj = Mt - 1 - jr;                                      // last column block first
  const size_t r0 = 64 * (size_t)rt, cj = 64 * (size_t)tj;
  const int q = tid & 3;
  const int sp = tid & 31, sk = tid >> 5;
  int* const frow = flags + (size_t)rt * Mt;

  d4 acc[4];
  {
    const double* Cw = Z + r0 + 16 * w + l15 + (cj + l4) * (size_t)ldz;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ni][r] = Cw[(size_t)(ni * 16 + 4 * r) * ldz];
  }
  {
    const double* Ljj = A + cj + cj * (size_t)ld;
    for (int t = tid; t < 64 * 64; t += 256) {
      const int cc = t >> 6, rr = t & 63;                          // element L_jj[rr][cc], read down the column
      Ls[rr][cc & 3][cc >> 2] = Ljj[rr + (size_t)cc * ld];
    }
    if (tid < 64) sdinv[tid] = dinv[cj + tid];
  }
  int khi = Mt;                                                    // blocks [khi, Mt) are applied
  while (khi > tj + 1) {
    if (tid == 0) {
      int kr = khi;
      const unsigned long long t_wait = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        while (kr > tj + 1 && __hip_atomic_load(frow + kr - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) --kr;
        if (kr < khi) break;
        if (__builtin_amdgcn_s_memrealtime() - t_wait > GPG_TILE_WAIT_TICKS ||
            __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicMax(info, GPG_INFO_INTERNAL);
          kr = -1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      sh_kr = kr;
    }
    __syncthreads();
    const int kr = __builtin_amdgcn_readfirstlane(sh_kr);
    if (kr < 0) return false;
    GPG_ACQUIRE();
    const size_t ck = 64 * (size_t)kr;                             // contraction rows [64 kr, 64 khi)
    wave_tile_gemm_nn(acc, Z + r0 + 2 * sp + (ck + sk) * (size_t)ldz, ldz, A + ck + cj * (size_t)ld, ld, 4 * (khi - kr), sA, sB, w,
                      l15, l4, sp, sk);
    __syncthreads();
    khi = kr;
  }
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
  if (64 * rt + 16 * w < valid) {
    GPG_QUAD_SUBST_REV(x, Ls, sdinv, q)
  }
  double* Xr = Z + r0 + (tid >> 2) + (cj + q) * (size_t)ldz;
#pragma unroll
  for (int m = 0; m < 16; ++m) GPG_ST(&Xr[(size_t)(4 * m) * ldz], x[m]);
  GPG_RELEASE();
  __syncthreads();
  if (tid == 0) GPG_FLAG_UP(frow + tj);
  return true;
}

__global__ void __launch_bounds__(256, 2)
rows_bwd_kernel(const double* __restrict__ A, int ld, const double* __restrict__ dinv, double* Z, int ldz, int Mt, int nrt,
                int valid, int* flags, int* abort_word, int* ticket, int* info) {
  __shared__ int sh_tix;
  const int ntask = Mt * nrt;
  for (int round = 0;; ++round) {
    const int tix = next_ticket(ticket, &sh_tix, round);
    if (tix >= ntask) return;
    if (!rows_bwd_task(tix, A, ld, dinv, Z, ldz, Mt, nrt, valid, flags, abort_word, info)) return;
  }
}

// ------------------------------------------------------------------------------------------------
// vec_solve_kernel<BWD, RR>: the same dataflow solves for at most RR (1 or 4) right-hand-side rows (posterior
// evaluation at one point, the alpha vector): no MFMA tile, no 64-row substitution.  Workgroup j owns column block
// j; lane <-> column c of the block, wave w accumulates the quarter m in [16w, 16w+16) of every finished block k
//     FWD: t[r][c] -= x_k[r][m] L[64j + c][64k + m]   (k < j)        BWD: t[r][c] -= z_k[r][m] L[64k + m][64j + c]  (k > j)
// with L read straight from memory (the block applied next is prefetched while the wave polls).  The solution is
// exchanged through a compact copy xc[r][Npad] that the host fills with a NaN sentinel: THE DATA IS THE FLAG -- a
// producer publishes x_k with agent-scope atomic stores, a consumer wave re-reads the 16 values it needs (one line)
// until none is the sentinel; no separate flag, no release/acquire fence on the chain.  The values are broadcast by
// v_readlane.  No barrier until the four partial sums meet in LDS; wave r (< R) then runs the 64-step substitution
// of row r in the scaled variable u = v / L_cc (two dependent instructions per step: v_readlane, v_fma) with the
// diagonal block held in registers.
// ------------------------------------------------------------------------------------------------
#define GPG_VEC_SENTINEL 0x7ff8dead7ff8deadull     // quiet NaN that no arithmetic produces; both halves equal (memsetD32)
template <int BWD, int RR>
__device__ __forceinline__ bool
vec_solve_task(int tix, const double* __restrict__ A, int ld, const double* __restrict__ dinv, double* W, int ldw, int Mt, int R,
               unsigned long long* xc, int Npad, int* abort_word, int* info) {
  __shared__ double part[4][RR][64];     // [wave][row][column] partial sums
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tj = BWD ? Mt - 1 - tix : tix;
  const size_t cj = 64 * (size_t)tj;
  const int ktot = BWD ? Mt - 1 - tj : tj;
  const double dv = dinv[cj + lane];
  // diagonal block, scaled by the lane's own reciprocal pivot: FWD lane c keeps row c of L_jj, BWD column c
  double Ls[64];
  double v0 = 0.0;
  if (w < R) {
#pragma unroll
    for (int m = 0; m < 64; ++m)
      Ls[m] = (BWD ? A[cj + m + (cj + lane) * (size_t)ld] : A[cj + lane + (cj + m) * (size_t)ld]) * dv;
    v0 = W[w + (cj + lane) * (size_t)ldw];
  }
  double t[RR];
#pragma unroll
  for (int r = 0; r < RR; ++r) t[r] = 0.0;
  double lcur[16];
  const bool carrier = lane < 16 * RR && (lane >> 4) < R;     // lane 16 r + mm carries x_k[r][16 w + mm]
  bool dead = false;
  for (int kk = 0; kk < ktot; ++kk) {      // FWD: k = kk, BWD: k = Mt-1-kk
    const size_t ck = 64 * (size_t)(BWD ? Mt - 1 - kk : kk);
#pragma unroll
    for (int mm = 0; mm < 16; ++mm) {
      const int m = 16 * w + mm;
      lcur[mm] = BWD ? A[ck + m + (cj + lane) * (size_t)ld] : A[cj + lane + (ck + m) * (size_t)ld];
    }
    const unsigned long long* src = xc + (size_t)(carrier ? lane >> 4 : 0) * Npad + ck + 16 * w + (lane & 15);
    unsigned long long bits = 0;
    unsigned long long t_wait = 0;
    for (;;) {
      bits = carrier ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      if (!__any(bits == GPG_VEC_SENTINEL)) break;
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t_wait == 0) t_wait = now;
      if (now - t_wait > GPG_TILE_WAIT_TICKS || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        if (lane == 0) {
          __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicMax(info, GPG_INFO_INTERNAL);
        }
        dead = true;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (dead) break;
    const double xv = __longlong_as_double((long long)bits);
#pragma unroll
    for (int r = 0; r < RR; ++r)
#pragma unroll
      for (int mm = 0; mm < 16; ++mm) t[r] -= readlane_d(xv, 16 * r + mm) * lcur[mm];
  }
#pragma unroll
  for (int r = 0; r < RR; ++r) part[w][r][lane] = t[r];
  __syncthreads();
  if (w < R && !dead) {
    const int r = w < RR ? w : 0;
    double u = (v0 + part[0][r][lane] + part[1][r][lane] + part[2][r][lane] + part[3][r][lane]) * dv;
    double x = 0.0;
    if (!BWD) {
#pragma unroll
      for (int m = 0; m < 64; ++m) {
        const double xm = readlane_d(u, m);
        u -= xm * Ls[m];                 // meaningful in lanes c > m; finished lanes are never read again
        x = (lane == m) ? xm : x;
      }
    } else {
#pragma unroll
      for (int m = 63; m >= 0; --m) {
        const double xm = readlane_d(u, m);
        u -= xm * Ls[m];
        x = (lane == m) ? xm : x;
      }
    }
    __hip_atomic_store(xc + (size_t)w * Npad + cj + lane, (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    W[w + (cj + lane) * (size_t)ldw] = x;
  }
  return !__syncthreads_or(dead ? 1 : 0);      // a wave that timed out ends the workgroup (the abort word drains the others)
}

struct VecSolveArgs {
  const double* A; int ld; const double* dinv; double* W; int ldw, Mt, R; unsigned long long* xc; int Npad; int* abort_word; int* ticket;
  int* info;
};

template <int BWD, int RR>
__device__ __noinline__ int vec_solve_task_call(int tix_v, unsigned long long kernarg_bits) {
  GPG_KERNARGS_FROM(VecSolveArgs, ap, kernarg_bits);
  const int tix = __builtin_amdgcn_readfirstlane(tix_v);
  return vec_solve_task<BWD, RR>(tix, ap->A, ap->ld, ap->dinv, ap->W, ap->ldw, ap->Mt, ap->R, ap->xc, ap->Npad, ap->abort_word, ap->info) ? 1 : 0;
}

template <int BWD, int RR>
__global__ void __launch_bounds__(256) vec_solve_kernel(VecSolveArgs) {
  __shared__ int sh_tix;
  for (int round = 0;; ++round) {
    GPG_KERNARGS(VecSolveArgs, ap);
    const int tix = next_ticket(ap->ticket, &sh_tix, round);
    if (tix >= ap->Mt) return;
    if (!__builtin_amdgcn_readfirstlane((vec_solve_task_call<BWD, RR>(tix, (unsigned long long)ap)))) return;
  }
}

// Ticket order of the factorisation's tasks (Mt tile columns, Rt >= Mt tile rows; B matrices interleaved).  Any order in which a
// task comes after the tasks it reads is valid (progress argument at tile_chol_kernel); two are offered:
//   order 0  tile column by tile column, rows top to bottom, the matrices of a batch interleaved column by column;
//   order 1  the same with the critical path pulled forward: in tile column j the tile (j+1, j) comes first and the NEXT diagonal
//            tile (j+1, j+1) right after it -- it only reads tile row j+1, which (j+1, j) completes -- then the rest of column j.
//            The diagonal tile's potrf / L21 / syrk / potrf chain (~100 us under contention) then runs while column j is still in its
//            MFMA loops, instead of after them with the whole column j+1 waiting for its pieces.
// fuse: the sub-diagonal tiles (j+1, j) of the matrix have no task of their own -- the diagonal task (j+1, j+1) of the 64-tile kernel
// computes and publishes them (tile_chol_task); it only reads earlier tile columns and the pieces of diagonal tile (j, j), all with
// smaller tickets.
template <typename F>
static void for_each_chol_task(int Mt, int Rt, int B, int order, F emit /* (b, i, j) */, bool fuse = false) {
  if (order == 0) {
    for (int j = 0; j < Mt; ++j)
      for (int b = 0; b < B; ++b)
        for (int i = j; i < Rt; ++i)
          if (!(fuse && i == j + 1 && i < Mt)) emit(b, i, j);
    return;
  }
  for (int b = 0; b < B; ++b) emit(b, 0, 0);
  for (int j = 0; j < Mt; ++j) {
    const bool next_diag = j + 1 < Mt;
    if (next_diag)
      for (int b = 0; b < B; ++b) { if (!fuse) emit(b, j + 1, j); emit(b, j + 1, j + 1); }
    for (int b = 0; b < B; ++b)
      for (int i = j + (next_diag ? 2 : 1); i < Rt; ++i) emit(b, i, j);
  }
}

// Ticket order of a launch (gpg_ctx::task_order: 0 column-major, 1 critical path first, -1 = measured choice).  With the round-3
// finalisation (a tile whose diagonal tile is complete on arrival takes the one-piece path) it pays to have the diagonal tiles done
// early: order 1 gains 3.7 % on 64 matrices of 2560 columns, 3 % on 16 of 4608, 2.5 % on the 64-tile batches, 1-3 % on one matrix
// of 9216 columns, and loses 1 % on one matrix of 18048 (tools/tile_probe, GPG_PROBE_ORDER, same box, twice).
static int chol_task_order(const gpg_ctx* c, int B) {
  if (c->task_order >= 0) return c->task_order & 1;
  return (B >= 2 || c->Npad < 16384) ? 1 : 0;
}

// Task list of the dataflow factorisation of one matrix, cached per shape and order.
const TileMap& get_tile_tasks(gpg_ctx* c, int Mt, int Rt, bool fuse) {
  const int order = chol_task_order(c, 1);
  const unsigned long long key = (1ull << 63) | ((unsigned long long)order << 60) | ((unsigned long long)(fuse ? 1 : 0) << 59) |
                                 ((unsigned long long)Mt << 20) | (unsigned)Rt;
  auto it = c->tilemaps.find(key);
  if (it != c->tilemaps.end()) return it->second;
  std::vector<int> list;
  for_each_chol_task(Mt, Rt, 1, order, [&](int, int i, int j) { list.push_back(i | (j << 16)); }, fuse);
  TileMap tm;
  tm.n = (int)list.size();
  tm.dev = nullptr;
  if (!gpg_dev_alloc(c, &tm.dev, sizeof(int) * list.size())) {
    static const TileMap none{nullptr, 0};       // nothing to launch; the API call reports the failure
    return none;
  }
  (void)hipMemcpy(tm.dev, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice);
  return c->tilemaps.emplace(key, tm).first->second;
}

// Grid of a persistent launch: as many workgroups as the device holds at once (occupancy x compute units), at most one
// per task.  More would only queue behind the resident ones, fewer would leave slots empty; neither affects
// correctness (ticket order, see tile_chol_kernel).
template <typename K>
static int persistent_grid(gpg_ctx* c, K kernel, long ntask, int block = 256) {
  const void* key = reinterpret_cast<const void*>(kernel);
  auto it = c->occupancy.find(key);
  if (it == c->occupancy.end()) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; }
    it = c->occupancy.emplace(key, nb).first;
  }
  long cap = (long)it->second * (c->num_cus > 0 ? c->num_cus : 256);
  if (c->grid_cap > 0 && c->grid_cap < cap) cap = c->grid_cap;                     // overlapped inverse: leaves slots to the launch it waits for
  if (c->max_workgroups > 0 && c->max_workgroups < cap) cap = c->max_workgroups;   // gpg_set_max_workgroups: tests of the progress argument
#if defined(GPG_NO_PERSIST) || defined(GPG_TICKET_ONESHOT)
  return (int)ntask;
#endif
  return (int)(ntask < cap ? ntask : cap);
}

// Completion flags of the dataflow launches (grown on demand); false: allocation failed, nothing may be launched.
// The buffer is a multiple of 64 bytes long and cleared in such multiples: the runtime's fill then is ONE kernel instead of an aligned
// body plus a tail.
static inline size_t flags_fill(size_t nflag) { return (nflag + 15) & ~(size_t)15; }
static bool ensure_tile_flags(gpg_ctx* c, size_t nflag) {
  nflag = flags_fill(nflag);
  if (c->tile_flags_cap >= nflag && c->tile_flags) return true;
  if (c->tile_flags) (void)hipFree(c->tile_flags);
  c->tile_flags = nullptr;
  c->tile_flags_cap = 0;
  if (!gpg_dev_alloc(c, &c->tile_flags, sizeof(int) * nflag)) return false;
  c->tile_flags_cap = nflag;
  return true;
}

// Factor A[c0:, c0:] (and carry the rows below the matrix) with the dataflow kernel, on c->stream.
static void launch_tile_chol(gpg_ctx* c, int c0) {
  const int Mt = (c->Npad - c0) / 64, Rt = (c->ld - c0) / 64;
  if (Mt <= 0) return;
  // fused diagonal tasks pay while the launch is chain-bound; from ~4000 columns on their doubled MFMA loops are the longer chain
  // (tools/tile_probe: -8 % at 640 / 1280 columns, -5 % at 2560, +5 % at 5120, +30 % at 9216)
  const bool fuse = c->fuse_subdiag != 0 && Mt <= c->fuse_subdiag_max_tiles;
  const TileMap& tm = get_tile_tasks(c, Mt, Rt, fuse);
  if (!tm.dev) return;
  const size_t nflag = (size_t)Mt * Rt + 1 + 4 * (size_t)Mt + 8;   // tile flags, abort word, four piece flags per diagonal tile, ticket words
  int* fl = c->chol_flags_override;                        // overlapped inverse: flags that stay valid after this launch (one-shot)
  c->chol_flags_override = nullptr;
  const bool keep = fl != nullptr && c0 == 0;
  if (!keep) {
    if (!ensure_tile_flags(c, nflag)) return;
    fl = c->tile_flags;
  }
  (void)hipMemsetAsync(fl, 0, sizeof(int) * flags_fill(nflag), c->stream);
  if (keep) (void)hipEventRecord(c->ev_flags, c->stream);  // from here on the second stream may poll them
  const double m = (double)(c->N - c0);                    // algorithmic flops: N^3 / 3 of the real matrix, not of the padded one
  gpg_prof_begin(c, GPG_PROF_GEMM_TRAIL, m > 0 ? m * m * m / 3.0 : 0.0);
  int* abort_word = fl + (size_t)Mt * Rt;
  hipLaunchKernelGGL(tile_chol_kernel, dim3(persistent_grid(c, tile_chol_kernel, tm.n)), dim3(256), 0, c->stream,
                     TileCholArgs{c->A, c->ld, c0, Mt, tm.dev, tm.n, fl, abort_word + 1, abort_word, fl + (nflag - 8),
                                  c->dinv, c->info, c->N, nullptr, 0, 0, 0, fuse ? 1 : 0});
  gpg_prof_end(c);
}

// B independent matrices of the context's shape (workspaces Abase + b a_stride, reciprocal pivots dinv_base +
// b d_stride, info[b]) factorised by ONE launch: the task lists of the matrices are interleaved tile column by tile
// column, so every matrix's dependency chain advances at the same time and the chip is filled by small problems
// whose single factorisation is latency-bound.  Task order per matrix is unchanged (progress argument holds).
static void launch_tile_chol_batch(gpg_ctx* c, int B, double* Abase, size_t a_stride, double* dinv_base, int d_stride,
                                   int* info_base) {
  const int Mt = c->Npad / 64, Rt = c->ld / 64;
  // batched launches are throughput-bound, not chain-bound: the fused task's doubled loop costs +27 % there (tile_probe, 64 x 1280 columns)
  const bool fuse = c->fuse_subdiag != 0 && B == 1 && Mt <= c->fuse_subdiag_max_tiles;
  const int order = chol_task_order(c, B);
  const unsigned long long key = (2ull << 62) | ((unsigned long long)order << 60) | ((unsigned long long)(fuse ? 1 : 0) << 59) |
                                 ((unsigned long long)B << 40) | ((unsigned long long)Mt << 20) | (unsigned)Rt;
  auto it = c->tilemaps.find(key);
  if (it == c->tilemaps.end()) {
    std::vector<int> list, bof;
    for_each_chol_task(Mt, Rt, B, order, [&](int b, int i, int j) { list.push_back(i | (j << 16)); bof.push_back(b); }, fuse);
    TileMap tm;
    tm.n = (int)list.size();
    if (!gpg_dev_alloc(c, &tm.dev, sizeof(int) * 2 * list.size())) return;
    (void)hipMemcpy(tm.dev, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(tm.dev + list.size(), bof.data(), sizeof(int) * bof.size(), hipMemcpyHostToDevice);
    it = c->tilemaps.emplace(key, tm).first;
  }
  const TileMap& tm = it->second;
  const size_t per = (size_t)Mt * Rt + 1 + 4 * (size_t)Mt, nflag = per * B + 8;   // + the ticket words of the launch
  int* fl = c->chol_flags_override;                        // overlapped inverse (gpg_overlap_inverse_begin): one-shot
  c->chol_flags_override = nullptr;
  const bool keep = fl != nullptr;
  if (!keep) {
    if (!ensure_tile_flags(c, nflag)) return;
    fl = c->tile_flags;
  }
  (void)hipMemsetAsync(fl, 0, sizeof(int) * flags_fill(nflag), c->stream);
  if (keep) (void)hipEventRecord(c->ev_flags, c->stream);
  const double m = (double)c->N;
  gpg_prof_begin(c, GPG_PROF_GEMM_TRAIL, B * m * m * m / 3.0);
  int* abort_word = fl + (size_t)Mt * Rt;                 // the abort word of matrix 0 serves the whole launch
  hipLaunchKernelGGL(tile_chol_kernel, dim3(persistent_grid(c, tile_chol_kernel, tm.n)), dim3(256), 0, c->stream,
                     TileCholArgs{Abase, c->ld, 0, Mt, tm.dev, tm.n, fl, abort_word + 1, abort_word, fl + (nflag - 8),
                                  dinv_base, info_base, c->N, tm.dev + tm.n, a_stride, d_stride, (int)per, fuse ? 1 : 0});
  gpg_prof_end(c);
}

// The whole matrix with the 128-tile dataflow kernel, on c->stream.
static void launch_tile128_chol_batch(gpg_ctx* c, int B, double* Abase, size_t a_stride, double* dinv_base, int d_stride, int* info_base);
static bool pair128_wanted(const gpg_ctx* c, int B);
static void launch_tile128_chol(gpg_ctx* c) {
  c->last_launch_paired = false;
  if (!c->chol_flags_override && pair128_wanted(c, 1)) {     // (the overlapped inverse reads this launch's flags: one tile per workgroup then)
    launch_tile128_chol_batch(c, 1, c->A, 0, c->dinv, 0, c->info);
    return;
  }
  const int Mt = c->Npad / 128, Rt = c->ld / 128;
  const TileMap& tm = get_tile_tasks(c, Mt, Rt, false);
  if (!tm.dev) return;
  const size_t nflag = (size_t)Mt * Rt + 1 + 9 * (size_t)Mt + 8;   // tile flags, abort word, 4 + 4 piece flags and the L21 flag per diagonal tile, ticket words
  int* fl = c->chol_flags_override;                        // overlapped inverse (gpg_overlap_inverse_begin): one-shot
  c->chol_flags_override = nullptr;
  const bool keep = fl != nullptr;
  if (!keep) {
    if (!ensure_tile_flags(c, nflag)) return;
    fl = c->tile_flags;
  }
  (void)hipMemsetAsync(fl, 0, sizeof(int) * flags_fill(nflag), c->stream);
  if (keep) (void)hipEventRecord(c->ev_flags, c->stream);
  const double m = (double)c->N;
  gpg_prof_begin(c, GPG_PROF_GEMM_TRAIL, m * m * m / 3.0);
  int* abort_word = fl + (size_t)Mt * Rt;
  hipLaunchKernelGGL(tile128_chol_kernel, dim3(persistent_grid(c, tile128_chol_kernel, tm.n)), dim3(256), 0, c->stream,
                     TileCholArgs{c->A, c->ld, 0, Mt, tm.dev, tm.n, fl, abort_word + 1, abort_word, fl + (nflag - 8),
                                  c->dinv, c->info, c->N, nullptr, 0, 0, 0});
  gpg_prof_end(c);
}

// Task list of pair128_chol_kernel for B >= 2 matrices: per tile column j first the diagonal tiles (paired over the matrices), then each
// matrix's tiles below the diagonal in pairs of consecutive rows, then what was left over (at most one ordinary tile per matrix, paired
// over the matrices) and the right-hand-side tile rows (light tasks: paired with each other).  A tile without a partner is listed
// twice (both teams do it).  Every task waits only for tasks of earlier tile columns and for the diagonal tiles of its own column.
static void build_pair_tasks(int Mt, int Rt, int B, std::vector<int>& list, std::vector<int>& bof, int order) {
  auto emit = [&](int ba, int ia, int bb, int ib, int j) {
    list.push_back(ia | (j << 16)); list.push_back(ib | (j << 16));
    bof.push_back(ba); bof.push_back(bb);
  };
  auto emit_diag = [&](int j) { for (int b = 0; b < B; b += 2) emit(b, j, b + 1 < B ? b + 1 : b, j, j); };
  // order 1 (critical path first, as for_each_chol_task): the first pair below the diagonal of tile column j and then the diagonal
  // tiles of column j + 1 go ahead of the rest of column j
  if (order == 1) emit_diag(0);
  for (int j = 0; j < Mt; ++j) {
    if (order != 1) emit_diag(j);
    std::vector<std::pair<int, int>> left;
    const bool ahead = order == 1 && j + 1 < Mt;
    for (int pass = ahead ? 0 : 1; pass < 2; ++pass) {          // pass 0: only the pair that holds row j + 1; pass 1: the others
      for (int b = 0; b < B; ++b) {
        int i = j + 1;
        for (; i + 1 < Mt; i += 2)
          if (!ahead || (pass == 0) == (i == j + 1)) emit(b, i, b, i + 1, j);
        if (i < Mt && (!ahead || (pass == 0) == (i == j + 1))) left.emplace_back(b, i);
      }
      if (pass == 0) {                                          // (a lone row j + 1 -- the last column pair -- pairs over the matrices right away)
        for (size_t k = 0; k < left.size(); k += 2) {
          const auto& a = left[k];
          const auto& bb = k + 1 < left.size() ? left[k + 1] : left[k];
          emit(a.first, a.second, bb.first, bb.second, j);
        }
        left.clear();
        emit_diag(j + 1);
      }
    }
    for (size_t k = 0; k < left.size(); k += 2) {
      const auto& a = left[k];
      const auto& bb = k + 1 < left.size() ? left[k + 1] : left[k];
      emit(a.first, a.second, bb.first, bb.second, j);
    }
    for (int i = Mt; i < Rt; ++i)                               // right-hand-side tile rows (one per matrix with R = 128)
      for (int b = 0; b < B; b += 2) emit(b, i, b + 1 < B ? b + 1 : b, i, j);
  }
}

static bool pair128_wanted(const gpg_ctx* c, int B) {
  if (c->pair_mode == 1) return true;
  if (c->pair_mode != 2) return false;
  // measured (tools/tile_probe, same box, twice; pair kernel without a barrier in its MFMA loop): ahead of tile128_chol_kernel at every
  // batched size -- 64 x 2560 columns +1.3 %, 16 x 4608 +1.7 %, 8 x 9216 +2.6 %, 8 x 12288 +2.7 %, 2 / 5 / 10 x 18048 +2.2 / +1.3 / +2.3 % --
  // and 1-2 % behind it for ONE matrix per launch (9216 ... 68096 columns)
  return B >= 2 || c->Npad >= c->pair_single_cols;
}

// The same for B matrices at once (see launch_tile_chol_batch): at the two ends of a factorisation the dependency
// chain leaves most of the chip idle, and a second matrix fills it.
static void launch_tile128_chol_batch(gpg_ctx* c, int B, double* Abase, size_t a_stride, double* dinv_base, int d_stride,
                                      int* info_base) {
  const int Mt = c->Npad / 128, Rt = c->ld / 128;
  // pair128_chol_kernel (512-thread workgroups, two tiles of a tile column each): a quarter less memory traffic than tile128_chol_kernel
  // (87 against 120 GB per cfg3 matrix) and, since its finalisation lost its spill reloads and its MFMA loop its barrier, 1-3 % faster
  // One very large matrix per launch (cfg5: 532 tile columns): the rows of a tile column still pair up; only its diagonal tile has no
  // partner and is given to both teams (532 of 142 000 tasks).
  const bool pair = pair128_wanted(c, B);
  const int order = chol_task_order(c, B);
  const unsigned long long key = (2ull << 62) | (1ull << 61) | ((unsigned long long)order << 60) | ((unsigned long long)(pair ? 1 : 0) << 59) |
                                 ((unsigned long long)B << 40) | ((unsigned long long)Mt << 20) | (unsigned)Rt;
  auto it = c->tilemaps.find(key);
  if (it == c->tilemaps.end()) {
    std::vector<int> list, bof;
    if (pair) build_pair_tasks(Mt, Rt, B, list, bof, order);
    else for_each_chol_task(Mt, Rt, B, order, [&](int b, int i, int j) { list.push_back(i | (j << 16)); bof.push_back(b); });
    TileMap tm;
    tm.n = (int)(pair ? list.size() / 2 : list.size());
    if (!gpg_dev_alloc(c, &tm.dev, sizeof(int) * 2 * list.size())) return;
    (void)hipMemcpy(tm.dev, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(tm.dev + list.size(), bof.data(), sizeof(int) * bof.size(), hipMemcpyHostToDevice);
    it = c->tilemaps.emplace(key, tm).first;
  }
  const TileMap& tm = it->second;
  const size_t nlist = pair ? 2 * (size_t)tm.n : (size_t)tm.n;
  const size_t per = (size_t)Mt * Rt + 1 + 9 * (size_t)Mt, nflag = per * B + 8;   // + the ticket words of the launch
  if (!ensure_tile_flags(c, nflag)) return;
  (void)hipMemsetAsync(c->tile_flags, 0, sizeof(int) * flags_fill(nflag), c->stream);
  const double m = (double)c->N;
  gpg_prof_begin(c, GPG_PROF_GEMM_TRAIL, B * m * m * m / 3.0);
  int* abort_word = c->tile_flags + (size_t)Mt * Rt;
  const TileCholArgs args{Abase, c->ld, 0, Mt, tm.dev, tm.n, c->tile_flags, abort_word + 1, abort_word, c->tile_flags + (nflag - 8),
                          dinv_base, info_base, c->N, tm.dev + nlist, a_stride, d_stride, (int)per};
  c->last_launch_paired = pair;
  if (pair)
    hipLaunchKernelGGL(pair128_chol_kernel, dim3(persistent_grid(c, pair128_chol_kernel, tm.n, 512)), dim3(512), 0, c->stream, args);
  else
    hipLaunchKernelGGL(tile128_chol_kernel, dim3(persistent_grid(c, tile128_chol_kernel, tm.n)), dim3(256), 0, c->stream, args);
  gpg_prof_end(c);
}

// Minv <- -(L L^T)^-1 (lower triangle, leading dimension Npad) from finished factors, through W = L^-T (Npad x Npad): two
// dataflow launches of N^3/3 flops each per matrix (tile128_trinv_kernel, tile128_wwt_kernel).  B matrices at once
// (factor b at Abase + b a_stride with reciprocal pivots dinv_base + b d_stride, W / Minv of matrix b at + b Npad^2;
// task lists interleaved tile column by tile column like the batched factorisation).  false: not applicable / no memory.
// phase 0: both launches on c->stream; 1: W = L^-T only (flag buffer fbuf, factorisation flags lflags: gpg_overlap_inverse_trinv);
// 2: -(W W^T) only (same fbuf); 3: -(W W^T) only, consuming W by its flags while phase 1 is still running (one matrix).
static bool launch_tile128_inverse_batch(gpg_ctx* c, int B, const double* Abase, size_t a_stride, const double* dinv_base, int d_stride,
                                         double* Wbase, double* Mbase, int* info_base, int phase = 0, int* fbuf = nullptr,
                                         int* lflags = nullptr, int lf_stride = 0) {
  const int Mt = c->Npad / 128, ldw = c->Npad;
  if (Mt < 1 || Mt > 0xffff || B < 1) return false;
  const size_t w_stride = (size_t)ldw * c->Npad;
  // small matrices: W = L^-T with 64 x 64 tiles (half the substitution chain per tile column); -(W W^T) stays on 128-tiles
  const bool small = c->inv_tile64_cols > 0 && c->Npad <= c->inv_tile64_cols;
  const int Mt64 = c->Npad / 64;
  const unsigned long long key = (3ull << 61) | ((unsigned long long)(small ? 1 : 0) << 60) | ((unsigned long long)B << 40) | (unsigned long long)Mt;
  auto it = c->tilemaps.find(key);
  if (it == c->tilemaps.end()) {
    std::vector<int> list, bof;
    const int Mw = small ? Mt64 : Mt;
    for (int i = 0; i < Mw; ++i)                    // W tiles (j, i): tile column by tile column, longest accumulation first
      for (int b = 0; b < B; ++b)
        for (int j = 0; j <= i; ++j) { list.push_back(j | (i << 16)); bof.push_back(b); }
    const size_t n1 = list.size();
    for (int a = 0; a < Mt; ++a)                    // M tiles (a, b), a >= b: longest contraction (smallest a) first
      for (int b = 0; b < B; ++b)
        for (int bb = 0; bb <= a; ++bb) { list.push_back(a | (bb << 16)); bof.push_back(b); }
    TileMap tm;
    tm.n = (int)n1;
    if (!gpg_dev_alloc(c, &tm.dev, sizeof(int) * 2 * list.size())) return false;       // [tasks W | tasks M | matrix of W task | matrix of M task]
    (void)hipMemcpy(tm.dev, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(tm.dev + list.size(), bof.data(), sizeof(int) * bof.size(), hipMemcpyHostToDevice);
    it = c->tilemaps.emplace(key, tm).first;
  }
  const TileMap& tm = it->second;
  const int n2 = B * (Mt * (Mt + 1) / 2);           // tiles of -(W W^T), always 128 x 128
  const int* tasks1 = tm.dev;
  const int* tasks2 = tm.dev + tm.n;
  const int* bof1 = tm.dev + (size_t)tm.n + n2;
  const int* bof2 = bof1 + tm.n;
  const size_t per = small ? (size_t)Mt64 * Mt64 : (size_t)Mt * Mt;
  const size_t nflag = per * B + 16;                // W tile flags per matrix | 9 ones | abort | ticket (trinv) | ticket (wwt)
  int* fl = fbuf;
  if (!fl) {
    if (!ensure_tile_flags(c, nflag)) return false;
    fl = c->tile_flags;
  }
  int* ones = fl + per * B;
  if (phase != 2 && phase != 3) {
    (void)hipMemsetAsync(fl, 0, sizeof(int) * flags_fill(nflag), c->stream);
    (void)hipMemsetD32Async((hipDeviceptr_t)ones, 1, 9, c->stream);
    for (int b = 0; b < B; ++b) gpg_launch_identity(c, Wbase + (size_t)b * w_stride, ldw);
    // overlapped inverse: from here on W's flags, abort word, tickets and identity are in place -- what -(W W^T) on the main stream,
    // launched while this stream is still busy (phase 3), must not start before
    if (phase == 1 && c->ev_winit) (void)hipEventRecord(c->ev_winit, c->stream);
    if (small)
      hipLaunchKernelGGL(tile64_trinv_kernel, dim3(persistent_grid(c, tile64_trinv_kernel, tm.n)), dim3(256), 0, c->stream,
                         Trinv64Args{Abase, c->ld, dinv_base, Wbase, ldw, Mt64, tasks1, tm.n, fl, ones + 9, ones + 10, info_base,
                                     B > 1 ? bof1 : nullptr, a_stride, w_stride, d_stride, (int)per, lflags, lf_stride});
    else
      hipLaunchKernelGGL(tile128_trinv_kernel, dim3(persistent_grid(c, tile128_trinv_kernel, tm.n)), dim3(256), 0, c->stream,
                         TrinvArgs{Abase, c->ld, dinv_base, Wbase, ldw, Mt, tasks1, tm.n, fl, ones, ones + 9, ones + 10, info_base,
                                   B > 1 ? bof1 : nullptr, a_stride, w_stride, d_stride, (int)per, B == 1 ? lflags : nullptr, lf_stride /* 64-tile columns of the factorisation, 0: 128-tile */});
    if (phase == 1) return true;
  }
  const bool wflags_live = phase == 3;   // -(W W^T) of ONE matrix while W = L^-T is still running on the other stream (same tiling, flags in fl)
  if (small && B == 1) {   // one small matrix: -(W W^T) on 64-tiles too (task list cached under its own key)
    const unsigned long long key64 = (3ull << 61) | (1ull << 59) | (unsigned long long)Mt64;
    auto i64 = c->tilemaps.find(key64);
    if (i64 == c->tilemaps.end()) {
      std::vector<int> l64;
      for (int a = 0; a < Mt64; ++a)
        for (int bb = 0; bb <= a; ++bb) l64.push_back(a | (bb << 16));
      TileMap t64;
      t64.n = (int)l64.size();
      if (!gpg_dev_alloc(c, &t64.dev, sizeof(int) * l64.size())) return false;
      (void)hipMemcpy(t64.dev, l64.data(), sizeof(int) * l64.size(), hipMemcpyHostToDevice);
      i64 = c->tilemaps.emplace(key64, t64).first;
    }
    hipLaunchKernelGGL(tile64_wwt_kernel, dim3(persistent_grid(c, tile64_wwt_kernel, i64->second.n)), dim3(256), 0, c->stream,
                       (const double*)Wbase, ldw, Mbase, ldw, Mt64, (const int*)i64->second.dev, i64->second.n, ones + 11,
                       wflags_live ? fl : (int*)nullptr, ones + 9, info_base);
    return true;
  }
  hipLaunchKernelGGL(tile128_wwt_kernel, dim3(persistent_grid(c, tile128_wwt_kernel, n2)), dim3(256), 0, c->stream,
                     (const double*)Wbase, ldw, Mbase, ldw, Mt, tasks2, n2, ones + 11, B > 1 ? bof2 : (const int*)nullptr, w_stride,
                     (const double*)nullptr, 0, (wflags_live && !small && B == 1) ? fl : (int*)nullptr, ones + 9, info_base);
  return true;
}

static bool launch_tile128_inverse(gpg_ctx* c, double* W, double* Minv) {
  return launch_tile128_inverse_batch(c, 1, c->A, 0, c->dinv, 0, W, Minv, c->info);
}

// ---- pieces of the Frobenius-norm condition number (gpg_cond_fro; reference GpHparaCon.py:209-236) --------------------
// partial[c] = sum over the rows r >= c (r, c < N) of the lower triangle of M of M[r][c]^2, off-diagonal entries twice
__global__ void __launch_bounds__(256) frob_lower_cols_kernel(const double* __restrict__ M, int ld, int N, double* __restrict__ partial) {
  __shared__ double red[4];
  const int c = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double* col = M + (size_t)c * ld;
  double s = 0.0;
  for (int r = c + threadIdx.x; r < N; r += 256) { const double v = col[r]; s += (r == c ? 1.0 : 2.0) * (v * v); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) red[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[c] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(256) sum_fixed_order_kernel(const double* __restrict__ v, int n, double* __restrict__ out) {
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) red[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (red[0] + red[1]) + (red[2] + red[3]);
}
// upper triangle <- transposed lower triangle (32 x 32 tiles through LDS, coalesced both ways)
__global__ void __launch_bounds__(256) symmetrize_kernel(double* __restrict__ M, int ld, int n) {
  __shared__ double t[32][33];
  const int bi = blockIdx.x, bj = blockIdx.y;            // tile (rows 32 bi.., columns 32 bj..), bi > bj strictly below; diagonal tiles in place
  if (bi < bj) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int r = 32 * bi + tx, c = 32 * bj + k;
    t[k][tx] = (r < n && c < n) ? M[(size_t)r + (size_t)c * ld] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int r = 32 * bj + tx, c = 32 * bi + k;       // element (r, c) of the upper part = element (c, r) of the lower part = t[r - 32 bj][c - 32 bi]
    if (r < n && c < n && r < c) M[(size_t)r + (size_t)c * ld] = t[tx][k];
  }
}

// W (rows x Npad, rows a multiple of 64) <- W L^-T with the dataflow kernel; returns false if it does not apply.
// abort word + ticket cleared and the exchange buffer filled with the sentinel: one launch instead of two fills
__global__ void __launch_bounds__(256) vec_solve_prep_kernel(int* __restrict__ flags, unsigned* __restrict__ x, size_t nwords, unsigned pattern) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < 2) flags[i] = 0;
  if (i < nwords) x[i] = pattern;
}

static bool launch_vec_solve(gpg_ctx* c, double* W, int ldw, int R, bool bwd) {
  const int Mt = c->Npad / 64;
  if (!ensure_tile_flags(c, 64)) return false;
  if (c->vec_x_cols < c->Npad || !c->vec_x) {
    if (c->vec_x) (void)hipFree(c->vec_x);
    c->vec_x_cols = 0;
    if (!gpg_dev_alloc(c, &c->vec_x, sizeof(double) * 4 * (size_t)c->Npad)) return false;
    c->vec_x_cols = c->Npad;
  }
  {
    const size_t nwords = (size_t)2 * R * c->Npad;                                             // abort word, ticket; sentinel words
    hipLaunchKernelGGL(vec_solve_prep_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, c->stream, c->tile_flags,
                       reinterpret_cast<unsigned*>(c->vec_x), nwords, (unsigned)(GPG_VEC_SENTINEL & 0xffffffffull));
  }
#define GPG_VEC_LAUNCH(BWD, RR)                                                                                          \
  hipLaunchKernelGGL((vec_solve_kernel<BWD, RR>), dim3(persistent_grid(c, vec_solve_kernel<BWD, RR>, Mt)), dim3(256), 0,  \
                     c->stream, VecSolveArgs{c->A, c->ld, c->dinv, W, ldw, Mt, R, (unsigned long long*)c->vec_x, c->Npad,  \
                                             c->tile_flags, c->tile_flags + 1, c->info})
  if (bwd) { if (R == 1) GPG_VEC_LAUNCH(1, 1); else GPG_VEC_LAUNCH(1, 4); }
  else     { if (R == 1) GPG_VEC_LAUNCH(0, 1); else GPG_VEC_LAUNCH(0, 4); }
#undef GPG_VEC_LAUNCH
  return true;
}

static bool launch_rows_bwd(gpg_ctx* c, double* Z, int ldz, int rows, int valid) {
  if (rows <= 0 || rows % 64 != 0) return false;
  if (valid >= 1 && valid <= 4 && c->Npad / 64 <= 512) return launch_vec_solve(c, Z, ldz, valid, true);
  const int Mt = c->Npad / 64, nrt = rows / 64;
  if ((long)Mt * nrt > c->rows_max_tasks) return false;
  const size_t nflag = (size_t)Mt * nrt + 2;                    // flags, abort word, ticket
  if (!ensure_tile_flags(c, nflag)) return false;
  (void)hipMemsetAsync(c->tile_flags, 0, sizeof(int) * flags_fill(nflag), c->stream);
  hipLaunchKernelGGL(rows_bwd_kernel, dim3(persistent_grid(c, rows_bwd_kernel, (long)Mt * nrt)), dim3(256), 0, c->stream,
                     (const double*)c->A, c->ld, (const double*)c->dinv, Z, ldz, Mt, nrt, valid < 0 ? rows : valid, c->tile_flags,
                     c->tile_flags + (nflag - 2), c->tile_flags + (nflag - 1), c->info);
  return true;
}

static bool launch_rows_fwd(gpg_ctx* c, double* W, int ldw, int rows, int valid) {
  if (rows <= 0 || rows % 64 != 0) return false;
  if (valid >= 1 && valid <= 4 && c->Npad / 64 <= 512) return launch_vec_solve(c, W, ldw, valid, false);
  const int Mt = c->Npad / 64, nrt = rows / 64;
  if ((long)Mt * nrt > c->rows_max_tasks) return false;   // beyond: the blocked sweep (or one launch per group of row tiles)
  const size_t nflag = (size_t)Mt * nrt + 2;                    // flags, abort word, ticket
  if (!ensure_tile_flags(c, nflag)) return false;
  (void)hipMemsetAsync(c->tile_flags, 0, sizeof(int) * flags_fill(nflag), c->stream);
  hipLaunchKernelGGL(rows_fwd_kernel, dim3(persistent_grid(c, rows_fwd_kernel, (long)Mt * nrt)), dim3(256), 0, c->stream,
                     (const double*)c->A, c->ld, (const double*)c->dinv, W, ldw, Mt, nrt, valid < 0 ? rows : valid, c->tile_flags,
                     c->tile_flags + (nflag - 2), c->tile_flags + (nflag - 1), c->info);
  return true;
}

}  // namespace

void gpg_launch_tile_chol(gpg_ctx* c, int c0) {
  launch_tile_chol(c, c0);
  if (c0 == 0) { c->last_factor_kernel = 1; c->last_factor_batch = 1; }
}
void gpg_launch_tile128_chol(gpg_ctx* c) {
  launch_tile128_chol(c);
  c->last_factor_kernel = c->last_launch_paired ? 3 : 2; c->last_factor_batch = 1;
}
// ---- value + gradient of ONE small matrix: W = L^-T overlapped with the factorisation ------------------------------------------------
// The two chains (factorisation: one diagonal tile after the other; W = L^-T: one tile column after the other) are each latency-bound on
// a small matrix and most of the chip idles through both.  The inverse's first launch goes to the context's lowest-priority stream
// (gpg_ctx::stream_inv) as soon as the factorisation has been enqueued; its tasks wait for the factorisation's diagonal-tile flags, so
// column tile j of W is under way right after tile column j of L.  Progress: both grids are persistent and draw tickets; a W task waits
// only for factorisation tasks (which never wait for W) and for W tasks with lower tickets.  The W launch becomes runnable as soon as
// the factorisation's flags have been cleared -- possibly BEFORE the factorisation kernel is resident -- so its grid is capped at half
// the co-resident capacity (gpg_ctx::grid_cap): whatever the dispatch order, the factorisation finds free slots, advances, and behind
// it the inverse.  Bounded waits apply as everywhere; a timeout here repeats the call once without the overlap (api.hip).
static bool single_uses_tile64(const gpg_ctx* c) { return c->tail_cols > 0 && c->Npad <= c->tail_cols; }
static size_t chol64_per(const gpg_ctx* c) {
  const size_t Mt = c->Npad / 64, Rt = c->ld / 64;
  return Mt * Rt + 1 + 4 * Mt;
}
// flag words of the factorisation launch that is overlapped: B matrices on 64-tiles, or ONE matrix on 128-tiles (large matrices)
static size_t chol64_nflag(const gpg_ctx* c, int B) {
  if (B == 1 && !single_uses_tile64(c)) {
    const size_t Mt = c->Npad / 128, Rt = c->ld / 128;
    return flags_fill(Mt * Rt + 1 + 9 * Mt + 8);
  }
  return flags_fill(chol64_per(c) * B + 8);
}
// Does a batch of B matrices go to the 128-tile factorisation kernel?  (gpg_launch_tile_chol_batch below decides with this.)
static bool batch_uses_tile128(const gpg_ctx* c, int B) {
  if (c->tail_cols >= (1 << 30)) return false;
  if (c->tail_cols == 0) return true;
  return c->Npad > c->tail_cols || (c->Npad >= 2048 && (long)B * (c->Npad / 128) >= 320);
}
bool gpg_overlap_inverse_begin(gpg_ctx* c, int B) {
  const bool small_inv = c->inv_tile64_cols > 0 && c->Npad <= c->inv_tile64_cols;   // W on 64-tiles; above (one matrix only): on 128-tiles
  // the factorisation must be one of the dataflow launches that take the flag override: B matrices on 64-tiles, one on 64- or 128-tiles
  // (the 64-tile W reads 64-tile flags: it needs the 64-tile factorisation; the 128-tile W takes either)
  const bool chol_ok = B > 1 ? !batch_uses_tile128(c, B) : (single_uses_tile64(c) || (c->chol_impl == 1 && !small_inv));
  if (!c->overlap_inverse || B < 1 || (!small_inv && B > 1) || !chol_ok || c->prof_mask != 0 || c->stream_inv == c->stream || !c->stream_inv)
    return false;
  const size_t Mt64 = c->Npad / 64;
  const size_t need = chol64_nflag(c, B) + flags_fill(Mt64 * Mt64 * B + 16);           // (the 128-tile W needs a quarter of the second term)
  if (c->keep_flags_cap < need) {
    if (c->keep_flags) (void)hipFree(c->keep_flags);
    c->keep_flags = nullptr; c->keep_flags_cap = 0;
    if (hipMalloc(reinterpret_cast<void**>(&c->keep_flags), sizeof(int) * need) != hipSuccess) { (void)hipGetLastError(); c->keep_flags = nullptr; return false; }
    c->keep_flags_cap = need;
  }
  if (!c->ev_flags && hipEventCreateWithFlags(&c->ev_flags, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (!c->ev_trinv && hipEventCreateWithFlags(&c->ev_trinv, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (!c->ev_winit && hipEventCreateWithFlags(&c->ev_winit, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
  c->chol_flags_override = c->keep_flags;
  return true;
}
bool gpg_overlap_inverse_trinv(gpg_ctx* c, int B, const double* Abase, size_t a_stride, const double* dinv_base, int d_stride, double* Wbase,
                               int* info_base) {
  hipStream_t main_stream = c->stream;
  (void)hipStreamWaitEvent(c->stream_inv, c->ev_flags, 0);          // the factorisation's flags have been cleared
  c->stream = c->stream_inv;
  c->grid_cap = c->num_cus > 0 ? c->num_cus : 256;                 // one workgroup per compute unit: half of the two-per-CU capacity
  c->overlap_used = true;
  // (last argument: per-matrix stride of the factorisation's flags for the batched 64-tile W; for the 128-tile W of one matrix the number of
  // 64-tile columns of the factorisation, 0 when it ran on 128-tiles itself)
  const bool small_inv = c->inv_tile64_cols > 0 && c->Npad <= c->inv_tile64_cols;
  const int lf = small_inv ? (int)chol64_per(c) : (single_uses_tile64(c) ? c->Npad / 64 : 0);
  const bool ok = launch_tile128_inverse_batch(c, B, Abase, a_stride, dinv_base, d_stride, Wbase, nullptr, info_base, 1,
                                               c->keep_flags + chol64_nflag(c, B), c->keep_flags, lf);
  (void)hipEventRecord(c->ev_trinv, c->stream_inv);
  c->stream = main_stream;
  c->grid_cap = 0;
  return ok;
}
// The factorisation that W = L^-T overlaps has failed (large matrices: the host reads its info word before going on): raise the abort word
// of the W launch; its tasks look at it first and the launch drains within one task.  One matrix.
void gpg_overlap_inverse_cancel(gpg_ctx* c) {
  const bool small_inv = c->inv_tile64_cols > 0 && c->Npad <= c->inv_tile64_cols;
  const size_t Mt = small_inv ? c->Npad / 64 : c->Npad / 128;
  int* ones = c->keep_flags + chol64_nflag(c, 1) + Mt * Mt;                            // layout of launch_tile128_inverse_batch: ... | 9 ones | abort | tickets
  (void)hipMemsetD32Async((hipDeviceptr_t)(ones + 9), 1, 1, c->stream);
  // wait until it has drained: workgroups that were inside a dependency wait report the abort through the info word like a timed-out
  // wait, and the next call clears that word on the main stream -- nothing of this launch may write it afterwards
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamSynchronize(c->stream_inv);
}
bool gpg_overlap_inverse_wwt(gpg_ctx* c, int B, const double* Abase, size_t a_stride, const double* dinv_base, int d_stride, double* Wbase,
                             double* Mbase, int* info_base) {
  if (B == 1 && c->overlap_inverse >= 2) {   // -(W W^T) too follows W tile column by tile column (its flags), on the main stream
    // ... but not before W's flags / abort word / tickets have been cleared and W set to the identity on the other stream: keep_flags is
    // reused across calls, its stale content reads "every tile of W is done"
    (void)hipStreamWaitEvent(c->stream, c->ev_winit, 0);
    return launch_tile128_inverse_batch(c, 1, Abase, a_stride, dinv_base, d_stride, Wbase, Mbase, info_base, 3, c->keep_flags + chol64_nflag(c, 1));
  }
  (void)hipStreamWaitEvent(c->stream, c->ev_trinv, 0);
  return launch_tile128_inverse_batch(c, B, Abase, a_stride, dinv_base, d_stride, Wbase, Mbase, info_base, 2, c->keep_flags + chol64_nflag(c, B));
}
bool gpg_launch_tile128_inverse(gpg_ctx* c, double* W, double* Minv) { return launch_tile128_inverse(c, W, Minv); }
// out_dev[0] = squared Frobenius norm of the symmetric N x N matrix whose lower triangle sits in M (leading dimension ld)
void gpg_launch_frob_lower(gpg_ctx* c, const double* M, int ld, double* partial, double* out_dev) {
  hipLaunchKernelGGL(frob_lower_cols_kernel, dim3(c->N), dim3(256), 0, c->stream, M, ld, c->N, partial);
  hipLaunchKernelGGL(sum_fixed_order_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)partial, c->N, out_dev);
}
void gpg_launch_symmetrize(gpg_ctx* c, double* M, int ld) {
  const int nt = (c->Npad + 31) / 32;
  hipLaunchKernelGGL(symmetrize_kernel, dim3(nt, nt), dim3(256), 0, c->stream, M, ld, c->Npad);
}
// M (lower triangle, leading dimension Npad) = - Sa Sb^T for full Npad x Npad operands (leading dimension Npad)
bool gpg_launch_full_abt(gpg_ctx* c, const double* Sa, const double* Sb, double* M) {
  const int Mt = c->Npad / 128;
  const unsigned long long key = (3ull << 61) | (1ull << 59) | (unsigned long long)Mt;
  auto it = c->tilemaps.find(key);
  if (it == c->tilemaps.end()) {
    std::vector<int> list;
    for (int a = 0; a < Mt; ++a)
      for (int b = 0; b <= a; ++b) list.push_back(a | (b << 16));
    TileMap tm;
    tm.n = (int)list.size();
    if (!gpg_dev_alloc(c, &tm.dev, sizeof(int) * list.size())) return false;
    (void)hipMemcpy(tm.dev, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice);
    it = c->tilemaps.emplace(key, tm).first;
  }
  const TileMap& tm = it->second;
  if (!ensure_tile_flags(c, 16)) return false;
  (void)hipMemsetAsync(c->tile_flags, 0, sizeof(int) * 16, c->stream);
  hipLaunchKernelGGL(tile128_wwt_kernel, dim3(persistent_grid(c, tile128_wwt_kernel, tm.n)), dim3(256), 0, c->stream, Sa, c->Npad, M,
                     c->Npad, Mt, (const int*)tm.dev, tm.n, c->tile_flags, (const int*)nullptr, (size_t)0, Sb, 1, (int*)nullptr, (int*)nullptr,
                     (int*)nullptr);
  return true;
}
bool gpg_launch_tile128_inverse_batch(gpg_ctx* c, int B, const double* Abase, size_t a_stride, const double* dinv_base, int d_stride,
                                      double* Wbase, double* Mbase, int* info_base) {
  return launch_tile128_inverse_batch(c, B, Abase, a_stride, dinv_base, d_stride, Wbase, Mbase, info_base);
}
bool gpg_launch_rows_fwd(gpg_ctx* c, double* W, int ldw, int rows, int valid) { return launch_rows_fwd(c, W, ldw, rows, valid); }
bool gpg_launch_rows_bwd(gpg_ctx* c, double* Z, int ldz, int rows, int valid) { return launch_rows_bwd(c, Z, ldz, rows, valid); }
void gpg_launch_tile_chol_batch(gpg_ctx* c, int B, double* Abase, size_t a_stride, double* dinv_base, int d_stride,
                                int* info_base) {
  // Which tile size: a forced mode decides; otherwise the 128-tile kernel (higher MFMA rate per workgroup, longer
  // dependency chain per column) as soon as the batch puts enough of its tasks in flight to hide that chain
  // (measured, tools/tile_probe 5 / 6: 2560 columns x 8 matrices 21 TF against 28 for the 64-tile kernel, x 32
  // matrices 37 against 33; 4608 x 8: 42 / 39; 9216 x 8: 59 / 47).
  const bool use128 = batch_uses_tile128(c, B);
  if (use128) launch_tile128_chol_batch(c, B, Abase, a_stride, dinv_base, d_stride, info_base);
  else launch_tile_chol_batch(c, B, Abase, a_stride, dinv_base, d_stride, info_base);
  c->last_factor_kernel = use128 ? (c->last_launch_paired ? 3 : 2) : 1;
  c->last_factor_batch = B;
}
