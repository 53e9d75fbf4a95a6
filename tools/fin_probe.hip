// Diagnostic (not product code): the off-diagonal finalisation X <- X L^-T of one 128 x 128 tile against a finished 128 x 128
// diagonal tile, timed inside a workload that looks like the factorisation (every workgroup alternates a burst of the
// direct-fragment MFMA loop with one finalisation; two workgroups per CU, out of phase), in two versions:
//   mode 0  the product's: accumulators -> memory -> quad-row VALU substitution (tile_solve_rows128)
//   mode 1  MFMA-blocked, on the accumulators as the MFMA loop leaves them: rank-8 steps -- the 8 x 8 diagonal blocks by exact VALU
//           substitution across the four lane groups, everything below them by v_mfma_f64_16x16x4 with the finished accumulator
//           registers as B operands (register r of a 16 x 16 D tile holds rows 4r..4r+3 in the B-operand layout) and -L from an LDS
//           image as A operands; the update of column block 1 is one more run of the direct-fragment loop over the stored X1.
// Prints the mean finalisation time per mode and checks both against a CPU triangular solve.
//    ./fin_probe [burst_ksteps=2048] [iters=8]
#include "../gpgradpy_amd/csrc/cholesky.hip"
#include "../gpgradpy_amd/csrc/cholesky_dataflow.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
void gpg_prof_begin(gpg_ctx*, int, double) {}
void gpg_prof_end(gpg_ctx*) {}
void gpg_launch_identity(gpg_ctx*, double*, int) {}

namespace {
constexpr int FLD = 66;                                   // leading dimension of the LDS image of a 64 x 64 diagonal block
__shared__ __attribute__((aligned(16))) double fin_Lc[64 * FLD];   // Lc[c * FLD + r] = L[r][c] (r >= c), 0 above the diagonal
__shared__ double fin_dinv[64];

// image of the 64 x 64 block at Lb (column-major, leading dimension ldl) + reciprocal pivots; whole workgroup, ends with a barrier
__device__ __forceinline__ void load_block_image(const double* Lb, int ldl, const double* dinv) {
  const int t = threadIdx.x, c = t >> 2, rb = (t & 3) * 16;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int r = rb + u;
    fin_Lc[c * FLD + r] = r >= c ? Lb[r + (size_t)c * ldl] : 0.0;
  }
  if (t < 64) fin_dinv[t] = dinv[t];
  __syncthreads();
}

// one rank-8 step of the substitution of this wave's 64 columns: columns c0 .. c0+7, c0 = 32 P0 + 8 R0 (block-local)
template <int P0, int R0>
__device__ __forceinline__ void subst_step(d4 (&acc)[4][4], int l15, int l4) {
  constexpr int c0 = 32 * P0 + 8 * R0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    __builtin_amdgcn_sched_barrier(0);                     // keep the LDS reads of later steps where they are (registers)
    const double d0 = fin_dinv[c0 + 2 * q], d1 = fin_dinv[c0 + 2 * q + 1], l10 = fin_Lc[(c0 + 2 * q) * FLD + c0 + 2 * q + 1];
    const int rr = c0 + 2 * l4;
    const double c00 = fin_Lc[(c0 + 2 * q) * FLD + rr], c01 = fin_Lc[(c0 + 2 * q + 1) * FLD + rr];
    const double c10 = fin_Lc[(c0 + 2 * q) * FLD + rr + 1], c11 = fin_Lc[(c0 + 2 * q + 1) * FLD + rr + 1];
    const bool mine = l4 == q, below = l4 > q;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      double a0 = acc[2 * P0][mi][R0], a1 = acc[2 * P0 + 1][mi][R0];
      const double f0 = a0 * d0, f1 = (a1 - f0 * l10) * d1;
      a0 = mine ? f0 : a0;
      a1 = mine ? f1 : a1;
      const double b0 = __shfl(a0, l15 + 16 * q, 64), b1 = __shfl(a1, l15 + 16 * q, 64);
      const double t0 = a0 - (b0 * c00 + b1 * c01), t1 = a1 - (b0 * c10 + b1 * c11);
      acc[2 * P0][mi][R0] = below ? t0 : a0;
      acc[2 * P0 + 1][mi][R0] = below ? t1 : a1;
    }
  }
#pragma unroll
  for (int pt = P0; pt < 2; ++pt)
#pragma unroll
    for (int et = 0; et < 2; ++et) {
      if (pt == P0 && R0 == 3) continue;                  // no rows left below this block in its own 32-column group
      const int row = 32 * pt + 2 * l15 + et;
#pragma unroll
      for (int es = 0; es < 2; ++es) {
        __builtin_amdgcn_sched_barrier(0);
        double a = -fin_Lc[(c0 + 2 * l4 + es) * FLD + row];
        if (pt == P0) a = l15 >= 4 * R0 + 4 ? a : 0.0;    // rows of this block and above it are final
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[2 * pt + et][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[2 * P0 + es][mi][R0], acc[2 * pt + et][mi], 0, 0, 0);
      }
    }
}

__device__ __forceinline__ void subst_block(d4 (&acc)[4][4], int l15, int l4) {
  subst_step<0, 0>(acc, l15, l4); subst_step<0, 1>(acc, l15, l4); subst_step<0, 2>(acc, l15, l4); subst_step<0, 3>(acc, l15, l4);
  subst_step<1, 0>(acc, l15, l4); subst_step<1, 1>(acc, l15, l4); subst_step<1, 2>(acc, l15, l4); subst_step<1, 3>(acc, l15, l4);
}

__device__ __forceinline__ void load_acc(d4 (&acc)[4][4], const double* Cw, int ld) {
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
}
__device__ __forceinline__ void store_acc(const d4 (&acc)[4][4], double* Cw, int ld) {
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

// X (this workgroup's tile, ld 128) <- acc L^-T, MFMA-blocked.  L: 128 x 128 lower triangular, column-major ldl; dinv its reciprocal pivots.
__device__ __forceinline__ void finalize_mfma(d4 (&acc)[4][4], double* X, int ldx, const double* L, int ldl, const double* dinv) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w & 1, wn = w >> 1, l15 = lane & 15, l4 = lane >> 4;
  double* Cw = X + wm * 64 + 2 * l15 + (size_t)(wn * 64 + 2 * l4) * ldx;
#pragma nounroll
  for (int half = 0; half < 2; ++half) {                  // column block 0 (waves wn = 0), then column block 1 (waves wn = 1)
    if (half == 1 && wn == 1)
      direct_tile_gemm_x2<GPG_MFMA_PF>(acc, X + wm * 64 + 2 * l15 + (size_t)l4 * ldx, ldx, L + 64 + 2 * l15 + (size_t)l4 * ldl, ldl, 16);
    load_block_image(L + (size_t)half * (64 + (size_t)64 * ldl), ldl, dinv + 64 * half);
    if (wn == half) {
      subst_block(acc, l15, l4);
      store_acc(acc, Cw, ldx);
    }
    __syncthreads();                                      // X1 is in memory (same CU: its L1 is write-through), the image is free
  }
}

template <int MODE>
__global__ void __launch_bounds__(256, 2) fin_probe_kernel(const double* __restrict__ X0, double* Xall, const double* L, const double* dinv,
                                                           const double* panels, int burst, int iters, int* ones, int* scratch,
                                                           unsigned long long* t_out) {
  __shared__ int sh_ok;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w & 1, wn = w >> 1, l15 = lane & 15, l4 = lane >> 4;
  double* X = Xall + (size_t)blockIdx.x * 128 * 128;
  const double* pan = panels + (size_t)blockIdx.x * 2 * 128 * 2048;     // own A and B panels, 128 x 2048 each
  d4 acc[4][4];
  unsigned long long tsum = 0;
  for (int it = 0; it < iters; ++it) {
    const double* x0p = X0;
    asm volatile("" : "+s"(x0p));                      // not loop-invariant for the compiler: no hoisting of the tile loads
    // MFMA burst (odd workgroups: half a burst first, so that the two workgroups of a CU are out of phase)
    load_acc(acc, x0p + wm * 64 + 2 * l15 + (size_t)(wn * 64 + 2 * l4) * 128, 128);
    int ks = (it == 0 && (blockIdx.x & 1)) ? burst / 2 : burst;
    for (int done = 0; done < ks; done += 512)
      direct_tile_gemm_x2<GPG_MFMA_PF>(acc, pan + wm * 64 + 2 * l15 + (size_t)l4 * 128, 128, pan + 128 * 2048 + wn * 64 + 2 * l15 + (size_t)l4 * 128, 128, 512);
    // keep the burst's result alive, then finalise the clean tile
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) asm volatile("" ::"v"(acc[ni][mi]));
    asm volatile("" : "+s"(x0p));
    load_acc(acc, x0p + wm * 64 + 2 * l15 + (size_t)(wn * 64 + 2 * l4) * 128, 128);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_setprio(2);
    if (MODE == 0) {
      store_acc(acc, X + wm * 64 + 2 * l15 + (size_t)(wn * 64 + 2 * l4) * 128, 128);
      __syncthreads();
      tile_solve_rows128(L, 128, dinv, X, 128, t128_U, t128_Ls, t128_sdinv, ones, ones + 4, ones, scratch, scratch + 1, &sh_ok);
    } else {
      finalize_mfma(acc, X, 128, L, 128, dinv);
    }
    __builtin_amdgcn_s_setprio(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    tsum += t1 - t0;
  }
  if (tid == 0) t_out[blockIdx.x] = tsum;
}
}  // namespace

int main(int argc, char** argv) {
  const int burst = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 8;
  const int G = 512, T = 128;
  std::vector<double> hL((size_t)T * T, 0.0), hd(T), hX((size_t)T * T), ref((size_t)T * T);
  srand(7);
  auto rnd = [] { return (rand() & 0xffff) / 65536.0 - 0.5; };
  for (int c = 0; c < T; ++c)
    for (int r = c; r < T; ++r) hL[r + (size_t)c * T] = r == c ? 1.0 + 0.5 * (rnd() + 0.5) : 0.2 * rnd();
  for (int c = 0; c < T; ++c) hd[c] = 1.0 / hL[c + (size_t)c * T];
  for (auto& v : hX) v = rnd();
  // reference: row by row forward substitution  x L^T = b
  for (int r = 0; r < T; ++r)
    for (int c = 0; c < T; ++c) {
      double s = hX[r + (size_t)c * T];
      for (int m = 0; m < c; ++m) s -= ref[r + (size_t)m * T] * hL[c + (size_t)m * T];
      ref[r + (size_t)c * T] = s / hL[c + (size_t)c * T];
    }
  double *dL, *dd, *dX0, *dX, *dP;
  int *dones, *dscr;
  unsigned long long* dt;
  hipMalloc(&dL, 8 * hL.size()); hipMalloc(&dd, 8 * T); hipMalloc(&dX0, 8 * hX.size()); hipMalloc(&dX, 8 * hX.size() * G);
  hipMalloc(&dP, 8 * (size_t)G * 2 * 128 * 2048); hipMalloc(&dones, 64 * 4); hipMalloc(&dscr, 64 * 4); hipMalloc(&dt, 8 * G);
  hipMemcpy(dL, hL.data(), 8 * hL.size(), hipMemcpyHostToDevice); hipMemcpy(dd, hd.data(), 8 * T, hipMemcpyHostToDevice);
  hipMemcpy(dX0, hX.data(), 8 * hX.size(), hipMemcpyHostToDevice);
  hipMemset(dP, 0, 8 * (size_t)G * 2 * 128 * 2048);
  std::vector<int> one(64, 1);
  hipMemcpy(dones, one.data(), 64 * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(dscr, 0, 64 * 4);
    hipMemset(dX, 0, 8 * hX.size() * G);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    if (mode == 0) hipLaunchKernelGGL(fin_probe_kernel<0>, dim3(G), dim3(256), 0, 0, dX0, dX, dL, dd, dP, burst, iters, dones, dscr, dt);
    else hipLaunchKernelGGL(fin_probe_kernel<1>, dim3(G), dim3(256), 0, 0, dX0, dX, dL, dd, dP, burst, iters, dones, dscr, dt);
    hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> ht(G);
    std::vector<double> out(hX.size() * G);
    hipMemcpy(ht.data(), dt, 8 * G, hipMemcpyDeviceToHost);
    hipMemcpy(out.data(), dX, 8 * out.size(), hipMemcpyDeviceToHost);
    double tmean = 0, err = 0;
    for (int g = 0; g < G; ++g) tmean += (double)ht[g] / iters / 100.0;
    for (int g = 0; g < G; g += 37)
      for (size_t k = 0; k < hX.size(); ++k) err = fmax(err, fabs(out[g * hX.size() + k] - ref[k]));
    printf("mode %d (%s): kernel %.2f ms, mean finalisation %.1f us per tile, max |X - ref| = %.2e\n", mode,
           mode == 0 ? "memory round trip + VALU quad substitution" : "MFMA-blocked on the accumulators", ms, tmean / G, err);
  }
  return 0;
}
