// Diagnostic (not product code): what does the quad-row substitution of a 64-column block cost a wave, in cycles per column step,
//   (a) alone on its SIMD, (b) beside a wave that issues v_mfma_f64_16x16x4 back to back (the partner workgroup's MFMA loop),
// with the L image in LDS (product) or in registers (no LDS traffic), at wave priority 0 or 3?  No global memory in the timed part.
//   ./subst_probe [reps=64]
#include "../gpgradpy_amd/csrc/chol_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

__shared__ __attribute__((aligned(16))) double sp_Ls[64][4][18];
__shared__ double sp_sdinv[64];

// role 0: substitution waves; role 1: MFMA waves.  grid = 2 * CUs workgroups of 256 threads; even workgroups substitute when
// (mode & 1), odd ones run MFMAs when (mode & 2).  out[wg] = cycles of the timed loop of wave 0.
template <int VARIANT>
__global__ void __launch_bounds__(256, 2) subst_probe_kernel(int mode, int reps, int prio, unsigned long long* out, double* sink) {
  const int tid = threadIdx.x, q = tid & 3;
  const bool subst_role = (blockIdx.x & 1) == 0;
  if (subst_role) {
    if (!(mode & 1)) return;
    for (int t = tid; t < 64 * 64; t += 256) {
      const int j = t >> 6, k = t & 63;
      sp_Ls[j][k & 3][k >> 2] = k > j ? 1e-3 * ((k * 7 + j * 3) % 11) : 0.0;
    }
    if (tid < 64) sp_sdinv[tid] = 1.0 + 1e-3 * tid;
    __syncthreads();
    double x0[16], x1[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { x0[m] = 1.0 + 1e-2 * (tid + m); x1[m] = 2.0 - 1e-2 * (tid - m); }
    if (prio) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
      if (VARIANT == 0) { GPG_QUAD_SUBST2(x0, x1, sp_Ls, sp_sdinv, q) }
      else {   // same arithmetic, L values from registers: no LDS reads inside the chain
        double lvr[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) lvr[m] = sp_Ls[m][q][m] + 1e-3;
#pragma unroll
        for (int mj = 0; mj < 16; ++mj) {
#define SP_STEP(QJ)                                                                          \
          {                                                                                  \
            const double xj0 = quad_bcast<QJ>(x0[mj]), xj1 = quad_bcast<QJ>(x1[mj]);         \
            _Pragma("unroll") for (int m = mj; m < 16; ++m) { x0[m] -= xj0 * lvr[m]; x1[m] -= xj1 * lvr[m]; } \
            __builtin_amdgcn_sched_barrier(0);                                               \
          }
          SP_STEP(0) SP_STEP(1) SP_STEP(2) SP_STEP(3)
#undef SP_STEP
        }
      }
#pragma unroll
      for (int m = 0; m < 16; ++m) { asm volatile("" : "+v"(x0[m])); asm volatile("" : "+v"(x1[m])); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (prio) __builtin_amdgcn_s_setprio(0);
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) s += x0[m] + x1[m];
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) out[blockIdx.x] = t1 - t0;
  } else {
    if (!(mode & 2)) return;
    d4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] = 0.0;
    const double a = 1.0 + tid * 1e-9, b = 1.0 - tid * 1e-9;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps * 40; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) out[blockIdx.x] = t1 - t0;
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 64;
  int ncu = 256;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  const int grid = 2 * ncu;
  unsigned long long* out; double* sink;
  hipMalloc(&out, grid * 8); hipMalloc(&sink, (size_t)grid * 256 * 8);
  std::vector<unsigned long long> h(grid);
  for (int variant = 0; variant < 2; ++variant)
    for (int mode : {1, 3})
      for (int prio : {0, 1}) {
        hipMemset(out, 0, grid * 8);
        for (int rep = 0; rep < 2; ++rep) {
          if (variant == 0) hipLaunchKernelGGL(subst_probe_kernel<0>, dim3(grid), dim3(256), 0, 0, mode, reps, prio, out, sink);
          else hipLaunchKernelGGL(subst_probe_kernel<1>, dim3(grid), dim3(256), 0, 0, mode, reps, prio, out, sink);
          hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
        double ssum = 0, msum = 0; int sn = 0, mn = 0;
        for (int b = 0; b < grid; ++b) { if (!h[b]) continue; if (b & 1) { msum += h[b]; ++mn; } else { ssum += h[b]; ++sn; } }
        printf("%s image, %s, prio %d: substitution %.0f cycles per 64-column block (two rows per quad) = %.1f per column step",
               variant ? "register" : "LDS", mode == 3 ? "beside MFMA waves" : "alone", 3 * prio, sn ? ssum / sn / reps : 0.0, sn ? ssum / sn / reps / 64 : 0.0);
        if (mn) printf(" | mfma %.1f cycles/instr", msum / mn / (reps * 40.0 * 8));
        printf("\n");
      }
  return 0;
}
