// Diagnostic (not product code): what does the quad-row substitution of a 64-column block cost a wave, in cycles per column step,
//   (a) alone on its SIMD, (b) beside a wave that issues v_mfma_f64_16x16x4 back to back (the partner workgroup's MFMA loop),
// with the L image in LDS (product) or in registers (no LDS traffic), at wave priority 0 or 3?  No global memory in the timed part.
//   ./subst_probe [reps=64]
#include "../gpgradpy_amd/csrc/chol_device.h"   // (direct_tile_gemm_acc, GPG_QUAD_SUBST2)
#include <cstdio>
#include <cstdlib>
#include <vector>

__shared__ __attribute__((aligned(16))) double sp_Ls[64][4][18];
__shared__ double sp_sdinv[64];

// role 0: substitution waves; role 1: MFMA waves.  grid = 2 * CUs workgroups of 256 threads; the FIRST half of the grid substitutes
// when (mode & 1), the second half runs MFMAs when (mode & 2): workgroup b goes to XCD b % 8 and the XCD fills its CUs one workgroup
// each before the second round, so that every CU gets one workgroup of each role (checked: hw[] records XCC / SE / CU ids; an
// even / odd split puts the two roles on different XCDs and measures nothing).  out[wg] = cycles of the timed loop of wave 0.
template <int VARIANT, int KSV = 16>
__global__ void __launch_bounds__(256, 2) subst_probe_kernel(int mode, int reps, int prio, unsigned long long* out, double* sink, unsigned* hw,
                                                             const double* panel) {
  const int tid = threadIdx.x, q = tid & 3;
  const bool subst_role = blockIdx.x < gridDim.x / 2 && !(mode & 32);   // mode & 32: every workgroup runs the product MFMA loop (its rate in isolation)
  if (tid == 0) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // HW_REG_XCC_ID: xcc_id [3:0]
    hw[blockIdx.x] = (xcc << 16) | (hwid & 0xff00);
  }
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
  } else if (mode & 4) {   // partner = the product's MFMA loop (direct_tile_gemm_acc) over a small panel that stays in L2
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 1, wn = w >> 1, l15 = lane & 15, l4 = lane >> 4;
    d4 acc[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ni][mi][r] = 0.0;
    const int ldp = 256;                                       // panel: 256 rows x 512 columns of doubles = 1 MB, shared by all workgroups
    const unsigned lane_off = (unsigned)(2 * l15 + l4 * ldp) * 8u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r)
      direct_tile_gemm_acc<1, 4, KSV>(acc, panel + wm * 64, lane_off, ldp, panel + 128 + wn * 64, lane_off, ldp, 128);
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) s += acc[ni][0][0] + acc[ni][3][3];
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) out[blockIdx.x] = (t1 - t0) * 40 / 256;      // scaled so that the host's "/ (reps * 40 * 8)" gives cycles per MFMA (128 k-steps x 16 per rep)
  } else if (mode & 16) {   // partner = register-only MFMA stream shaped like the product loop: 16 accumulators, 4 x 4 distinct operand registers per k-step
    d4 acc[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ni][mi][r] = 0.0;
    double fa[2][4], fb[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) { fa[u][i] = 1.0 + (tid + i + 4 * u) * 1e-9; fb[u][i] = 1.0 - (tid + i + 4 * u) * 1e-9; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps * 20; ++r) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[u][ni], fa[u][mi], acc[ni][mi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) s += acc[ni][0][0] + acc[ni][3][3];
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) out[blockIdx.x] = (t1 - t0) / 2;            // 32 MFMAs per iteration, reps * 20 iterations; host divides by reps * 40 * 8
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
  unsigned long long* out; double* sink; unsigned* hw; double* panel;
  hipMalloc(&out, grid * 8); hipMalloc(&sink, (size_t)grid * 256 * 8); hipMalloc(&hw, grid * 4);
  hipMalloc(&panel, 256 * 1024 * 8);
  {
    std::vector<double> hp(256 * 1024);
    for (size_t i = 0; i < hp.size(); ++i) hp[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;   // (zeros make the MFMA pipe look faster than it is)
    hipMemcpy(panel, hp.data(), hp.size() * 8, hipMemcpyHostToDevice);
  }
  std::vector<unsigned long long> h(grid);
  std::vector<unsigned> hh(grid);
  for (int variant = 0; variant < 2; ++variant)
    for (int mode : {1, 3, 19, 7, 15, 36})
      for (int prio : {0, 1}) {
        hipMemset(out, 0, grid * 8);
        for (int rep = 0; rep < 2; ++rep) {
          if (mode == 15) hipLaunchKernelGGL((subst_probe_kernel<0, 0>), dim3(grid), dim3(256), 0, 0, 7, reps, prio, out, sink, hw, panel);   // product loop without its barrier
          else if (variant == 0) hipLaunchKernelGGL((subst_probe_kernel<0, 16>), dim3(grid), dim3(256), 0, 0, mode, reps, prio, out, sink, hw, panel);
          else hipLaunchKernelGGL((subst_probe_kernel<1, 16>), dim3(grid), dim3(256), 0, 0, mode, reps, prio, out, sink, hw, panel);
          hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
        double ssum = 0, msum = 0; int sn = 0, mn = 0;
        for (int b = 0; b < grid; ++b) { if (!h[b]) continue; if (b >= grid / 2 || mode == 36) { msum += h[b]; ++mn; } else { ssum += h[b]; ++sn; } }
        if (mode == 3 && prio == 0 && variant == 0) {   // do the two roles share compute units?
          hipMemcpy(hh.data(), hw, grid * 4, hipMemcpyDeviceToHost);
          int shared = 0;
          for (int a = 0; a < grid / 2; ++a)
            for (int b = grid / 2; b < grid; ++b)
              if (hh[a] == hh[b]) { ++shared; break; }
          printf("(%d of %d substituting workgroups share their CU with an MFMA workgroup)\n", shared, grid / 2);
        }
        if (mode == 36) {
          if (variant == 0 && prio == 0) printf("product MFMA loop on every workgroup (two per CU, panel resident in L2): %.1f cycles per MFMA and wave (128 = the pipe shared perfectly by two waves)\n", msum / mn / (reps * 40.0 * 8));
          continue;
        }
        printf("%s image, %s, prio %d: substitution %.0f cycles per 64-column block (two rows per quad) = %.1f per column step",
               variant ? "register" : "LDS", mode == 3 ? "beside MFMA waves" : mode == 19 ? "beside a register-only MFMA stream of the product's shape (16 accumulators)" : mode == 7 ? "beside the product MFMA loop" : mode == 15 ? "beside the product MFMA loop without its barrier" : "alone", 3 * prio, sn ? ssum / sn / reps : 0.0, sn ? ssum / sn / reps / 64 : 0.0);
        if (mn) printf(" | mfma %.1f cycles/instr", msum / mn / (reps * 40.0 * 8));
        printf("\n");
      }
  return 0;
}
