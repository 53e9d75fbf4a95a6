// Diagnostic probe (not product code): measures the gfx950 fp64 MFMA and fp64 VALU FMA issue
// rates and verifies the v_mfma_f64_16x16x4_f64 operand / result lane maps with exact integers.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_probe.hip -o mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// --- layout check: D = A(16x4) * B(4x16) with one wave
__global__ void layout_kernel(const double* A, const double* B, double* D) {
  int l = threadIdx.x;
  // claimed map: lane l holds A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  // claimed result map: col = l&15, row = (l>>4) + 4*reg
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) mfma_rate(double* out, int iters, double seed) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) valu_rate(double* out, int iters, double seed) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  double a = seed + threadIdx.x * 1e-9, b = 1e-9 * seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// hybrid: waves 0..3 MFMA, waves 4..7 VALU (512 threads per block, one block per CU)
__global__ void __launch_bounds__(512) hybrid_rate(double* out, int iters, double seed) {
  int w = threadIdx.x >> 6;
  double s = 0;
  if (w < 4) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double a = seed + threadIdx.x * 1e-9, b = 1e-9 * seed;
    for (int it = 0; it < iters * 8; ++it) {   // 8 mfma * 64cyc = 512 cyc ; 16 fma*4cyc = 64 cyc -> x8
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    for (int i = 0; i < 16; ++i) s += acc[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  // layout check
  std::vector<double> A(64), B(64), D(256), Dref(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = (i + 1) * 10 + k;          // asymmetric
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (k + 1) * 100 + 3 * j + 1;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) Dref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dD;
  CHECK(hipMalloc(&dA, 64 * 8)); CHECK(hipMalloc(&dB, 64 * 8)); CHECK(hipMalloc(&dD, 256 * 8));
  CHECK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
  layout_kernel<<<1, 64>>>(dA, dB, dD);
  CHECK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 256; ++i) if (D[i] != Dref[i]) ++bad;
  printf("layout check: %d mismatches of 256 (%s)\n", bad, bad ? "FAIL" : "OK");

  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int ncu = prop.multiProcessorCount;
  printf("device %s, CUs %d, clock %d kHz\n", prop.name, ncu, prop.clockRate);
  double* out; CHECK(hipMalloc(&out, (size_t)ncu * 8 * 1024 * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms;
  const int iters = 20000;
#define RUN_MFMA(NACC, WPC)                                                                       \
  {                                                                                               \
    int blocks = ncu * (WPC) / 4;                                                                 \
    mfma_rate<NACC><<<blocks, 256>>>(out, 100, 1.0);                                              \
    CHECK(hipDeviceSynchronize());                                                                \
    CHECK(hipEventRecord(e0));                                                                    \
    mfma_rate<NACC><<<blocks, 256>>>(out, iters, 1.0);                                            \
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));                                    \
    CHECK(hipEventElapsedTime(&ms, e0, e1));                                                      \
    double fl = (double)blocks * 4 * iters * NACC * 2048.0;                                       \
    printf("mfma_f64 16x16x4: nacc=%d waves/CU=%d  %.3f ms  %.2f TFLOP/s  (%.1f cyc/mfma/SIMD @2.4GHz)\n", \
           NACC, WPC, ms, fl / ms * 1e-9, ms * 1e-3 * 2.4e9 / ((double)iters * NACC * ((WPC) / 4.0)));        \
  }
  RUN_MFMA(1, 4) RUN_MFMA(2, 4) RUN_MFMA(4, 4) RUN_MFMA(16, 4) RUN_MFMA(4, 8) RUN_MFMA(16, 8)
#define RUN_VALU(NACC, WPC)                                                                       \
  {                                                                                               \
    int blocks = ncu * (WPC) / 4;                                                                 \
    valu_rate<NACC><<<blocks, 256>>>(out, 100, 1.0);                                              \
    CHECK(hipDeviceSynchronize());                                                                \
    CHECK(hipEventRecord(e0));                                                                    \
    valu_rate<NACC><<<blocks, 256>>>(out, iters * 4, 1.0);                                        \
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));                                    \
    CHECK(hipEventElapsedTime(&ms, e0, e1));                                                      \
    double fl = (double)blocks * 256 * (iters * 4.0) * NACC * 2.0;                                \
    printf("v_fma_f64: nacc=%d waves/CU=%d  %.3f ms  %.2f TFLOP/s\n", NACC, WPC, ms, fl / ms * 1e-9); \
  }
  RUN_VALU(8, 4) RUN_VALU(16, 8) RUN_VALU(16, 16)
  {
    hybrid_rate<<<ncu, 512>>>(out, 100, 1.0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hybrid_rate<<<ncu, 512>>>(out, iters, 1.0);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double fl_m = (double)ncu * 4 * iters * 8 * 2048.0;
    double fl_v = (double)ncu * 256 * (iters * 8.0) * 16 * 2.0;
    printf("hybrid (4 mfma waves + 4 valu waves per CU): %.3f ms  mfma %.2f + valu %.2f = %.2f TFLOP/s\n", ms,
           fl_m / ms * 1e-9, fl_v / ms * 1e-9, (fl_m + fl_v) / ms * 1e-9);
  }
  return 0;
}
