// Diagnostic probe (not product code): fp64 MFMA vs fp64 VALU pipes on gfx950, with in-kernel clocks.
// Reports cycles per instruction (s_memtime), the clock the chip holds (s_memtime / s_memrealtime)
// and whether MFMA-f64 waves and VALU-f64 waves co-issue on one SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Stamp { unsigned long long c0, c1, r0, r1; };

__device__ inline void mfma_body(int iters, double a, double b, double& sink) {
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  for (int i = 0; i < 8; ++i) sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
}
__device__ inline void valu_body(int iters, double a, double b, double& sink) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  for (int i = 0; i < 16; ++i) sink += acc[i];
}

// block = (NM + NV) * 4 waves; waves [0, 4*NM) run MFMA, the rest run VALU FMA.
__global__ void __launch_bounds__(1024) pipes(double* out, Stamp* st, int nm_waves, int it_m, int it_v, double seed) {
  int w = threadIdx.x >> 6;
  double s = 0;
  double a = seed + threadIdx.x * 1e-9, b = 1e-9 * seed;
  __syncthreads();
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (w < nm_waves) mfma_body(it_m, a, b, s); else valu_body(it_v, a, b, s);
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 16 + w] = Stamp{c0, c1, r0, r1};
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int ncu = prop.multiProcessorCount;
  double* out; CHECK(hipMalloc(&out, (size_t)ncu * 1024 * 8));
  Stamp* dst; CHECK(hipMalloc(&dst, (size_t)ncu * 16 * sizeof(Stamp)));
  std::vector<Stamp> st(ncu * 16);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  struct Cfg { int nm, nv, it_m, it_v; };
  // per SIMD: nm MFMA waves, nv VALU waves (block has 4*(nm+nv) waves, one block per CU)
  std::vector<Cfg> cfgs = {
      {1, 0, 40000, 0}, {2, 0, 40000, 0}, {0, 1, 0, 400000}, {0, 2, 0, 400000}, {0, 4, 0, 200000},
      {1, 1, 40000, 400000}, {1, 2, 40000, 300000}, {1, 3, 40000, 200000}, {2, 2, 20000, 300000},
  };
  for (auto c : cfgs) {
    int waves = 4 * (c.nm + c.nv);
    // warm the clocks
    pipes<<<ncu, waves * 64>>>(out, dst, 4 * c.nm, c.it_m / 4, c.it_v / 4, 1.0);
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventRecord(e0));
    pipes<<<ncu, waves * 64>>>(out, dst, 4 * c.nm, c.it_m, c.it_v, 1.0);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(st.data(), dst, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    // per-role medians
    std::vector<double> cyc_m, cyc_v, t_m, t_v, clk;
    for (int bId = 0; bId < ncu; ++bId) for (int w = 0; w < waves; ++w) {
      Stamp& s = st[bId * 16 + w];
      double dc = double(s.c1 - s.c0), dr = double(s.r1 - s.r0) * 10e-9;  // 100 MHz ticks -> s
      clk.push_back(dc / dr);
      if (w < 4 * c.nm) { cyc_m.push_back(dc / (double(c.it_m) * 8)); t_m.push_back(dr); }
      else { cyc_v.push_back(dc / (double(c.it_v) * 16)); t_v.push_back(dr); }
    }
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    double tm = med(t_m), tv = med(t_v);
    double fl_m = (double)ncu * 4 * c.nm * c.it_m * 8 * 2048.0, fl_v = (double)ncu * 4 * c.nv * 64 * (double)c.it_v * 16 * 2.0;
    printf("per-SIMD waves: mfma=%d valu=%d | wall %.2f ms | clock %.3f GHz | mfma: %.1f cyc/inst, role time %.2f ms, %.2f TF | valu: %.2f cyc/inst, role time %.2f ms, %.2f TF | sum-at-overlap %.2f TF\n",
           c.nm, c.nv, ms, med(clk) * 1e-9, med(cyc_m), tm * 1e3, tm > 0 ? fl_m / tm * 1e-12 : 0.0, med(cyc_v), tv * 1e3,
           tv > 0 ? fl_v / tv * 1e-12 : 0.0, (tm > 0 ? fl_m / tm : 0.0) * 1e-12 + (tv > 0 ? fl_v / tv : 0.0) * 1e-12);
  }
  return 0;
}
