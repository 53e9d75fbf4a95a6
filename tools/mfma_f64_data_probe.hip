// Diagnostic (not product code): does the fp64 MFMA rate / clock depend on operand data?
// Runs a register-only v_mfma_f64_16x16x4_f64 loop (one or two waves per SIMD) for ~1 s per data
// pattern and reports cycles/MFMA (s_memtime), the held clock (s_memtime / s_memrealtime) and TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long c0, c1, r0, r1; };
__global__ void __launch_bounds__(512) mfma_loop(const double* __restrict__ av, const double* __restrict__ bv, double* out,
                                                 Stamp* st, int iters, double decay) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = av[t * 4 + i]; b[i] = bv[t * 4 + i]; }
  d4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[ni * 4 + mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ni], b[mi], acc[ni * 4 + mi], 0, 0, 0);
    // rotate operands a little so that products change sign / magnitude like a real GEMM k-loop
    double t0 = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = -t0 * decay;
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[t] = s;
  if ((threadIdx.x & 63) == 0) st[t >> 6] = Stamp{c0, c1, r0, r1};
}
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  for (int wps = 1; wps <= 2; ++wps) {
    const int threads = 256 * wps, nthr = ncu * threads, nwave = nthr / 64;
    double *av, *bv, *out; Stamp* dst;
    hipMalloc(&av, nthr * 4 * 8); hipMalloc(&bv, nthr * 4 * 8); hipMalloc(&out, nthr * 8); hipMalloc(&dst, nwave * sizeof(Stamp));
    std::vector<double> ha(nthr * 4), hb(nthr * 4);
    const char* names[3] = {"zeros", "ones", "random[-1,1)"};
    for (int pat = 0; pat < 3; ++pat) {
      srand(7);
      for (size_t i = 0; i < ha.size(); ++i) {
        ha[i] = pat == 0 ? 0.0 : pat == 1 ? 1.0 : 2.0 * rand() / RAND_MAX - 1.0;
        hb[i] = pat == 0 ? 0.0 : pat == 1 ? 1.0 : 2.0 * rand() / RAND_MAX - 1.0;
      }
      hipMemcpy(av, ha.data(), ha.size() * 8, hipMemcpyHostToDevice);
      hipMemcpy(bv, hb.data(), hb.size() * 8, hipMemcpyHostToDevice);
      const int iters = 600000 / wps;   // ~0.3-0.4 s per launch
      std::vector<Stamp> st(nwave);
      double tf = 0, clk = 0, cyc = 0;
      for (int rep = 0; rep < 4; ++rep) {   // back-to-back: ~1.5 s of load, report the last launch
        mfma_loop<<<ncu, threads>>>(av, bv, out, dst, iters, 0.999);
        hipDeviceSynchronize();
      }
      hipMemcpy(st.data(), dst, nwave * sizeof(Stamp), hipMemcpyDeviceToHost);
      std::vector<double> vc, vk, vt;
      for (auto& s : st) {
        double dc = double(s.c1 - s.c0), dr = double(s.r1 - s.r0) * 10e-9;
        vc.push_back(dc / (double(iters) * 16)); vk.push_back(dc / dr); vt.push_back(dr);
      }
      std::sort(vc.begin(), vc.end()); std::sort(vk.begin(), vk.end()); std::sort(vt.begin(), vt.end());
      cyc = vc[vc.size() / 2]; clk = vk[vk.size() / 2];
      tf = (double)nwave * iters * 16 * 2048.0 / vt[vt.size() / 2] * 1e-12;
      printf("waves/SIMD=%d data=%-13s: %.1f cyc/MFMA/wave, clock %.3f GHz, %.1f TFLOP/s\n", wps, names[pat], cyc, clk * 1e-9, tf);
    }
    hipFree(av); hipFree(bv); hipFree(out); hipFree(dst);
  }
  return 0;
}
