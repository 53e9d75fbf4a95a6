// Diagnostic (not product code): can ONE wave interleave fp64 VALU FMAs between its fp64 MFMAs at no cost?
// Mimics a GEMM k-step: 16 independent MFMAs (64x64 wave tile) + NV independent VALU FMAs (an extra strip),
// two waves per SIMD.  Reports cycles per k-step and the combined flop rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long c0, c1, r0, r1; };

template <int NV, bool SCHED>
__global__ void __launch_bounds__(512) hybrid(double* out, Stamp* st, int iters, double seed) {
  d4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
  double vacc[NV > 0 ? NV : 1];
  for (int i = 0; i < NV; ++i) vacc[i] = i;
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = seed + threadIdx.x * 1e-3 + i; b[i] = seed - threadIdx.x * 1e-3 - i; }
  double va = seed * 1e-3 + threadIdx.x * 1e-6, vb = 1e-9;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        acc[ni * 4 + mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ni], b[mi], acc[ni * 4 + mi], 0, 0, 0);
        if (NV > 0) {
#pragma unroll
          for (int v = 0; v < NV / 16; ++v) {
            const int idx = (ni * 4 + mi) * (NV / 16) + v;
            vacc[idx] = __builtin_fma(vacc[idx], va, vb);
          }
          if (SCHED) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, NV / 16, 0);    // NV/16 VALU
          }
        }
      }
    }
    va += 1e-12;
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NV; ++i) s += vacc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{c0, c1, r0, r1};
}

template <int NV, bool SCHED>
void run(int ncu, int wps) {
  const int threads = 256 * wps, nthr = ncu * threads, nwave = nthr / 64;
  double* out; Stamp* dst;
  hipMalloc(&out, nthr * 8); hipMalloc(&dst, nwave * sizeof(Stamp));
  const int iters = 200000 / wps;
  for (int rep = 0; rep < 3; ++rep) { hybrid<NV, SCHED><<<ncu, threads>>>(out, dst, iters, 1.0); hipDeviceSynchronize(); }
  std::vector<Stamp> st(nwave);
  hipMemcpy(st.data(), dst, nwave * sizeof(Stamp), hipMemcpyDeviceToHost);
  std::vector<double> vc, vt;
  for (auto& s : st) { vc.push_back(double(s.c1 - s.c0) / iters); vt.push_back(double(s.r1 - s.r0) * 10e-9); }
  std::sort(vc.begin(), vc.end()); std::sort(vt.begin(), vt.end());
  const double t = vt[vt.size() / 2];
  const double fm = (double)nwave * iters * 16 * 2048.0, fv = (double)nwave * iters * NV * 128.0;
  printf("waves/SIMD=%d NV=%3d sched=%d: %.0f cyc per k-step per wave | MFMA %.1f + VALU %.1f = %.1f TFLOP/s\n", wps, NV, (int)SCHED,
         vc[vc.size() / 2], fm / t * 1e-12, fv / t * 1e-12, (fm + fv) / t * 1e-12);
  hipFree(out); hipFree(dst);
}
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  for (int wps = 1; wps <= 2; ++wps) {
    run<0, false>(ncu, wps);
    run<32, false>(ncu, wps); run<32, true>(ncu, wps);
    run<64, false>(ncu, wps); run<64, true>(ncu, wps);
    run<128, false>(ncu, wps); run<128, true>(ncu, wps);
  }
  return 0;
}
