// Diagnostic (not product code): per-CU throughput of the two ways of bringing an L2-resident 16 KiB chunk into
// a CU: LDS-DMA (global_load_lds_dwordx4) vs plain global_load_dwordx4 into registers (+ ds_write_b128).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
struct Stamp { unsigned long long c0, c1; };

template <int MODE>   // 0: LDS-DMA, 1: registers + ds_write, 2: registers only (no LDS)
__global__ void __launch_bounds__(256) loader(const double* __restrict__ src, size_t stride_wg, int iters, double* out, Stamp* st) {
  __shared__ __attribute__((aligned(16))) double smem[4 * 2304];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const double* g = src + (size_t)blockIdx.x * stride_wg + 2 * lane;
  double2 acc = {0, 0};
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int stage = it & 3;
    double* sa = smem + stage * 2304 + w * 144;
    const double* gp = g + (size_t)(it & 15) * 2048;      // 16 chunks of 16 KiB = 256 KiB per workgroup, L2 resident
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(gp + (w + 4 * q) * 128), (lds_ptr_t)(sa + 4 * q * 144), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      double2 r[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] = *reinterpret_cast<const double2*>(gp + (w + 4 * q) * 128);
      if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<double2*>(sa + 4 * q * 144 + 2 * lane) = r[q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc.x += r[q].x; acc.y += r[q].y; }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  out[blockIdx.x * 256 + tid] = smem[tid] + acc.x + acc.y;
  if (tid == 0) st[blockIdx.x] = Stamp{c0, c1};
}

template <int MODE>
void run(const char* name, const double* src, int nblk, size_t stride, double* out, Stamp* dst) {
  const int iters = 4096;
  loader<MODE><<<nblk, 256>>>(src, stride, 64, out, dst);
  hipDeviceSynchronize();
  loader<MODE><<<nblk, 256>>>(src, stride, iters, out, dst);
  hipDeviceSynchronize();
  std::vector<Stamp> st(nblk);
  hipMemcpy(st.data(), dst, nblk * sizeof(Stamp), hipMemcpyDeviceToHost);
  std::vector<double> v;
  for (auto& s : st) v.push_back(double(s.c1 - s.c0));
  std::sort(v.begin(), v.end());
  const double cyc = v[v.size() / 2];
  printf("%-34s blocks/CU=%d: %.1f cycles per 16 KiB chunk per workgroup = %.1f B/clk per workgroup\n", name, nblk / 256, cyc / iters,
         16384.0 * iters / cyc);
}
int main() {
  const size_t stride = 16 * 2048;            // doubles per workgroup region (256 KiB)
  double *src, *out; Stamp* dst;
  hipMalloc(&src, sizeof(double) * stride * 512);
  hipMemset(src, 0, sizeof(double) * stride * 512);
  hipMalloc(&out, 8 * 512 * 256); hipMalloc(&dst, 512 * sizeof(Stamp));
  for (int nblk : {256, 512}) {
    run<0>("LDS-DMA global_load_lds_dwordx4", src, nblk, stride, out, dst);
    run<1>("global_load_dwordx4 + ds_write_b128", src, nblk, stride, out, dst);
    run<2>("global_load_dwordx4 (registers only)", src, nblk, stride, out, dst);
  }
  // same 256 KiB for every workgroup (stride 0): hits in L2 for sure
  printf("-- every workgroup reads the same 256 KiB --\n");
  run<0>("LDS-DMA global_load_lds_dwordx4", src, 512, 0, out, dst);
  run<2>("global_load_dwordx4 (registers only)", src, 512, 0, out, dst);
  return 0;
}
