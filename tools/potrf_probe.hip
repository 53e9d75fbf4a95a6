// Diagnostic (not product code): potrf64_wave alone -- one wave factorising a 64 x 64 tile held in LDS, as the diagonal task of the
// 64-tile dataflow kernel does -- timed per 16-column sub-block (s_memtime cycles), with and without the piece flags being published.
#define GPG_POTRF_STAMP
#include "../gpgradpy_amd/csrc/cholesky.hip"
#include "../gpgradpy_amd/csrc/cholesky_dataflow.hip"
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
void gpg_prof_begin(gpg_ctx*, int, double) {}
void gpg_prof_end(gpg_ctx*) {}
void gpg_launch_identity(gpg_ctx*, double*, int) {}
namespace {
__global__ void __launch_bounds__(256, 2) potrf_probe_kernel(const double* src, double* blk, double* dinv, int* flags, int publish, int reps,
                                                             unsigned long long* stamps, int wg) {
  __shared__ __attribute__((aligned(16))) double U[64 * 80];
  __shared__ __attribute__((aligned(16))) double St[64][64];
  const int tid = threadIdx.x;
  for (int rep = 0; rep < reps; ++rep) {
    for (int t = tid; t < 64 * 64; t += 256) U[(t >> 6) * 80 + (t & 63)] = src[t];     // U[col * 80 + row]
    __syncthreads();
    if (tid == 0) g_potrf_stamp = stamps + (size_t)rep * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wg) potrf64_wg(U, 80, St, blk + (size_t)blockIdx.x * 64 * 64, 64, dinv + blockIdx.x * 64, publish ? flags + 4 * blockIdx.x : nullptr);
    else if (tid < 64) potrf64_wave(U, 80, St, blk + (size_t)blockIdx.x * 64 * 64, 64, dinv + blockIdx.x * 64, publish ? flags + 4 * blockIdx.x : nullptr);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) { stamps[(size_t)rep * 16 + 8] = t0; stamps[(size_t)rep * 16 + 9] = t1; }
    __syncthreads();
  }
}
}  // namespace
int main() {
  const int T = 64;
  std::vector<double> h((size_t)T * T);
  srand(3);
  for (int c = 0; c < T; ++c)
    for (int r = 0; r < T; ++r) h[r + (size_t)c * T] = r == c ? 2.0 : 0.01 * ((rand() & 0xff) / 256.0 - 0.5);
  for (int c = 0; c < T; ++c)
    for (int r = 0; r < c; ++r) h[r + (size_t)c * T] = h[c + (size_t)r * T];
  double *src, *blk, *dinv; int* flags; unsigned long long* st;
  const int reps = 50;
  hipMalloc(&src, 8 * h.size()); hipMalloc(&blk, 8 * h.size() * 4); hipMalloc(&dinv, 8 * 64 * 4); hipMalloc(&flags, 64); hipMalloc(&st, 8 * 16 * reps);
  hipMemcpy(src, h.data(), 8 * h.size(), hipMemcpyHostToDevice);
  for (int wg = 0; wg < 2; ++wg)
  for (int publish = 0; publish < 2; ++publish) {
    hipMemset(st, 0, 8 * 16 * reps); hipMemset(flags, 0, 64);
    hipLaunchKernelGGL(potrf_probe_kernel, dim3(1), dim3(256), 0, 0, src, blk, dinv, flags, publish, reps, st, wg);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<unsigned long long> hs(16 * reps);
    hipMemcpy(hs.data(), st, 8 * hs.size(), hipMemcpyDeviceToHost);
    double tot = 0, ph[8] = {0};
    for (int r = 10; r < reps; ++r) {
      const unsigned long long* o = &hs[(size_t)r * 16];
      tot += (double)(o[9] - o[8]);
      for (int k = 0; k < 8; ++k) ph[k] += (double)(o[k] - (k == 0 ? o[8] : o[k - 1]));
    }
    const int n = reps - 10;
    // check the factor of the last repetition: L L^T = A on the lower triangle
    {
      std::vector<double> L((size_t)T * T);
      hipMemcpy(L.data(), blk, 8 * L.size(), hipMemcpyDeviceToHost);
      double err = 0;
      for (int r = 0; r < T; ++r)
        for (int c = 0; c <= r; ++c) {
          double sacc = 0;
          for (int k = 0; k <= c; ++k) sacc += L[r + (size_t)k * T] * L[c + (size_t)k * T];
          err = fmax(err, fabs(sacc - h[r + (size_t)c * T]));
        }
      printf("%s max |L L^T - A| = %.2e; ", wg ? "potrf64_wg  " : "potrf64_wave", err);
    }
    printf("publish=%d: potrf64_wave %.0f cycles (%.2f us at 2.4 GHz); per sub-block [update+load | pivots+store+publish]:", publish, tot / n, tot / n / 2400.0);
    for (int s = 0; s < 4; ++s) printf("  s%d %.0f | %.0f", s, ph[2 * s] / n, ph[2 * s + 1] / n);
    printf("\n");
  }
  return 0;
}
