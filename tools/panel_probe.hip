// Diagnostic (not product code): times the fused B_p kernel (panel_solve_kernel) alone and, built with
// -DGPG_STAMP, prints the cycle shares of its phases.
#include "../gpgradpy_amd/csrc/cholesky.hip"
#include "../gpgradpy_amd/csrc/cholesky_dataflow.hip"
#include <cstdio>
#include <cstdlib>
void gpg_prof_begin(gpg_ctx*, int, double) {}
void gpg_prof_end(gpg_ctx*) {}
void gpg_launch_identity(gpg_ctx*, double*, int) {}
int main(int argc, char** argv) {
  int rows = argc > 1 ? atoi(argv[1]) : 16384, nb = argc > 2 ? atoi(argv[2]) : 512, impl = argc > 3 ? atoi(argv[3]) : 1;
  gpg_ctx c;
  c.panel_impl = impl;
  hipStreamCreate(&c.stream);
  const int ld = rows + nb;
  c.ld = ld; c.Npad = nb; c.N = nb;
  double *A, *dinv;
  hipMalloc(&A, sizeof(double) * (size_t)ld * nb);
  hipMalloc(&dinv, sizeof(double) * nb);
  c.A = A; c.dinv = dinv;
  std::vector<double> h((size_t)ld * nb), hd(nb);
  srand(1);
  for (auto& v : h) v = (rand() / (double)RAND_MAX - 0.5) * 1e-2;
  for (int i = 0; i < nb; ++i) { h[i + (size_t)i * ld] = 1.0 + 0.01 * i; hd[i] = 1.0 / h[i + (size_t)i * ld]; }
  hipMemcpy(A, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  hipMemcpy(dinv, hd.data(), sizeof(double) * nb, hipMemcpyHostToDevice);
#ifdef GPG_STAMP
  unsigned long long* dbuf = nullptr;
  if (hipMalloc(&dbuf, 4096 * 32 * 8) != hipSuccess || dbuf == nullptr) { printf("stamp buffer alloc failed\n"); return 1; }
  hipMemset(dbuf, 0, 4096 * 32 * 8);
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &dbuf, sizeof(dbuf)) != hipSuccess) { printf("symbol copy failed\n"); return 1; }
  hipDeviceSynchronize();
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) solve_below_block(&c, 0, nb);
  hipStreamSynchronize(c.stream);
  const int reps = 10;
  hipEventRecord(e0, c.stream);
  for (int rep = 0; rep < reps; ++rep) solve_below_block(&c, 0, nb);
  hipEventRecord(e1, c.stream);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
#ifdef GPG_STAMP
  {
    std::vector<unsigned long long> hb(4096 * 32);
    hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost);
    double sum[5] = {0, 0, 0, 0, 0}; int nw = 0;
    for (int b = 0; b < 4096; ++b) for (int w = 0; w < 4; ++w) { const unsigned long long* o = &hb[(b * 4 + w) * 8]; if (o[3] == 0) continue; ++nw; for (int q = 0; q < 5; ++q) sum[q] += (double)o[q]; }
    printf("stamps over %d waves (cycles of the 100 MHz s_memtime clock x24 = core cycles?) per panel: pre %.0f gemm %.0f transpose %.0f subst %.0f store %.0f\n", nw,
           sum[0] / nw, sum[1] / nw, sum[2] / nw, sum[3] / nw, sum[4] / nw);
  }
#endif
  printf("rows=%d nb=%d impl=%d: %.1f us/panel, %.2f TFLOP/s (rows nb^2)\n", rows, nb, impl, ms / reps * 1e3,
         (double)rows * nb * nb * reps / ms * 1e-9);
  return 0;
}
