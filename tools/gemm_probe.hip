// Diagnostic (not product code): times the trailing-update kernel alone on a square lower update and
// ablates its phases (build with -DGPG_ABLATE_CLOAD / -DGPG_ABLATE_DMA).
#include "../gpgradpy_amd/csrc/cholesky.hip"
#include "../gpgradpy_amd/csrc/cholesky_dataflow.hip"
#include <cstdio>
#include <cstdlib>
void gpg_prof_begin(gpg_ctx*, int, double) {}
void gpg_prof_end(gpg_ctx*) {}
void gpg_launch_identity(gpg_ctx*, double*, int) {}
int main(int argc, char** argv) {
  int Nt = argc > 1 ? atoi(argv[1]) : 16384, K = argc > 2 ? atoi(argv[2]) : 256, impl = argc > 3 ? atoi(argv[3]) : 1;
  gpg_ctx c;
  c.gemm_impl = impl;
  hipStreamCreate(&c.stream);
  int ld = Nt + 128;
  double *C, *P;
  hipMalloc(&C, sizeof(double) * (size_t)ld * Nt);
  hipMalloc(&P, sizeof(double) * (size_t)ld * K);
  std::vector<double> h((size_t)ld * K);
  srand(1);
  for (auto& v : h) v = (rand() / (double)RAND_MAX - 0.5) * 1e-2;
  hipMemcpy(P, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  hipMemset(C, 0, sizeof(double) * (size_t)ld * Nt);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  {
    int nb = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_dma_kernel<4>, 256, 0);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)gemm_dma_kernel<4>);
    printf("gemm_dma_kernel: occupancy API %d blocks/CU, regs %d, static LDS %zu B\n", nb, fa.numRegs, fa.sharedSizeBytes);
  }
#ifdef GPG_STAMP
  unsigned long long* dbuf = nullptr;
  if (hipMalloc(&dbuf, 4096 * 24 * 8) != hipSuccess || dbuf == nullptr) { printf("stamp buffer alloc failed\n"); return 1; }
  hipMemset(dbuf, 0, 4096 * 24 * 8);
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &dbuf, sizeof(dbuf)) != hipSuccess) { printf("symbol copy failed\n"); return 1; }
  hipDeviceSynchronize();
#endif
  for (int rep = 0; rep < 3; ++rep) launch_gemm_trailing(&c, C, ld, P, ld, P, ld, ld, Nt, K, 0);
  hipStreamSynchronize(c.stream);
  const int reps = 10;
  hipEventRecord(e0, c.stream);
  for (int rep = 0; rep < reps; ++rep) launch_gemm_trailing(&c, C, ld, P, ld, P, ld, ld, Nt, K, 0);
  hipEventRecord(e1, c.stream);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)Nt * (Nt + 1.0) * K * reps;
#ifdef GPG_STAMP
  {
    hipMemset(dbuf, 0, 4096 * 16 * 8);
    launch_gemm_trailing(&c, C, ld, P, ld, P, ld, ld, Nt, K, 0);
    hipStreamSynchronize(c.stream);
    std::vector<unsigned long long> hb(4096 * 16);
    hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost);
    double sum[4] = {0, 0, 0, 0}; int nw = 0;
    for (int b = 0; b < 4096; ++b) for (int w = 0; w < 4; ++w) { const unsigned long long* o = &hb[(b * 4 + w) * 4]; if (o[1] == 0) continue; ++nw; for (int q = 0; q < 4; ++q) sum[q] += (double)o[q]; }
    {
      // per-CU timeline: do the C phases of co-resident workgroups coincide?
      std::vector<unsigned long long> tl(4096 * 8);
      hipMemcpy(tl.data(), dbuf + 4096 * 16, tl.size() * 8, hipMemcpyDeviceToHost);
      FILE* f = fopen("../gpurun_out/gemm_timeline.csv", "w");
      if (f) {
        fprintf(f, "block,xcc,hwid,start,main_start,main_end,end\n");
        for (int b = 0; b < 4096; ++b) {
          const unsigned long long* o = &tl[(size_t)b * 8];
          if (o[1] == 0) continue;
          fprintf(f, "%d,%llu,%llu,%llu,%llu,%llu,%llu\n", b, o[0] >> 32, o[0] & 0xffffffffULL, o[1], o[2], o[3], o[4]);
        }
        fclose(f);
      }
    }
    printf("stamps over %d waves: issue %.0f + compute %.0f cycles per chunk; main loop %.0f cycles = %.2f us per tile -> held clock %.3f GHz\n", nw,
           sum[0] / nw / (K / 8), sum[1] / nw / (K / 8), sum[2] / nw, sum[3] / nw * 0.01, (sum[2] / nw) / (sum[3] / nw * 10.0));
  }
#endif
  printf("Nt=%d K=%d impl=%d: %.3f ms/launch, %.2f TFLOP/s (algorithmic, lower triangle)\n", Nt, K, impl, ms / reps, fl / ms * 1e-9);
  return 0;
}
