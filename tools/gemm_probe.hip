// Diagnostic (not product code): times the trailing-update kernel alone on a square lower update and
// ablates its phases (build with -DGPG_ABLATE_CLOAD / -DGPG_ABLATE_DMA).
#include "../gpgradpy_amd/csrc/cholesky.hip"
#include <cstdio>
#include <cstdlib>
void gpg_prof_begin(gpg_ctx*, int, double) {}
void gpg_prof_end(gpg_ctx*) {}
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
  for (int rep = 0; rep < 3; ++rep) launch_gemm_trailing(&c, C, ld, P, ld, P, ld, ld, Nt, K, 0);
  hipStreamSynchronize(c.stream);
  const int reps = 10;
  hipEventRecord(e0, c.stream);
  for (int rep = 0; rep < reps; ++rep) launch_gemm_trailing(&c, C, ld, P, ld, P, ld, ld, Nt, K, 0);
  hipEventRecord(e1, c.stream);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)Nt * (Nt + 1.0) * K * reps;
  printf("Nt=%d K=%d impl=%d: %.3f ms/launch, %.2f TFLOP/s (algorithmic, lower triangle)\n", Nt, K, impl, ms / reps, fl / ms * 1e-9);
  return 0;
}
