// Diagnostic (not product code): runs the dataflow Cholesky kernels alone on a synthetic SPD matrix; with
// -DGPG_STAMP writes a per-task timeline (start, end, spin / MFMA cycles, runs) to gpurun_out/tile_timeline.csv.
#include "../gpgradpy_amd/csrc/cholesky.hip"
#include "../gpgradpy_amd/csrc/cholesky_dataflow.hip"
#include <cstdio>
#include <cstdlib>
void gpg_prof_begin(gpg_ctx*, int, double) {}
void gpg_prof_end(gpg_ctx*) {}
void gpg_launch_identity(gpg_ctx*, double*, int) {}
__global__ void fill_spd(double* A, int ld, int n) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)ld * n;
  if (idx >= total) return;
  const int r = idx % ld, c = idx / ld;
  unsigned h = (unsigned)(r * 2654435761u) ^ (unsigned)(c * 40503u + 12345u);
  h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
  const int lo = r < c ? r : c, hi = r < c ? c : r;
  unsigned g = (unsigned)(lo * 2654435761u) ^ (unsigned)(hi * 40503u + 12345u);
  g ^= g >> 13; g *= 0x5bd1e995u; g ^= g >> 15;
  double v = ((g & 0xffff) / 65536.0 - 0.5) * 1e-3;
  if (r == c) v = 2.0;
  if (r >= n) v = ((h & 0xffff) / 65536.0 - 0.5);
  A[idx] = v;
}
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 18048, impl = argc > 2 ? atoi(argv[2]) : 1;
  gpg_ctx c;
  hipStreamCreate(&c.stream);
  c.stream_upd = c.stream; c.lookahead = 0;
  c.Npad = n; c.N = n; c.ld = n + 128; c.nb_outer = 512;
  c.chol_impl = impl == 1; c.tail_cols = impl == 2 ? n : 0;
  if (getenv("GPG_PROBE_MAX_WG")) c.max_workgroups = atoi(getenv("GPG_PROBE_MAX_WG"));   // cap on the persistent grid (e.g. 256 = one workgroup per CU)
  c.num_cus = 256;
  if (getenv("GPG_PROBE_ORDER")) c.task_order = atoi(getenv("GPG_PROBE_ORDER"));
  if (getenv("GPG_PAIR")) c.pair_mode = atoi(getenv("GPG_PAIR"));
  if (getenv("GPG_PROBE_FUSE")) { c.fuse_subdiag = atoi(getenv("GPG_PROBE_FUSE")); c.fuse_subdiag_max_tiles = 1 << 20; }
  hipMalloc(&c.A, sizeof(double) * (size_t)c.ld * n);
  hipMalloc(&c.dinv, sizeof(double) * n);
  hipMalloc(&c.info, sizeof(int));
  hipMemset(c.info, 0, sizeof(int));
#ifdef GPG_STAMP
  unsigned long long* dbuf = nullptr;
  if (hipMalloc(&dbuf, GPG_STAMP_MAX * 32 * 8) != hipSuccess || dbuf == nullptr) { printf("stamp buffer alloc failed\n"); return 1; }
  hipMemset(dbuf, 0, GPG_STAMP_MAX * 32 * 8);
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &dbuf, sizeof(dbuf)) != hipSuccess) { printf("symbol copy failed\n"); return 1; }
  hipDeviceSynchronize();
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t total = (size_t)c.ld * n;
  const int B = (impl == 5 || impl == 6) ? (argc > 3 ? atoi(argv[3]) : 8) : 1;   // 5: batched 128-tile kernel, 6: batched 64-tile kernel
  double* bA = nullptr; double* bD = nullptr; int* binfo = nullptr;
  if (B > 1) { hipMalloc(&bA, sizeof(double) * (size_t)c.ld * n * B); hipMalloc(&bD, sizeof(double) * (size_t)n * B); hipMalloc(&binfo, sizeof(int) * B); hipMemset(binfo, 0, sizeof(int) * B); c.chol_impl = 1; c.tail_cols = impl == 6 ? n : 0; }
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    if (B > 1) for (int b = 0; b < B; ++b) hipLaunchKernelGGL(fill_spd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.stream, bA + (size_t)b * total, c.ld, n);
    else hipLaunchKernelGGL(fill_spd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.stream, c.A, c.ld, n);
    hipEventRecord(e0, c.stream);
    if (B > 1) gpg_launch_tile_chol_batch(&c, B, bA, total, bD, n, binfo);
    else gpg_cholesky(&c);
    hipEventRecord(e1, c.stream);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  int info = -1, ab = -1;
  hipMemcpy(&info, c.info, sizeof(int), hipMemcpyDeviceToHost);
  if (c.tile_flags) {   // abort word: after the Mt x Rt tile flags
    const int T = (impl == 1 || impl == 5) ? 128 : 64;
    hipMemcpy(&ab, c.tile_flags + (size_t)(c.Npad / T) * (c.ld / T), sizeof(int), hipMemcpyDeviceToHost);
  }
  printf("n=%d impl=%d: %.3f ms, %.2f TFLOP/s (n^3/3), info=%d abort=%d\n", n, impl, best, (double)B * n * n * n / 3.0 / best * 1e-9, info, ab);
#ifdef GPG_STAMP
  {
    std::vector<unsigned long long> hb(GPG_STAMP_MAX * 32);
    hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost);
    FILE* f = fopen(argc > 4 ? argv[4] : "../gpurun_out/tile_timeline.csv", "w");
    if (f) {
      fprintf(f, "block,ti,tj,start,end,spin_cyc,gemm_cyc,runs,fin0,f0,f1,f2,f3,f4,wg,f5,f6,f7,p0w,p0i,p0s,p1w,p1i,p1s,p2w,p2i,p2s,p3w,p3i,p3s,x12,x13,x14,finwait\n");
      unsigned long long t0 = ~0ull;
      for (int b = 0; b < GPG_STAMP_MAX; ++b) if (hb[b * 8 + 1] && hb[b * 8] < t0) t0 = hb[b * 8];
      for (int b = 0; b < GPG_STAMP_MAX; ++b) {
        const unsigned long long* o = &hb[(size_t)b * 8];
        if (o[1] == 0) continue;
        const unsigned long long* g = &hb[(size_t)GPG_STAMP_MAX * 8 + (size_t)b * 8];
        fprintf(f, "%d,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu\n", b, o[6] & 0xffff, o[6] >> 16, o[0] - t0, o[1] - t0, o[2], o[3], o[4], o[5] - t0, g[0] - t0, g[1] - t0, g[2] - t0, g[3] - t0, g[4] ? g[4] - t0 : 0ull, o[7], g[5] ? g[5] - t0 : 0ull, g[6] ? g[6] - t0 : 0ull, g[7] > 16 ? g[7] - t0 : g[7]);
        const unsigned long long* pz = &hb[(size_t)GPG_STAMP_MAX * 16 + (size_t)b * 16];
        fseek(f, -1, SEEK_CUR);
        for (int k = 0; k < 12; ++k) fprintf(f, ",%llu", pz[k] ? pz[k] - t0 : 0ull);
        for (int k = 12; k < 16; ++k) fprintf(f, ",%llu", pz[k]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
#endif
  return 0;
}
