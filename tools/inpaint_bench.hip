// Standalone A/B harness for the fused (inpainting data term + last DDIM-step VJP) kernel (not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/inpaint_bench.hip -o tools/inpaint_bench && tools/inpaint_bench
// Includes the product's kernel file, runs the round-1 form k_mix_bwd_inpaint<true> and k_mix_bwd_inpaint_px<VPT> for
// several VPT on B = 64 chains of 3 x 256 x 256 with a random whole-pixel mask (8 % kept), checks g_xt / g_e bit for bit
// against the round-1 form and prints the median launch time over buffer sets larger than the Infinity Cache.
#include "../noise-space-hmc_amd/csrc/ddim_mix.hip"
// the two tile-count helpers ddim_mix.hip references live in other product files
extern "C" int nhmc_leapfrog_tiles(int64_t n_elem) { return (int)((n_elem + NHMC_TILE - 1) / NHMC_TILE); }
extern "C" int nhmc_sr_tiles(int, int, int) { return 0; }
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Set { float *xt, *e, *gx, *ge; };

int main() {
  const int B = 64, C = 3, DIM = 256;
  const int64_t hw = (int64_t)DIM * DIM, N = C * hw, n4 = N / 4, hw4 = hw / 4;
  const int R = 4;
  std::vector<float> h(N * B), he(2 * N * B);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 8388608.0f - 1.0f; };
  for (auto& v : h) v = rnd();
  for (auto& v : he) v = 2.0f * rnd();
  std::vector<uint32_t> words(hw / 32);
  std::vector<int32_t> prefix(hw / 32);
  int kept = 0;
  for (size_t w = 0; w < words.size(); ++w) {
    prefix[w] = kept;
    uint32_t bits = 0;
    for (int b = 0; b < 32; ++b) if (rnd() > 0.84f) { bits |= 1u << b; ++kept; }
    words[w] = bits;
  }
  const int64_t M = (int64_t)kept * C;
  std::vector<float> hy(M * B);
  for (auto& v : hy) v = rnd();
  printf("kept pixels %d of %ld (M = %ld)\n", kept, (long)hw, (long)M);
  std::vector<Set> sets(R);
  for (auto& t : sets) {
    CK(hipMalloc(&t.xt, N * B * 4)); CK(hipMalloc(&t.e, 2 * N * B * 4)); CK(hipMalloc(&t.gx, N * B * 4)); CK(hipMalloc(&t.ge, 2 * N * B * 4));
    CK(hipMemcpy(t.xt, h.data(), N * B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(t.e, he.data(), 2 * N * B * 4, hipMemcpyHostToDevice));
    CK(hipMemset(t.ge, 0, 2 * N * B * 4));
  }
  float *y, *at, *atn; uint32_t* dw; int32_t* dp; double* ws;
  CK(hipMalloc(&y, M * B * 4)); CK(hipMemcpy(y, hy.data(), M * B * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dw, words.size() * 4)); CK(hipMemcpy(dw, words.data(), words.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dp, prefix.size() * 4)); CK(hipMemcpy(dp, prefix.data(), prefix.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> ha(B, 0.5214230418f), hn(B, 1.0f);
  CK(hipMalloc(&at, B * 4)); CK(hipMalloc(&atn, B * 4));
  CK(hipMemcpy(at, ha.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(atn, hn.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&ws, 8 * 4096 * B));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = 4.0 * N * B * 4;
  const int LAUNCH = 80, ROUNDS = 5;

  auto old_form = [&](Set& t) {
    dim3 grid((unsigned)nhmc_leapfrog_tiles(N), B);
    hipLaunchKernelGGL(k_mix_bwd_inpaint<true>, grid, dim3(NHMC_BLOCK), 0, 0, (const float4*)t.xt, (const float4*)t.e, 2 * n4, at, atn, y,
                       (const int4*)nullptr, dw, dp, C, hw, M, (float4*)t.gx, (float4*)t.ge, ws, n4, 0);
  };
  std::vector<float> ref_gx(N * B), ref_ge(2 * N * B), got(2 * N * B);
  old_form(sets[0]); CK(hipDeviceSynchronize());
  CK(hipMemcpy(ref_gx.data(), sets[0].gx, N * B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ref_ge.data(), sets[0].ge, 2 * N * B * 4, hipMemcpyDeviceToHost));

  auto bench = [&](const char* name, auto launch, bool check) {
    if (check) {
      CK(hipMemset(sets[0].gx, 0xff, N * B * 4));
      launch(sets[0]); CK(hipDeviceSynchronize());
      CK(hipMemcpy(got.data(), sets[0].gx, N * B * 4, hipMemcpyDeviceToHost));
      const bool okx = memcmp(got.data(), ref_gx.data(), N * B * 4) == 0;
      CK(hipMemcpy(got.data(), sets[0].ge, 2 * N * B * 4, hipMemcpyDeviceToHost));
      const bool oke = memcmp(got.data(), ref_ge.data(), 2 * N * B * 4) == 0;
      if (!okx || !oke) printf("  !! %s differs from the round-1 form (g_xt %d, g_e %d)\n", name, okx, oke);
    }
    std::vector<float> best;
    for (int r = 0; r < ROUNDS; ++r) {
      for (int i = 0; i < R; ++i) launch(sets[i]);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < LAUNCH; ++i) launch(sets[i % R]);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best.push_back(ms * 1e3f / LAUNCH);
    }
    std::sort(best.begin(), best.end());
    printf("%-34s min %.2f us  med %.2f us  -> %.0f GB/s (med, 4T)\n", name, best[0], best[ROUNDS / 2], bytes / (best[ROUNDS / 2] * 1e-6) / 1e9);
    fflush(stdout);
  };
#define PXV(V) bench("px vpt" #V, [&](Set& t) { \
    dim3 grid((unsigned)((hw4 + NHMC_BLOCK * (V) - 1) / (NHMC_BLOCK * (V))), C, B); \
    hipLaunchKernelGGL(k_mix_bwd_inpaint_px<V>, grid, dim3(NHMC_BLOCK), 0, 0, (const float4*)t.xt, (const float4*)t.e, 2 * n4, at, atn, y, \
                       dw, dp, C, hw4, M, (float4*)t.gx, (float4*)t.ge, ws, 0); }, true)
  for (int rep = 0; rep < 2; ++rep) {
    bench("round-1 form (slotless, vpt2)", old_form, false);
    PXV(1); PXV(2); PXV(4); PXV(8);
  }
  return 0;
}
