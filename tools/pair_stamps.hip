// Per-phase timing of k_pair256 (the spectral chain's two-products-per-launch kernel): the product kernel file compiled
// with NHMC_PAIR_STAMPS, thread 0 of every workgroup stamping clock64() at its phase boundaries.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/pair_stamps.hip -o tools/pair_stamps && tools/pair_stamps [chains] [warm-up launches]
// Prints, per phase, the mean / median / p90 cycles of a workgroup, split by the round the workgroup started in.
#define NHMC_PAIR_STAMPS 1
#include "../noise-space-hmc_amd/csrc/spectral_gemm.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int chains = argc > 1 ? std::atoi(argv[1]) : 64, C = 3, D = 256;
  const int n_img = chains * C, n_wg = 4 * n_img;
  const size_t img = (size_t)n_img * D * D;
  std::vector<float> h(img), f((size_t)D * D), dm((size_t)C * D * D);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f; };
  for (auto& v : h) v = rnd();
  for (auto& v : f) v = rnd() * 0.0625f;
  for (auto& v : dm) v = rnd();
  float *IN, *S1, *S2, *OUT, *DM;
  long long* ST;
  CK(hipMalloc(&IN, img * 4)); CK(hipMalloc(&OUT, img * 4)); CK(hipMalloc(&S1, f.size() * 4)); CK(hipMalloc(&S2, f.size() * 4));
  CK(hipMalloc(&DM, dm.size() * 4)); CK(hipMalloc(&ST, (size_t)n_wg * 8 * sizeof(long long)));
  CK(hipMemcpy(IN, h.data(), img * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(S1, f.data(), f.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(S2, f.data(), f.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(DM, dm.data(), dm.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(ST, 0, (size_t)n_wg * 8 * sizeof(long long)));
  long long* null_ptr = nullptr;
  const int warm = argc > 2 ? std::atoi(argv[2]) : 3;                // back-to-back launches without stamps first (300: ~40 ms,
  CK(hipMemcpyToSymbol(HIP_SYMBOL(nhmc_pair_stamps), &null_ptr, sizeof(long long*)));   // the chip at its loaded clock), then the stamped one
  for (int it = 0; it < warm; ++it)
    if (pair256<EPI_MULD, false>(IN, S1, S2, OUT, DM, nullptr, nullptr, n_img, C, nullptr)) { std::printf("launch failed\n"); return 1; }
  CK(hipMemcpyToSymbol(HIP_SYMBOL(nhmc_pair_stamps), &ST, sizeof(long long*)));
  if (pair256<EPI_MULD, false>(IN, S1, S2, OUT, DM, nullptr, nullptr, n_img, C, nullptr)) { std::printf("launch failed\n"); return 1; }
  CK(hipDeviceSynchronize());
  {                                                                  // the unstamped kernel's duration, by events
    CK(hipMemcpyToSymbol(HIP_SYMBOL(nhmc_pair_stamps), &null_ptr, sizeof(long long*)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, nullptr));
    for (int it = 0; it < 50; ++it)
      if (pair256<EPI_MULD, false>(IN, S1, S2, OUT, DM, nullptr, nullptr, n_img, C, nullptr)) { std::printf("launch failed\n"); return 1; }
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("50 back-to-back launches on the same buffers: %.1f us each\n", ms * 1000.f / 50);
    // the same over 8 input / output pairs (8 x 2 x 50 MB at 64 chains: more than the 256 MB memory-side cache holds), and
    // chained (each launch reads what the previous one wrote), as the product's four launches are
    constexpr int R = 8;
    float* buf[R + 1];
    for (int r = 0; r <= R; ++r) { CK(hipMalloc(&buf[r], img * 4)); CK(hipMemcpy(buf[r], IN, img * 4, hipMemcpyDeviceToDevice)); }
    for (int mode = 0; mode < 2; ++mode) {
      CK(hipEventRecord(e0, nullptr));
      for (int it = 0; it < 48; ++it) {
        const float* src = mode == 0 ? buf[it % R] : buf[it % (R + 1)];
        float* dst = mode == 0 ? buf[(it + R / 2) % R] : buf[(it + 1) % (R + 1)];
        if (pair256<EPI_MULD, false>(src, S1, S2, dst, DM, nullptr, nullptr, n_img, C, nullptr)) { std::printf("launch failed\n"); return 1; }
      }
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("48 launches rotating over %d buffers%s: %.1f us each\n", mode == 0 ? R : R + 1,
                  mode == 0 ? "" : ", each reading the previous one's output", ms * 1000.f / 48);
    }
  }
  std::vector<long long> st((size_t)n_wg * 8);
  CK(hipMemcpy(st.data(), ST, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
  // start times on the constant-rate wall clock (100 MHz, chip-wide); phase lengths on the shader clock of the workgroup's own CU
  long long w_min = st[7], w_max = 0;
  for (int w = 0; w < n_wg; ++w) { w_min = std::min(w_min, st[(size_t)w * 8 + 7]); w_max = std::max(w_max, st[(size_t)w * 8 + 7]); }
  const double total = (double)(w_max - w_min) + 1.0;
  std::printf("k_pair256<EPI_MULD> %d chains after %d warm-up launches: %d workgroups, first start -> last start %.1f us\n", chains, warm, n_wg, total / 100.0);
  const char* names[6] = {"S1 slab staging + barrier", "phase 1 MFMA loop", "T1 slab -> LDS + 2 barriers", "phase 2 MFMA loop",
                          "epilogue staging (acc -> LDS)", "epilogue (aux reads, stores)"};
  // by start time: round 0 = started with the launch (2 per CU), round 2 = the last starters (the half-occupied round)
  for (int round = -1; round < 3; ++round) {
    std::vector<std::vector<double>> d(7);
    int count = 0;
    for (int w = 0; w < n_wg; ++w) {
      const double rel = (double)(st[(size_t)w * 8 + 7] - w_min) / total;
      const int r = rel < 0.05 ? 0 : (rel < 0.75 ? 1 : 2);
      if (round >= 0 && r != round) continue;
      ++count;
      for (int p = 0; p < 6; ++p) d[p].push_back((double)(st[(size_t)w * 8 + p + 1] - st[(size_t)w * 8 + p]));
      d[6].push_back((double)(st[(size_t)w * 8 + 6] - st[(size_t)w * 8]));
    }
    if (!count) continue;
    if (round < 0) std::printf("all %d workgroups:\n", count); else std::printf("round %d (%d workgroups):\n", round, count);
    for (int p = 0; p < 7; ++p) {
      std::sort(d[p].begin(), d[p].end());
      double mean = 0; for (double v : d[p]) mean += v; mean /= d[p].size();
      std::printf("   %-34s mean %8.0f  median %8.0f  p90 %8.0f cycles%s\n", p < 6 ? names[p] : "whole workgroup", mean, d[p][d[p].size() / 2],
                  d[p][(size_t)(d[p].size() * 0.9)], p == 1 || p == 3 ? "   (256 MFMAs per wave: 16 384 cycles of one SIMD's pipe per wave, 2 waves per SIMD)" : "");
    }
  }
  return 0;
}
