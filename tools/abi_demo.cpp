// Plain C++ consumer of the C ABI (no Python, no torch): proves libnhmc.so stands alone behind include/nhmc.h.
//   hipcc -O2 -ffp-contract=off tools/abi_demo.cpp -Iinclude -Lnoise-space-hmc_amd -lnhmc \
//         -Wl,-rpath,$PWD/noise-space-hmc_amd -o /tmp/abi_demo && /tmp/abi_demo
// Runs the fused leapfrog update (MID) on 8 chains of 3x64x64, checks it bit for bit against the scalar loop the
// reference's ops amount to (main_sampling.py:713 then :706), then exercises an error path.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "nhmc.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s\n", hipGetErrorString(e_)); return 2; } } while (0)

int main() {
  const int B = 8;
  const int64_t N = 3 * 64 * 64;
  std::vector<float> x(B * N), p(B * N), g(B * N);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f; };
  for (auto& v : x) v = rnd();
  for (auto& v : p) v = rnd();
  for (auto& v : g) v = 3.0f * rnd();
  std::vector<double> eps(B), sig(B);
  for (int b = 0; b < B; ++b) { eps[b] = 0.05 * std::pow(0.95, b); sig[b] = 0.1 + 0.2 * b; }
  const double m = 1.3, m_inv = 1.0 / m;

  float *dx, *dp, *dg; double *de, *ds;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dp, x.size() * 4)); CK(hipMalloc(&dg, x.size() * 4));
  CK(hipMalloc(&de, B * 8)); CK(hipMalloc(&ds, B * 8));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dp, p.data(), x.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, g.data(), x.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(de, eps.data(), B * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, sig.data(), B * 8, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));

  if (nhmc_abi_version() != NHMC_ABI_VERSION) { std::printf("ABI version mismatch\n"); return 1; }
  int rc = nhmc_leapfrog_fused(NHMC_LF_MID, dx, dp, dg, nullptr, de, ds, m_inv, B, N, nullptr, st);
  if (rc != NHMC_OK) { std::printf("launch failed: %s (%s)\n", nhmc_status_string(rc), nhmc_last_launch_error()); return 1; }
  CK(hipStreamSynchronize(st));
  std::vector<float> hx(x.size()), hp(x.size());
  CK(hipMemcpy(hx.data(), dx, x.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hp.data(), dp, x.size() * 4, hipMemcpyDeviceToHost));

  size_t bad = 0;
  for (int b = 0; b < B; ++b) {
    const float kf = (float)(1.0 / (2.0 * (sig[b] * sig[b]))), ef = (float)eps[b], ex = (float)(eps[b] * m_inv);
    for (int64_t i = 0; i < N; ++i) {
      const size_t k = (size_t)b * N + i;
      const float t1 = kf * g[k];          // separate roundings, as the reference's ATen ops
      const float G = x[k] + t1;
      const float t2 = ef * G;
      const float pn = p[k] - t2;
      const float t3 = ex * pn;
      const float xn = x[k] + t3;
      if (std::memcmp(&pn, &hp[k], 4) || std::memcmp(&xn, &hx[k], 4)) ++bad;
    }
  }
  std::printf("leapfrog MID: %zu of %zu elements differ from the scalar reference loop\n", bad, x.size());

  // error paths: misaligned pointer, bad element count -- refused before any launch
  const int e1 = nhmc_leapfrog_fused(NHMC_LF_MID, dx + 1, dp, dg, nullptr, de, ds, m_inv, B, N, nullptr, st);
  const int e2 = nhmc_leapfrog_fused(NHMC_LF_MID, dx, dp, dg, nullptr, de, ds, m_inv, B, N - 1, nullptr, st);
  const int e3 = nhmc_leapfrog_fused(NHMC_LF_FIRST, dx, dp, dg, nullptr, de, ds, m_inv, B, N, nullptr, st);
  std::printf("error paths: %d %d %d (%s)\n", e1, e2, e3, nhmc_status_string(e1));
  const bool ok = bad == 0 && e1 == NHMC_ERR_ALIGN && e2 == NHMC_ERR_ALIGN && e3 == NHMC_ERR_ARG;
  std::printf(ok ? "ABI demo OK\n" : "ABI demo FAILED\n");
  return ok ? 0 : 1;
}
