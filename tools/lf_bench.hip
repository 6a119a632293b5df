// Standalone A/B harness for the fused leapfrog update (not part of the product).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/lf_bench.hip -o /tmp/lf_bench && /tmp/lf_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <int VPT, int BLK, bool NT, bool NTS = NT>
__global__ __launch_bounds__(BLK) void k_lf(v4f* __restrict__ x, v4f* __restrict__ p, const v4f* __restrict__ g,
                                            const double* __restrict__ eps, const double* __restrict__ sig, double m_inv,
                                            long n4) {
  const int chain = blockIdx.y;
  const double e = eps[chain], s = sig[chain];
  const float kf = (float)(1.0 / (2.0 * (s * s))), ef = (float)e, ex = (float)(e * m_inv);
  const long base = (long)chain * n4;
  const long t0 = (long)blockIdx.x * (BLK * VPT) + threadIdx.x;
  v4f xv[VPT], pv[VPT], gv[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const long q = t0 + (long)i * BLK;
    if (q < n4) {
      if (NT) {
        xv[i] = __builtin_nontemporal_load(&x[base + q]);
        pv[i] = __builtin_nontemporal_load(&p[base + q]);
        gv[i] = __builtin_nontemporal_load(&g[base + q]);
      } else { xv[i] = x[base + q]; pv[i] = p[base + q]; gv[i] = g[base + q]; }
    }
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const long q = t0 + (long)i * BLK;
    if (q >= n4) continue;
    v4f G = xv[i] + kf * gv[i];
    pv[i] = pv[i] - ef * G;
    xv[i] = xv[i] + ex * pv[i];
    if (NTS) { __builtin_nontemporal_store(pv[i], &p[base + q]); __builtin_nontemporal_store(xv[i], &x[base + q]); }
    else { p[base + q] = pv[i]; x[base + q] = xv[i]; }
  }
}

// grid-stride persistent form: gridDim.x blocks per chain-agnostic flat range
template <int BLK, bool NT>
__global__ __launch_bounds__(BLK) void k_lf_gs(v4f* __restrict__ x, v4f* __restrict__ p, const v4f* __restrict__ g,
                                               const double* __restrict__ eps, const double* __restrict__ sig, double m_inv,
                                               long n4, long total4) {
  for (long q0 = (long)blockIdx.x * BLK * 4; q0 < total4; q0 += (long)gridDim.x * BLK * 4) {
    v4f xv[4], pv[4], gv[4];
    float kf[4], ef[4], ex[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long q = q0 + threadIdx.x + (long)i * BLK;
      if (q < total4) {
        const int chain = (int)(q / n4);
        const double e = eps[chain], s = sig[chain];
        kf[i] = (float)(1.0 / (2.0 * (s * s))); ef[i] = (float)e; ex[i] = (float)(e * m_inv);
        if (NT) { xv[i] = __builtin_nontemporal_load(&x[q]); pv[i] = __builtin_nontemporal_load(&p[q]); gv[i] = __builtin_nontemporal_load(&g[q]); }
        else { xv[i] = x[q]; pv[i] = p[q]; gv[i] = g[q]; }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long q = q0 + threadIdx.x + (long)i * BLK;
      if (q >= total4) continue;
      v4f G = xv[i] + kf[i] * gv[i];
      pv[i] = pv[i] - ef[i] * G;
      xv[i] = xv[i] + ex[i] * pv[i];
      if (NT) { __builtin_nontemporal_store(pv[i], &p[q]); __builtin_nontemporal_store(xv[i], &x[q]); }
      else { p[q] = pv[i]; x[q] = xv[i]; }
    }
  }
}

// traffic twin with no arithmetic: 3 reads, 2 writes (ceiling for this access shape)
template <int VPT, int BLK>
__global__ __launch_bounds__(BLK) void k_copy32(v4f* __restrict__ x, v4f* __restrict__ p, const v4f* __restrict__ g, long n4) {
  const long base = (long)blockIdx.y * n4;
  const long t0 = (long)blockIdx.x * (BLK * VPT) + threadIdx.x;
  v4f a[VPT], b[VPT], c[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) { const long q = t0 + (long)i * BLK; if (q < n4) { a[i] = x[base + q]; b[i] = p[base + q]; c[i] = g[base + q]; } }
#pragma unroll
  for (int i = 0; i < VPT; ++i) { const long q = t0 + (long)i * BLK; if (q < n4) { p[base + q] = a[i] + c[i]; x[base + q] = b[i]; } }
}

struct Bufs { v4f *x, *p, *g; };

int main() {
  const int B = 64; const long N = 3L * 256 * 256, n4 = N / 4, total4 = n4 * B;
  const int R = 6;
  std::vector<Bufs> sets(R);
  std::vector<float> h(N * B);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
  for (auto& s : sets) {
    CK(hipMalloc(&s.x, N * B * 4)); CK(hipMalloc(&s.p, N * B * 4)); CK(hipMalloc(&s.g, N * B * 4));
    CK(hipMemcpy(s.x, h.data(), N * B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(s.p, h.data(), N * B * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(s.g, h.data(), N * B * 4, hipMemcpyHostToDevice));
  }
  double *eps, *sig; CK(hipMalloc(&eps, B * 8)); CK(hipMalloc(&sig, B * 8));
  std::vector<double> he(B, 1e-3), hs(B, 1.7);
  CK(hipMemcpy(eps, he.data(), B * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(sig, hs.data(), B * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = 5.0 * N * B * 4;
  const int LAUNCH = 120, ROUNDS = 5;

  auto bench = [&](const char* name, auto launch) {
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
    printf("%-34s min %.2f us  med %.2f us  -> %.0f GB/s (med)\n", name, best[0], best[ROUNDS / 2], bytes / (best[ROUNDS / 2] * 1e-6) / 1e9);
    fflush(stdout);
  };

#define V(VPT, BLK, NT) bench("lf vpt" #VPT " blk" #BLK " nt" #NT, [&](Bufs& s) { \
    dim3 grid((unsigned)((n4 + (BLK) * (VPT) - 1) / ((BLK) * (VPT))), B); \
    hipLaunchKernelGGL((k_lf<VPT, BLK, NT>), grid, dim3(BLK), 0, 0, s.x, s.p, s.g, eps, sig, 1.0, n4); })
#define W(VPT, BLK, NTL, NTS) bench("lf vpt" #VPT " blk" #BLK " ntl" #NTL " nts" #NTS, [&](Bufs& s) { \
    dim3 grid((unsigned)((n4 + (BLK) * (VPT) - 1) / ((BLK) * (VPT))), B); \
    hipLaunchKernelGGL((k_lf<VPT, BLK, NTL, NTS>), grid, dim3(BLK), 0, 0, s.x, s.p, s.g, eps, sig, 1.0, n4); })
  for (int rep = 0; rep < 2; ++rep) {
    W(2, 256, true, true); W(2, 256, true, false); W(2, 256, false, true); W(1, 256, true, true); W(3, 256, true, true);
    W(1, 512, true, true); W(1, 1024, true, true); W(2, 512, true, true); W(3, 512, true, true); W(6, 256, true, true);
    W(2, 128, true, true); W(1, 128, true, true); W(2, 64, true, true);
  }
  return 0;
}
