// Standalone A/B harness for the spectral GEMM  OUT = IN^T * S  (fp32 MFMA), not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// PIPE: 0 = synchronous staging, 1 = register prefetch (one LDS buffer), 2 = LDS double buffer + register prefetch
// LPF : software-pipeline the LDS fragment reads one k-step ahead
template <int T, int NW, int BK, int PIPE, bool LPF>
__global__ __launch_bounds__(64 * NW * NW) void k_gemm(const float* __restrict__ IN, const float* __restrict__ S,
                                                       float* __restrict__ OUT, int d) {
  constexpr int NT = 64 * NW * NW, FR = T / (32 * NW), NV = (BK * T / 4) / NT, NB = PIPE == 2 ? 2 : 1;
  __shared__ float As[NB][BK * T];
  __shared__ float Bs[NB][BK * T];
  const int img = blockIdx.z, tr = blockIdx.y * T, tc = blockIdx.x * T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wi = wave / NW, wj = wave % NW, lr = lane & 31, lh = lane >> 5;
  const float* __restrict__ X = IN + (long)img * d * d;
  f32x16 acc[FR][FR];
#pragma unroll
  for (int a = 0; a < FR; ++a)
#pragma unroll
    for (int b = 0; b < FR; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  v4f pa[NV], pb[NV];
#define FETCH(K0) _Pragma("unroll") for (int v = 0; v < NV; ++v) { const int idx = tid + v * NT, kk = idx / (T / 4), c4 = idx % (T / 4); \
    pa[v] = *reinterpret_cast<const v4f*>(&X[(long)((K0) + kk) * d + tr + c4 * 4]); pb[v] = *reinterpret_cast<const v4f*>(&S[(long)((K0) + kk) * d + tc + c4 * 4]); }
#define STAGE(BUF) _Pragma("unroll") for (int v = 0; v < NV; ++v) { const int idx = tid + v * NT, kk = idx / (T / 4), c4 = idx % (T / 4); \
    *reinterpret_cast<v4f*>(&As[BUF][kk * T + c4 * 4]) = pa[v]; *reinterpret_cast<v4f*>(&Bs[BUF][kk * T + c4 * 4]) = pb[v]; }
#define COMPUTE(BUF) { \
    float a[FR], b[FR], an[FR], bn[FR]; \
    if (LPF) { _Pragma("unroll") for (int f = 0; f < FR; ++f) { a[f] = As[BUF][lh * T + (wi * FR + f) * 32 + lr]; b[f] = Bs[BUF][lh * T + (wj * FR + f) * 32 + lr]; } } \
    _Pragma("unroll") for (int kk = 0; kk < BK; kk += 2) { \
      if (LPF) { if (kk + 2 < BK) { _Pragma("unroll") for (int f = 0; f < FR; ++f) { an[f] = As[BUF][(kk + 2 + lh) * T + (wi * FR + f) * 32 + lr]; bn[f] = Bs[BUF][(kk + 2 + lh) * T + (wj * FR + f) * 32 + lr]; } } } \
      else { _Pragma("unroll") for (int f = 0; f < FR; ++f) { a[f] = As[BUF][(kk + lh) * T + (wi * FR + f) * 32 + lr]; b[f] = Bs[BUF][(kk + lh) * T + (wj * FR + f) * 32 + lr]; } } \
      _Pragma("unroll") for (int fa = 0; fa < FR; ++fa) _Pragma("unroll") for (int fb = 0; fb < FR; ++fb) \
        acc[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fa], b[fb], acc[fa][fb], 0, 0, 0); \
      if (LPF) { _Pragma("unroll") for (int f = 0; f < FR; ++f) { a[f] = an[f]; b[f] = bn[f]; } } \
    } }
  if (PIPE == 0) {
    for (int k0 = 0; k0 < d; k0 += BK) { FETCH(k0) STAGE(0) __syncthreads(); COMPUTE(0) __syncthreads(); }
  } else if (PIPE == 1) {
    FETCH(0) STAGE(0) __syncthreads();
    for (int k0 = 0; k0 < d; k0 += BK) {
      const bool more = k0 + BK < d;
      if (more) { FETCH(k0 + BK) }
      COMPUTE(0)
      __syncthreads();
      if (more) { STAGE(0) __syncthreads(); }
    }
  } else {
    FETCH(0) STAGE(0) __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < d; k0 += BK) {
      const bool more = k0 + BK < d;
      if (more) { FETCH(k0 + BK) }
      if (buf == 0) { COMPUTE(0) if (more) { STAGE(1) } } else { COMPUTE(1) if (more) { STAGE(0) } }
      __syncthreads();
      buf ^= 1;
    }
  }
#pragma unroll
  for (int fa = 0; fa < FR; ++fa)
#pragma unroll
    for (int fb = 0; fb < FR; ++fb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tr + (wi * FR + fa) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = tc + (wj * FR + fb) * 32 + lr;
        OUT[(long)img * d * d + (long)row * d + col] = acc[fa][fb][r];
      }
}

// Variant 2: WM x WN waves (rectangular wave grid), optional rotated k order per block (stagger), optional bound
template <int T, int WM, int WN, int BK, bool ROT, int MINW>
__global__ __launch_bounds__(64 * WM * WN, MINW) void k_gemm2(const float* __restrict__ IN, const float* __restrict__ S,
                                                                float* __restrict__ OUT, int d) {
  constexpr int NT = 64 * WM * WN, FRI = T / (32 * WM), FRJ = T / (32 * WN), NV = (BK * T / 4) / NT;
  __shared__ float As[BK * T];
  __shared__ float Bs[BK * T];
  const int img = blockIdx.z, tr = blockIdx.y * T, tc = blockIdx.x * T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wi = wave / WN, wj = wave % WN, lr = lane & 31, lh = lane >> 5;
  const float* __restrict__ X = IN + (long)img * d * d;
  f32x16 acc[FRI][FRJ];
#pragma unroll
  for (int a = 0; a < FRI; ++a)
#pragma unroll
    for (int b = 0; b < FRJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int nk = d / BK;
  const int rot = ROT ? (int)((blockIdx.x + blockIdx.y * 3 + blockIdx.z * 5) % nk) : 0;
  for (int it = 0; it < nk; ++it) {
    const int k0 = ((it + rot) % nk) * BK;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = tid + v * NT, kk = idx / (T / 4), c4 = idx % (T / 4);
      *reinterpret_cast<v4f*>(&As[kk * T + c4 * 4]) = *reinterpret_cast<const v4f*>(&X[(long)(k0 + kk) * d + tr + c4 * 4]);
      *reinterpret_cast<v4f*>(&Bs[kk * T + c4 * 4]) = *reinterpret_cast<const v4f*>(&S[(long)(k0 + kk) * d + tc + c4 * 4]);
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < BK; kk += 2) {
      float a[FRI], b[FRJ];
#pragma unroll
      for (int f = 0; f < FRI; ++f) a[f] = As[(kk + lh) * T + (wi * FRI + f) * 32 + lr];
#pragma unroll
      for (int f = 0; f < FRJ; ++f) b[f] = Bs[(kk + lh) * T + (wj * FRJ + f) * 32 + lr];
#pragma unroll
      for (int fa = 0; fa < FRI; ++fa)
#pragma unroll
        for (int fb = 0; fb < FRJ; ++fb) acc[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fa], b[fb], acc[fa][fb], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int fa = 0; fa < FRI; ++fa)
#pragma unroll
    for (int fb = 0; fb < FRJ; ++fb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tr + (wi * FRI + fa) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = tc + (wj * FRJ + fb) * 32 + lr;
        OUT[(long)img * d * d + (long)row * d + col] = acc[fa][fb][r];
      }
}

int main() {
  const int d = 256, NIMG = 192;
  const size_t n = (size_t)NIMG * d * d;
  std::vector<float> h(n), hs(d * d);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  for (int i = 0; i < d * d; ++i) hs[i] = (float)((i * 40503u) % 1999) / 1000.f - 1.f;
  float *IN, *S, *OUT, *REF; CK(hipMalloc(&IN, n * 4)); CK(hipMalloc(&OUT, n * 4)); CK(hipMalloc(&REF, n * 4)); CK(hipMalloc(&S, d * d * 4));
  CK(hipMemcpy(IN, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(S, hs.data(), d * d * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = 2.0 * NIMG * d * d * (double)d;
  std::vector<float> ref(n), out(n);
  bool have_ref = false;
  auto bench = [&](const char* name, auto launch) {
    CK(hipMemset(OUT, 0, n * 4));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 20; ++i) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / 20 * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    CK(hipMemcpy(out.data(), OUT, n * 4, hipMemcpyDeviceToHost));
    double err = 0;
    if (!have_ref) { ref = out; have_ref = true; } else for (size_t i = 0; i < n; i += 997) err = std::max(err, (double)std::fabs(out[i] - ref[i]));
    printf("%-40s med %.1f us  -> %.1f TFLOP/s   maxdiff %.2e\n", name, ts[2], flop / (ts[2] * 1e-6) / 1e12, err); fflush(stdout);
  };
#define G(T, NW, BK, PIPE, LPF) bench("T" #T " NW" #NW " BK" #BK " PIPE" #PIPE " LPF" #LPF, [&]() { \
    hipLaunchKernelGGL((k_gemm<T, NW, BK, PIPE, LPF>), dim3(d / T, d / T, NIMG), dim3(64 * NW * NW), 0, 0, IN, S, OUT, d); })
#define H(T, WM, WN, BK, ROT, MINW) bench("v2 T" #T " W" #WM "x" #WN " BK" #BK " ROT" #ROT " MINW" #MINW, [&]() { \
    hipLaunchKernelGGL((k_gemm2<T, WM, WN, BK, ROT, MINW>), dim3(d / T, d / T, NIMG), dim3(64 * WM * WN), 0, 0, IN, S, OUT, d); })
  for (int rep = 0; rep < 2; ++rep) {
    G(128, 2, 32, 0, false);
    H(128, 2, 2, 32, false, 1); H(128, 2, 2, 32, true, 1); H(128, 2, 2, 32, false, 4); H(128, 2, 2, 32, true, 4);
    H(128, 4, 2, 32, false, 1); H(128, 4, 2, 32, true, 1); H(128, 2, 4, 32, true, 1); H(128, 4, 4, 32, true, 1);
    H(128, 2, 2, 16, true, 1); H(128, 4, 2, 16, true, 1); H(128, 2, 2, 64, true, 1);
  }
  return 0;
}
