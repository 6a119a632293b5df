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
                                                                float* __restrict__ OUT, int d, int pmode = 0, int krep = 1, int nostore = 0) {
  {
    const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    int pr = 0;
    if (pmode == 1) pr = (L >> 3) % 3;
    else if (pmode == 2) pr = (L >> 8) % 3;
    else if (pmode == 3) pr = (L >> 5) % 3;
    else if (pmode == 4) pr = (int)((L * 2654435761u) >> 30);
    else if (pmode == 5) pr = L % 3;
    else if (pmode == 6) pr = (L >> 2) % 3;
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr == 3) __builtin_amdgcn_s_setprio(3);
  }
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
  for (int it = 0; it < nk * krep; ++it) {
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
        if (!nostore || acc[fa][fb][r] == 123.456f) OUT[(long)img * d * d + (long)row * d + col] = acc[fa][fb][r];
      }
}

// Variant 3: all-glds staging (global_load_lds_dwordx4 straight into a lane-linear LDS image), NB LDS buffers in ONE
// __shared__ array, one barrier per K-step: tile t+1 is in flight while tile t is multiplied.
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int T, int NW, int BK, int MINW>
__global__ __launch_bounds__(64 * NW * NW, MINW) void k_gemm3(const float* __restrict__ IN, const float* __restrict__ S,
                                                               float* __restrict__ OUT, int d, int krep = 1, long long* stamps = nullptr) {
  constexpr int NT = 64 * NW * NW, FR = T / (32 * NW), NV = (BK * T / 4) / NT;
  __shared__ float L[2][2][BK * T];                     // [buffer][operand][k][T]
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
#define GLDS(K0, BUF) _Pragma("unroll") for (int v = 0; v < NV; ++v) { const int idx = tid + v * NT, kk = idx / (T / 4), c4 = idx % (T / 4); \
    __builtin_amdgcn_global_load_lds((gptr_t)&X[(long)((K0) + kk) * d + tr + c4 * 4], (lptr_t)&L[BUF][0][idx * 4], 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gptr_t)&S[(long)((K0) + kk) * d + tc + c4 * 4], (lptr_t)&L[BUF][1][idx * 4], 16, 0, 0); }
  GLDS(0, 0)
  const int nk = d / BK * krep;
  for (int it = 0; it < nk; ++it) {
    const int buf = it & 1;
    const bool st = stamps && (threadIdx.x & 63) == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z % 24 == 0 && it < 48;
    long long* sp = stamps + ((blockIdx.z / 24) * 4 + (threadIdx.x >> 6)) * 48 * 4 + it * 4;
    if (st) sp[0] = clock64();
    __syncthreads();                                     // vmcnt(0) + barrier: tile `it` landed, tile it-1 fully read
    if (st) sp[1] = clock64();
    if (it + 1 < nk) { GLDS(((it + 1) * BK) % d, buf ^ 1) }
    if (st) sp[2] = clock64();
    const float* As = L[buf][0];
    const float* Bs = L[buf][1];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[FR], b[FR];
#pragma unroll
      for (int f = 0; f < FR; ++f) { a[f] = As[(kk + lh) * T + (wi * FR + f) * 32 + lr]; b[f] = Bs[(kk + lh) * T + (wj * FR + f) * 32 + lr]; }
#pragma unroll
      for (int fa = 0; fa < FR; ++fa)
#pragma unroll
        for (int fb = 0; fb < FR; ++fb) acc[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fa], b[fb], acc[fa][fb], 0, 0, 0);
    }
    if (st) sp[3] = clock64();
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

// Sustained fp32-MFMA ceiling: registers only, 4 independent accumulators per wave, the product kernel's occupancy.
__global__ __launch_bounds__(256) void k_mfma_peak(float* out, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = (float)threadIdx.x * 1e-3f, y = (float)blockIdx.x * 1e-4f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 123.456f) out[0] = s;
}

// Loop dissection: MODE 0 = LDS fragment reads + MFMA only; 1 = + the two barriers per K-step; 2 = + global->LDS staging
template <int MODE, int UNR>
__global__ __launch_bounds__(256) void k_loop(const float* __restrict__ IN, const float* __restrict__ S,
                                              float* __restrict__ OUT, int d, int krep, long long* clk = nullptr) {
  const long long c0 = clock64(), w0 = wall_clock64();
  constexpr int T = 128, BK = 32, NT = 256, FR = 2, NV = (BK * T / 4) / NT;
  __shared__ float As[BK * T];
  __shared__ float Bs[BK * T];
  const int img = blockIdx.z, tr = blockIdx.y * T, tc = blockIdx.x * T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wi = wave / 2, wj = wave % 2, lr = lane & 31, lh = lane >> 5;
  const float* __restrict__ X = IN + (long)img * d * d;
  f32x16 acc[FR][FR];
#pragma unroll
  for (int a = 0; a < FR; ++a)
#pragma unroll
    for (int b = 0; b < FR; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  if (MODE == 5) { for (int i = tid; i < BK * T; i += NT) { As[i] = X[i]; Bs[i] = S[i + 4096]; } __syncthreads(); }
  else if (MODE != 2) { for (int i = tid; i < BK * T; i += NT) { As[i] = 0.001f * i; Bs[i] = 0.002f * i; } __syncthreads(); }
  const int nk = d / BK;
  for (int it = 0; it < nk * krep; ++it) {
    const int k0 = (it % nk) * BK;
    if (MODE == 2) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int idx = tid + v * NT, kk = idx / (T / 4), c4 = idx % (T / 4);
        *reinterpret_cast<v4f*>(&As[kk * T + c4 * 4]) = *reinterpret_cast<const v4f*>(&X[(long)(k0 + kk) * d + tr + c4 * 4]);
        *reinterpret_cast<v4f*>(&Bs[kk * T + c4 * 4]) = *reinterpret_cast<const v4f*>(&S[(long)(k0 + kk) * d + tc + c4 * 4]);
      }
    }
    if (MODE == 3) {                                   // global loads only; result folded into one LDS word nobody reads
      v4f t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int idx = tid + v * NT, kk = idx / (T / 4), c4 = idx % (T / 4);
        t += *reinterpret_cast<const v4f*>(&X[(long)(k0 + kk) * d + tr + c4 * 4]);
        t += *reinterpret_cast<const v4f*>(&S[(long)(k0 + kk) * d + tc + c4 * 4]);
      }
      if (t.x == 123.456f) As[tid] = t.y;
    }
    if (MODE == 4) {                                   // LDS writes only
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int idx = tid + v * NT, kk = idx / (T / 4), c4 = idx % (T / 4);
        const v4f t = {0.001f * it, 0.002f * v, 0.5f, 0.25f};
        *reinterpret_cast<v4f*>(&As[kk * T + c4 * 4]) = t;
        *reinterpret_cast<v4f*>(&Bs[kk * T + c4 * 4]) = t;
      }
    }
    if (MODE >= 1) __syncthreads();
#pragma unroll UNR
    for (int kk = 0; kk < BK; kk += 2) {
      float a[FR], b[FR];
#pragma unroll
      for (int f = 0; f < FR; ++f) { a[f] = As[(kk + lh) * T + (wi * FR + f) * 32 + lr]; b[f] = Bs[(kk + lh) * T + (wj * FR + f) * 32 + lr]; }
#pragma unroll
      for (int fa = 0; fa < FR; ++fa)
#pragma unroll
        for (int fb = 0; fb < FR; ++fb) acc[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fa], b[fb], acc[fa][fb], 0, 0, 0);
    }
    if (MODE >= 1) __syncthreads();
  }
  if (clk && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && (blockIdx.z % 37) == 0) { clk[2 * (blockIdx.z / 37)] = clock64() - c0; clk[2 * (blockIdx.z / 37) + 1] = wall_clock64() - w0; }
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

// Variant 4: loader-wave specialisation.  Waves 0..3 only read fragments and issue MFMAs; wave 4 streams both operand
// tiles with glds into the other LDS buffer.  One barrier per K-step, executed by all five waves.
template <int BK, int NBUF>
__global__ __launch_bounds__(320) void k_gemm4(const float* __restrict__ IN, const float* __restrict__ S,
                                               float* __restrict__ OUT, int d, int krep, long long* unused) {
  constexpr int T = 128, FR = 2;
  __shared__ float L[NBUF][2][BK * T];
  const int img = blockIdx.z, tr = blockIdx.y * T, tc = blockIdx.x * T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wi = (wave >> 1) & 1, wj = wave & 1, lr = lane & 31, lh = lane >> 5;
  const float* __restrict__ X = IN + (long)img * d * d;
  const int nk = d / BK * krep;
  // loader: one glds moves 64 lanes x 16 B = 2 rows of 128 floats; BK/2 per operand per K-step
#define LOAD_TILE(K0, BUF) _Pragma("unroll") for (int v = 0; v < BK / 2; ++v) { const int kk = v * 2 + (lane >> 5), c4 = lane & 31; \
    __builtin_amdgcn_global_load_lds((gptr_t)&X[(long)((K0) + kk) * d + tr + c4 * 4], (lptr_t)&L[BUF][0][(v * 64 + lane) * 4], 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gptr_t)&S[(long)((K0) + kk) * d + tc + c4 * 4], (lptr_t)&L[BUF][1][(v * 64 + lane) * 4], 16, 0, 0); }
  if (wave == 4) {
    for (int p = 0; p < NBUF - 1; ++p) { LOAD_TILE((p * BK) % d, p) }
    for (int it = 0; it < nk; ++it) {
      // tile `it` must have landed before the barrier; tiles it+1 .. it+NBUF-2 may stay in flight
      if (NBUF == 2) __builtin_amdgcn_s_waitcnt(0x0f70 | 0);            // vmcnt(0) (gfx9 encoding: lgkm/exp untouched)
      else __builtin_amdgcn_s_waitcnt(0x0f70 | (BK * (NBUF - 2) > 15 ? 15 : BK * (NBUF - 2)));
      __builtin_amdgcn_s_barrier();
      if (it + NBUF - 1 < nk) { LOAD_TILE(((it + NBUF - 1) * BK) % d, (it + NBUF - 1) % NBUF) }
    }
    return;
  }
  f32x16 acc[FR][FR];
#pragma unroll
  for (int a = 0; a < FR; ++a)
#pragma unroll
    for (int b = 0; b < FR; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  for (int it = 0; it < nk; ++it) {
    __builtin_amdgcn_s_barrier();
    const float* As = L[it % NBUF][0];
    const float* Bs = L[it % NBUF][1];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[FR], b[FR];
#pragma unroll
      for (int f = 0; f < FR; ++f) { a[f] = As[(kk + lh) * T + (wi * FR + f) * 32 + lr]; b[f] = Bs[(kk + lh) * T + (wj * FR + f) * 32 + lr]; }
#pragma unroll
      for (int fa = 0; fa < FR; ++fa)
#pragma unroll
        for (int fb = 0; fb < FR; ++fb) acc[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fa], b[fb], acc[fa][fb], 0, 0, 0);
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
#define HP(PM) bench("v2 T128 2x2 BK32 prio-mode " #PM, [&]() { \
    hipLaunchKernelGGL((k_gemm2<128, 2, 2, 32, false, 1>), dim3(d / 128, d / 128, NIMG), dim3(256), 0, 0, IN, S, OUT, d, PM); })
#define HK(KR, NS) bench("v2 T128 2x2 BK32 krep " #KR " nostore " #NS, [&]() { \
    hipLaunchKernelGGL((k_gemm2<128, 2, 2, 32, false, 1>), dim3(d / 128, d / 128, NIMG), dim3(256), 0, 0, IN, S, OUT, d, 0, KR, NS); })
#define GD(T, NW, BK, DYN) bench("sync T" #T " NW" #NW " BK" #BK " +dynLDS " #DYN, [&]() { \
    hipLaunchKernelGGL((k_gemm<T, NW, BK, 0, false>), dim3(d / T, d / T, NIMG), dim3(64 * NW * NW), DYN, 0, IN, S, OUT, d); })
#define G3(T, NW, BK, MINW) bench("glds T" #T " NW" #NW " BK" #BK " MINW" #MINW, [&]() { \
    hipLaunchKernelGGL((k_gemm3<T, NW, BK, MINW>), dim3(d / T, d / T, NIMG), dim3(64 * NW * NW), 0, 0, IN, S, OUT, d); })
  for (int wg : {256, 512, 768, 1024}) {
    const int iters = 4096;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_mfma_peak, dim3(wg), dim3(256), 0, 0, OUT, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_mfma_peak, dim3(wg), dim3(256), 0, 0, OUT, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mfma peak, %d blocks x 4 waves: %.1f us/launch -> %.1f TFLOP/s\n", wg, ms / 10 * 1e3,
           (double)wg * 4 * iters * 4 * 4096.0 / (ms / 10 * 1e-3) / 1e12); fflush(stdout);
  }
  auto loop_time = [&](auto kern, int krep) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(2, 2, NIMG), dim3(256), 0, 0, IN, S, OUT, d, krep);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(2, 2, NIMG), dim3(256), 0, 0, IN, S, OUT, d, krep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 20 * 1e3;
  };
  auto loop_time2 = [&](auto kern, int krep) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(2, 2, NIMG), dim3(256), 0, 0, IN, S, OUT, d, krep, (long long*)nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(2, 2, NIMG), dim3(256), 0, 0, IN, S, OUT, d, krep, (long long*)nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 20 * 1e3;
  };
#define LP(MODE, UNR) { auto kern = [&](int kr) { return loop_time2(k_loop<MODE, UNR>, kr); }; double t2 = kern(2), t6 = kern(6); \
    printf("loop MODE %d UNR %d: krep2 %.1f us, krep6 %.1f us -> %.1f us per product-loop = %.1f TFLOP/s, fixed %.1f us\n", MODE, UNR, t2, t6, (t6 - t2) / 4, \
           flop / ((t6 - t2) / 4 * 1e-6) / 1e12, t2 - 2 * (t6 - t2) / 4); fflush(stdout); }
  LP(1, 16) LP(5, 16) LP(2, 16)
  { long long* clk; CK(hipMalloc(&clk, 16 * 8)); long long hc[12];
#define CLK_UNUSED(MODE) { hipLaunchKernelGGL((k_loop<MODE, 16>), dim3(2, 2, NIMG), dim3(256), 0, 0, IN, S, OUT, d, 40, clk); CK(hipDeviceSynchronize()); \
      hipLaunchKernelGGL((k_loop<MODE, 16>), dim3(2, 2, NIMG), dim3(256), 0, 0, IN, S, OUT, d, 40, clk); CK(hipDeviceSynchronize()); \
      CK(hipMemcpy(hc, clk, 12 * 8, hipMemcpyDeviceToHost)); \
      for (int q = 0; q < 5; ++q) printf("MODE %d block %d: %lld shader cycles / %lld ticks(100MHz) -> %.0f MHz\n", MODE, q * 37, hc[2 * q], hc[2 * q + 1], hc[2 * q] / (hc[2 * q + 1] / 100.0)); }
    }
  if (0) { const int NB = 8, NS = NB * 4 * 48 * 4; long long* st; CK(hipMalloc(&st, NS * 8)); CK(hipMemset(st, 0, NS * 8));
    std::vector<long long> hs2(NS);
    for (int i = 0; i < 2; ++i) { hipLaunchKernelGGL((k_gemm3<128, 2, 16, 1>), dim3(2, 2, NIMG), dim3(256), 0, 0, IN, S, OUT, d, 3, st); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(hs2.data(), st, NS * 8, hipMemcpyDeviceToHost));
    for (int b = 0; b < NB; ++b) for (int w = 0; w < 4; w += 3) {
      const long long* q = &hs2[(b * 4 + w) * 48 * 4];
      double bar = 0, iss = 0, cmp = 0; for (int it = 8; it < 40; ++it) { bar += q[it * 4 + 1] - q[it * 4]; iss += q[it * 4 + 2] - q[it * 4 + 1]; cmp += q[it * 4 + 3] - q[it * 4 + 2]; }
      printf("stamps glds BK16 block %3d wave %d: start %lld  per K-step(16): barrier %.0f  glds-issue %.0f  compute %.0f  cycles; first barrier %lld, total(48 steps) %lld\n", b * 24, w, q[0] - hs2[0], bar / 32, iss / 32, cmp / 32, q[1] - q[0], q[47 * 4 + 3] - q[0]);
    } }

#define LP3(BK, MINW) { auto kern = k_gemm3<128, 2, BK, MINW>; double t2 = loop_time2(kern, 2), t6 = loop_time2(kern, 6); \
    printf("glds BK %d MINW %d: krep2 %.1f us, krep6 %.1f us -> %.1f us per product-loop = %.1f TFLOP/s, fixed %.1f us\n", BK, MINW, t2, t6, (t6 - t2) / 4, \
           flop / ((t6 - t2) / 4 * 1e-6) / 1e12, t2 - 2 * (t6 - t2) / 4); fflush(stdout); }
  LP3(16, 1)
  auto loop_time4 = [&](auto kern, int krep) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(2, 2, NIMG), dim3(320), 0, 0, IN, S, OUT, d, krep, (long long*)nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(2, 2, NIMG), dim3(320), 0, 0, IN, S, OUT, d, krep, (long long*)nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 20 * 1e3;
  };
#define LP4(BK, NBUF) { auto kern = k_gemm4<BK, NBUF>; double t1 = loop_time4(kern, 1), t2 = loop_time4(kern, 2), t6 = loop_time4(kern, 6); \
    CK(hipMemset(OUT, 0, n * 4)); hipLaunchKernelGGL(kern, dim3(2, 2, NIMG), dim3(320), 0, 0, IN, S, OUT, d, 1, (long long*)nullptr); CK(hipMemcpy(out.data(), OUT, n * 4, hipMemcpyDeviceToHost)); \
    double cs = 0; for (size_t i = 0; i < n; i += 997) cs += out[i]; \
    printf("loader-wave BK %d NBUF %d: krep1 %.1f us (%.1f TFLOP/s), krep2 %.1f, krep6 %.1f -> %.1f us per product-loop = %.1f TFLOP/s, fixed %.1f us, checksum %.6e\n", BK, NBUF, t1, flop / (t1 * 1e-6) / 1e12, t2, t6, (t6 - t2) / 4, \
           flop / ((t6 - t2) / 4 * 1e-6) / 1e12, t2 - 2 * (t6 - t2) / 4, cs); fflush(stdout); }
  LP4(16, 2)
  { CK(hipMemset(OUT, 0, n * 4)); hipLaunchKernelGGL((k_gemm<128, 2, 32, 0, false>), dim3(2, 2, NIMG), dim3(256), 0, 0, IN, S, OUT, d); CK(hipMemcpy(out.data(), OUT, n * 4, hipMemcpyDeviceToHost));
    double cs = 0; for (size_t i = 0; i < n; i += 997) cs += out[i]; printf("reference checksum %.6e\n", cs); }
  for (int rep = 0; rep < 1; ++rep) {
    G(128, 2, 32, 0, false);
    HK(1, 0); HK(2, 0); HK(3, 0); HK(1, 1); HK(2, 1);
    H(128, 2, 2, 32, false, 4);
  }
  return 0;
}
