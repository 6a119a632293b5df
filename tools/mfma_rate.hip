// What does v_mfma_f32_32x32x2_f32 sustain on this chip, by operand source?  (not part of the product)
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o tools/mfma_rate && tools/mfma_rate
// MODE 0: operands constant registers;  1: operands change every MFMA (VALU update, data toggling);
// MODE 2: b operand read from LDS per MFMA (ds_read_b32), a from a register ring filled by global loads 16 ahead
//         -- the operand feed of k_pair256's phase 1.  Each runs NW waves per block, one block per CU slot, ~2 ms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ __launch_bounds__(512) void k_rate(const float* __restrict__ g, float* out, int iters) {
  __shared__ float lds[256 * 64];
  for (int i = threadIdx.x; i < 256 * 64; i += blockDim.x) lds[i] = g[i];
  __syncthreads();
  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5, wave = threadIdx.x >> 6;
  float x = g[threadIdx.x], y = g[threadIdx.x + 512];
  const float* ap = g + wave * 32 + lr + lh * 256;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 2) {
      float f[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) f[j] = ap[(2 * ((i * 16 + j) & 127)) * 256];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int kk = 2 * j + ((i & 7) << 5);
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[j], lds[(kk + lh) * 64 + (a & 1) * 32 + lr], acc[a], 0, 0, 0);
      }
    } else if (MODE == 3) {                      // as 2, LDS operands explicitly fetched one k-pair ahead
      float f[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) f[j] = ap[(2 * ((i * 16 + j) & 127)) * 256];
      const int k0 = (i & 7) << 5;
      float b0 = lds[(k0 + lh) * 64 + lr], b1 = lds[(k0 + lh) * 64 + 32 + lr];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int kn = 2 * ((j + 1) & 15) + k0;
        const float n0 = lds[(kn + lh) * 64 + lr], n1 = lds[(kn + lh) * 64 + 32 + lr];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[j], b0, acc[0], 0, 0, 0);
        acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[j], b1, acc[1 % NACC], 0, 0, 0);
        b0 = n0; b1 = n1;
      }
    } else if (MODE == 4) {                      // 2 x 2 fragments per wave: a[2] from the global ring, b[2] from LDS, 4 MFMAs per k-pair
      float f[16], h[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) { f[j] = ap[(2 * ((i * 16 + j) & 127)) * 256]; h[j] = ap[(2 * ((i * 16 + j) & 127)) * 256 + 32]; }
      const int k0 = (i & 7) << 5;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int kk = 2 * j + k0;
        const float b0 = lds[(kk + lh) * 64 + lr], b1 = lds[(kk + lh) * 64 + 32 + lr];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[j], b0, acc[0], 0, 0, 0);
        acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[j], b1, acc[1 % NACC], 0, 0, 0);
        acc[2 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(h[j], b0, acc[2 % NACC], 0, 0, 0);
        acc[3 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(h[j], b1, acc[3 % NACC], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (MODE == 1) { x = x * 1.0001f + 0.37f; y = y * 0.9999f - 0.11f; }
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 123.456f) out[0] = s;
}

int main() {
  float *g, *out;
  CK(hipMalloc(&g, 256 * 256 * 4)); CK(hipMalloc(&out, 4));
  float* h = (float*)malloc(256 * 256 * 4);
  unsigned s = 1u;
  for (int i = 0; i < 256 * 256; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 8388608.0f - 1.0f; }
  CK(hipMemcpy(g, h, 256 * 256 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto kern, int nacc, int threads, int blocks, int iters) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, g, out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double flop = (double)blocks * (threads / 64) * iters * 16.0 * nacc * 4096.0;
      if (rep == 2) printf("%-58s %2d waves/CU  %7.3f ms  -> %6.1f TFLOP/s\n", name, blocks / 256 * threads / 64, ms, flop / (ms * 1e-3) / 1e12);
    }
    fflush(stdout);
  };
  for (int it : {8000}) {
    printf("iters %d\n", it);
    run("constant operands, 4 acc, 256 thr x 1024 blocks", k_rate<0, 4>, 4, 256, 1024, it);
    run("changing operands, 4 acc, 256 thr x 1024 blocks", k_rate<1, 4>, 4, 256, 1024, it);
    run("changing operands, 2 acc, 512 thr x 256 blocks (8 waves/CU)", k_rate<1, 2>, 2, 512, 256, it);
    run("changing operands, 2 acc, 512 thr x 512 blocks (16 waves/CU)", k_rate<1, 2>, 2, 512, 512, it);
    run("LDS b + global-ring a, 2 acc, 512 thr x 256 blocks", k_rate<2, 2>, 2, 512, 256, it / 4);
    run("LDS b + global-ring a, 2 acc, 512 thr x 512 blocks", k_rate<2, 2>, 2, 512, 512, it / 4);
    run("  + LDS operands one k-pair ahead, 512 thr x 256 blocks", k_rate<3, 2>, 2, 512, 256, it / 4);
    run("  + LDS operands one k-pair ahead, 512 thr x 512 blocks", k_rate<3, 2>, 2, 512, 512, it / 4);
    run("2x2 fragments (4 acc), 256 thr x 256 blocks (4 waves/CU)", k_rate<4, 4>, 4, 256, 256, it / 4);
    run("2x2 fragments (4 acc), 256 thr x 512 blocks (8 waves/CU)", k_rate<4, 4>, 4, 256, 512, it / 4);
    run("2x2 fragments (4 acc), 256 thr x 1024 blocks (16 waves/CU)", k_rate<4, 4>, 4, 256, 1024, it / 4);
  }
  return 0;
}
