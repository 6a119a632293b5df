// Shared device/host helpers for libnhmc (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nhmc.h"

#define NHMC_WAVE 64
#define NHMC_BLOCK 256            // 4 waves, one per SIMD of a CU
#define NHMC_VEC_PER_THREAD 2     // float4 per thread per stream -> 2048 elements per tile (measured best: see DESIGN.md)
#define NHMC_TILE (NHMC_BLOCK * NHMC_VEC_PER_THREAD * 4)

static inline bool nhmc_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline hipStream_t nhmc_s(nhmc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline thread_local hipError_t nhmc_last_hip_error = hipSuccess;   // last launch error of this thread (diagnostics)
static inline int nhmc_launch_status() {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return NHMC_OK;
  nhmc_last_hip_error = e;
  return NHMC_ERR_LAUNCH;
}
// hipGetLastError() is per-thread and sticky: the host framework around us leaves benign codes behind
// (e.g. hipErrorNotReady from event polling).  Clear it right before every launch so that
// nhmc_launch_status() reports this launch only.
#define NHMC_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// Wave-level sum over 64 lanes with shuffles; result valid in lane 0.
__device__ __forceinline__ double nhmc_wave_sum(double v) {
#pragma unroll
  for (int off = NHMC_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, NHMC_WAVE);
  return v;
}

// Block-level sum of up to NV values per thread (256 threads = 4 waves): shuffle inside the
// wave, 4 x NV doubles through LDS, fixed order -> deterministic.  Result valid in thread 0.
template <int NV>
__device__ __forceinline__ void nhmc_block_sum(double (&v)[NV], double* lds /* [4*NV] */) {
  const int lane = threadIdx.x & (NHMC_WAVE - 1), wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = nhmc_wave_sum(v[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) lds[wave * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (lds[0 * NV + i] + lds[1 * NV + i]) + (lds[2 * NV + i] + lds[3 * NV + i]);
  }
}

// Sum of an R x R block of pixels held as R float4 strips per lane by LANES = R / 4 adjacent, LANES-aligned lanes of a
// wave, in the REFERENCE'S ORDER: row by row, left to right, one running fp32 sum (SuperResolution.H is a matmul of the
// unfolded patch -- index kh * R + kw -- with a column of +-1/R, Hfuncs.py:188-200: a k-ascending FMA chain with a power-
// of-two weight, i.e. exactly this sequential sum times 1/R^2; so is avg_pool2d).  The running sum is handed from lane to
// lane with shuffles: R * LANES steps.  A pairwise tree over the lanes (round 2) adds the same numbers in another order
// and was 1e-7 off per evaluation, 4.7e-6 after a whole sr16 run.  Returns the block sum in every lane of the group.
template <int R, int LANES>
__device__ __forceinline__ float nhmc_block_sum_rowmajor(const float4 (&v)[R], int lane_in_group) {
  const int lane = threadIdx.x & (NHMC_WAVE - 1), base = lane - lane_in_group;
  float acc = 0.0f;
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
#pragma unroll
    for (int ln = 0; ln < LANES; ++ln) {
      const float prev = __shfl(acc, base + (ln == 0 ? LANES - 1 : ln - 1), NHMC_WAVE);
      if (lane_in_group == ln) {
        acc = (rr == 0 && ln == 0) ? 0.0f : prev;
        acc += v[rr].x; acc += v[rr].y; acc += v[rr].z; acc += v[rr].w;
      }
    }
  }
  return __shfl(acc, base + LANES - 1, NHMC_WAVE);
}

// Streaming (non-temporal) 16-byte accesses: every image element on this path is touched once per kernel, so
// loads and stores carry the nt hint (measured on MI355X, fused update at B = 64: 49.5 -> 41.8 us).
typedef float nhmc_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nhmc_ldnt(const float4* p) {
  const nhmc_v4f v = __builtin_nontemporal_load(reinterpret_cast<const nhmc_v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nhmc_stnt(float4* p, const float4& v) {
  nhmc_v4f w;
  w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
  __builtin_nontemporal_store(w, reinterpret_cast<nhmc_v4f*>(p));
}

__device__ __forceinline__ float nhmc_clip1(float v) { return fminf(fmaxf(v, -1.0f), 1.0f); }
__device__ __forceinline__ float nhmc_in1(float v) { return (v >= -1.0f && v <= 1.0f) ? 1.0f : 0.0f; }

// Square roots: plain sqrtf.  hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt makes sqrtf and the fp32 division
// IEEE correctly rounded on gfx950 (no fast-math flag is set in build.py); tests/test_hygiene_gpu.py sweeps all 1001
// alpha-bar table entries through the kernels against numpy's correctly rounded root.
