// Fused GroupNorm (+ FiLM scale/shift) (+ SiLU) for the score networks, forward and input-gradient (row a16 of SURVEY.md
// section 8: the score boundary stays PyTorch-ROCm, north_star; this file only fuses the normalisation / activation glue
// between its convolutions).
//
// Why: with the FFHQ U-Net in the loop a leapfrog step of 64 chains is 1.85 s of GPU time, of which MIOpen's fp32
// convolutions are 68 % and ATen's GroupNorm / SiLU / FiLM elementwise kernels 23 % (profiles/r02_kernel_stats_e2e_steady.csv):
// per ResBlock half, ATen runs moments + normalise + (x (1+scale)) + (+shift) + SiLU forward (5 reads, 4 writes of the
// activation) and SiLU' + FiLM' + two GroupNorm-backward kernels backward (7 reads, 3 writes).  Here:
//     forward   k_gn_stats (R x) -> k_gn_apply (R x, W y)                      2 reads, 1 write
//     backward  k_gn_bwd_stats (R x, dy) -> k_gn_bwd_apply (R x, dy, W dx)     4 reads, 1 write
// and nothing but x is kept for the backward (u = gn(x) a + b and SiLU'(u) are recomputed in registers).
//
//   u = ((x - mean_g) rstd_g gamma_c + beta_c) (1 + scale_bc) + shift_bc ;   y = act ? u sigmoid(u) : u
//   reference modules: guided_diffusion/nn.py GroupNorm32 + SiLU, unet_ffhq.py:310-321 (scale-shift norm);
//   ldm/modules/diffusionmodules/model.py:38-39 (Normalize, eps 1e-6) + nonlinearity.
// Statistics: fp64 sums of x and x^2 per (sample, group), split over `splits` blocks and reduced in a fixed order by
// the consumer (deterministic, no atomics); var = E[x^2] - mean^2 in fp64.  Only the input gradient is produced: the
// networks are frozen (requires_grad False) and the FiLM terms come from the time embedding, which does not depend on x.
#include "nhmc_common.h"

namespace {

constexpr int GN_TILE = NHMC_BLOCK * 2 * 4;      // elements per block: 2 float4 per thread

struct GnArgs {
  const float* gamma; const float* beta;          // [C]
  const float* film; int64_t film_stride;         // nullable: [B][film_stride] with scale at [c], shift at [C + c]
  const float* pre; int64_t pre_stride;           // nullable: x + pre[b * pre_stride + c] is what gets normalised (the bias of
                                                  // the convolution that produced x, + a per-sample embedding term): the
                                                  // producer then runs without its broadcast bias-add pass
  int C, G; int64_t hw; float eps; int act; int splits;
};

__device__ __forceinline__ float gn_sigmoid(float u) { return 1.0f / (1.0f + expf(-u)); }

// partial sums of one (sample, group): ws[(bg * splits + split) * NV + v]
template <bool BWD>
__global__ __launch_bounds__(NHMC_BLOCK) void k_gn_stats(const float4* __restrict__ x, const float4* __restrict__ dy,
                                                         const double* __restrict__ fwd_ws, GnArgs a,
                                                         double* __restrict__ ws) {
  const int bg = blockIdx.y, split = blockIdx.x;
  const int b = bg / a.G, g = bg % a.G, cpg = a.C / a.G;
  const int64_t n4 = (int64_t)cpg * a.hw / 4, base = (int64_t)bg * n4;
  const int64_t per = (n4 + a.splits - 1) / a.splits, lo = (int64_t)split * per, hi = min(n4, lo + per);
  float mean = 0.f, rstd = 0.f;
  if (BWD) {
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < a.splits; ++k) { s += fwd_ws[((int64_t)bg * a.splits + k) * 2]; ss += fwd_ws[((int64_t)bg * a.splits + k) * 2 + 1]; }
    const double n = (double)cpg * (double)a.hw, m = s / n;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(fmax(ss / n - m * m, 0.0) + (double)a.eps));
  }
  double s0 = 0.0, s1 = 0.0;
  for (int64_t q = lo + threadIdx.x; q < hi; q += NHMC_BLOCK) {
    const float4 xv = x[base + q];
    const float* xe = reinterpret_cast<const float*>(&xv);
    const int ch = g * cpg + (int)((q * 4) / a.hw);
    const float pb = a.pre ? a.pre[(int64_t)b * a.pre_stride + ch] : 0.0f;
    if (!BWD) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) { const float v = xe[c] + pb; t0 += v; t1 += v * v; }
      s0 += (double)t0; s1 += (double)t1;
    } else {
      const float4 dv = dy[base + q];
      const float* de = reinterpret_cast<const float*>(&dv);
      float ga = a.gamma[ch], be = a.beta[ch];
      if (a.film) { const float sc = 1.0f + a.film[(int64_t)b * a.film_stride + ch]; ga *= sc; be = be * sc + a.film[(int64_t)b * a.film_stride + a.C + ch]; }
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float xh = ((xe[c] + pb) - mean) * rstd;
        float du = de[c];
        if (a.act) { const float u = xh * ga + be, sg = gn_sigmoid(u); du = du * (sg * (1.0f + u * (1.0f - sg))); }
        const float dxh = du * ga;
        t0 += dxh; t1 += dxh * xh;
      }
      s0 += (double)t0; s1 += (double)t1;
    }
  }
  __shared__ double red[8];
  double v[2] = {s0, s1};
  nhmc_block_sum<2>(v, red);
  if (threadIdx.x == 0) { ws[((int64_t)bg * a.splits + split) * 2] = v[0]; ws[((int64_t)bg * a.splits + split) * 2 + 1] = v[1]; }
}

// BWD, `acc` (optional): a second gradient of x -- the block input also feeds the skip path -- added to the result here
// (fl(dx) + acc, what autograd's accumulation would compute in a pass of its own: 2 reads + 1 write saved).
template <bool BWD>
__global__ __launch_bounds__(NHMC_BLOCK) void k_gn_apply(const float4* __restrict__ x, const float4* __restrict__ dy,
                                                         const double* __restrict__ fwd_ws, const double* __restrict__ bwd_ws,
                                                         GnArgs a, float4* __restrict__ out, const float4* __restrict__ acc) {
  const int bg = blockIdx.y;
  const int b = bg / a.G, g = bg % a.G, cpg = a.C / a.G;
  const int64_t n4 = (int64_t)cpg * a.hw / 4, base = (int64_t)bg * n4;
  __shared__ float st[4];
  if (threadIdx.x == 0) {
    double s = 0.0, ss = 0.0, d0 = 0.0, d1 = 0.0;
    for (int k = 0; k < a.splits; ++k) {
      s += fwd_ws[((int64_t)bg * a.splits + k) * 2]; ss += fwd_ws[((int64_t)bg * a.splits + k) * 2 + 1];
      if (BWD) { d0 += bwd_ws[((int64_t)bg * a.splits + k) * 2]; d1 += bwd_ws[((int64_t)bg * a.splits + k) * 2 + 1]; }
    }
    const double n = (double)cpg * (double)a.hw, m = s / n;
    st[0] = (float)m;
    st[1] = (float)(1.0 / sqrt(fmax(ss / n - m * m, 0.0) + (double)a.eps));
    st[2] = (float)(d0 / n);
    st[3] = (float)(d1 / n);
  }
  __syncthreads();
  const float mean = st[0], rstd = st[1], m0 = st[2], m1 = st[3];
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * 2) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    const float4 xv = x[base + q];
    const float* xe = reinterpret_cast<const float*>(&xv);
    const int ch = g * cpg + (int)((q * 4) / a.hw);
    const float pb = a.pre ? a.pre[(int64_t)b * a.pre_stride + ch] : 0.0f;
    float ga = a.gamma[ch], be = a.beta[ch];
    if (a.film) { const float sc = 1.0f + a.film[(int64_t)b * a.film_stride + ch]; ga *= sc; be = be * sc + a.film[(int64_t)b * a.film_stride + a.C + ch]; }
    float4 o;
    float* oe = reinterpret_cast<float*>(&o);
    if (!BWD) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float u = (((xe[c] + pb) - mean) * rstd) * ga + be;
        oe[c] = a.act ? u * gn_sigmoid(u) : u;
      }
    } else {
      const float4 dv = dy[base + q];
      const float* de = reinterpret_cast<const float*>(&dv);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float xh = ((xe[c] + pb) - mean) * rstd;
        float du = de[c];
        if (a.act) { const float u = xh * ga + be, sg = gn_sigmoid(u); du = du * (sg * (1.0f + u * (1.0f - sg))); }
        oe[c] = rstd * ((du * ga - m0) - xh * m1);
      }
      if (acc) {
        const float4 av = acc[base + q];
        o.x = o.x + av.x; o.y = o.y + av.y; o.z = o.z + av.z; o.w = o.w + av.w;
      }
    }
    out[base + q] = o;
  }
}

// out = (h + bias_c) + other: the bias of the convolution that produced h folded into the residual add that follows it
// (unet_ffhq.py:321 `self.skip_connection(x) + h`): one 2R + 1W pass instead of a broadcast bias add (R + W) and an add.
__global__ __launch_bounds__(NHMC_BLOCK) void k_bias_add2(const float4* __restrict__ h, const float* __restrict__ bias,
                                                          const float4* __restrict__ other, float4* __restrict__ out,
                                                          int64_t n4, int64_t hw4, int C) {
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * 2) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    const float bc = bias[(int)((q / hw4) % C)];
    const float4 a = h[q], b = other[q];
    float4 o;
    o.x = (a.x + bc) + b.x; o.y = (a.y + bc) + b.y; o.z = (a.z + bc) + b.z; o.w = (a.w + bc) + b.w;
    out[q] = o;
  }
}

int gn_check(const void* x, const void* gamma, const void* beta, int n, int C, int G, int64_t hw, int splits) {
  if (!x || !gamma || !beta || n <= 0 || C <= 0 || G <= 0 || hw <= 0 || splits <= 0) return NHMC_ERR_ARG;
  if (C % G || ((int64_t)(C / G) * hw) % 4 || hw % 4 || (int64_t)n * G > 65535 || splits > 64) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(x)) return NHMC_ERR_ALIGN;
  return NHMC_OK;
}

}  // namespace

extern "C" int nhmc_gn_splits(int n, int channels, int groups, int64_t hw) {
  // enough blocks for the statistics pass: >= ~2048 blocks in flight, at least 2048 elements per block
  const int64_t per_group = (int64_t)(channels / groups) * hw;
  int64_t s = (2048 + (int64_t)n * groups - 1) / ((int64_t)n * groups);
  const int64_t cap = (per_group + GN_TILE - 1) / GN_TILE;
  if (s > cap) s = cap;
  if (s > 64) s = 64;
  return (int)(s < 1 ? 1 : s);
}

extern "C" int nhmc_gn_act_fwd(const float* x, const float* gamma, const float* beta, const float* film, int64_t film_stride,
                               const float* pre, int64_t pre_stride, float eps, int act, float* y, double* ws, int splits,
                               int n, int channels, int groups, int64_t hw, nhmc_stream_t stream) {
  int rc = gn_check(x, gamma, beta, n, channels, groups, hw, splits);
  if (rc) return rc;
  if (!y || !ws || !nhmc_aligned16(y)) return NHMC_ERR_ARG;
  const GnArgs a{gamma, beta, film, film_stride, pre, pre_stride, channels, groups, hw, eps, act, splits};
  const int64_t n4 = (int64_t)(channels / groups) * hw / 4;
  hipStream_t st = nhmc_s(stream);
  NHMC_LAUNCH(k_gn_stats<false>, dim3((unsigned)splits, (unsigned)(n * groups)), dim3(NHMC_BLOCK), 0, st, (const float4*)x,
              (const float4*)nullptr, (const double*)nullptr, a, ws);
  if ((rc = nhmc_launch_status())) return rc;
  NHMC_LAUNCH(k_gn_apply<false>, dim3((unsigned)((n4 + NHMC_BLOCK * 2 - 1) / (NHMC_BLOCK * 2)), (unsigned)(n * groups)),
              dim3(NHMC_BLOCK), 0, st, (const float4*)x, (const float4*)nullptr, ws, (const double*)nullptr, a, (float4*)y,
              (const float4*)nullptr);
  return nhmc_launch_status();
}

extern "C" int nhmc_gn_act_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* film,
                               int64_t film_stride, const float* pre, int64_t pre_stride, float eps, int act,
                               const double* fwd_ws, const float* dx_add, float* dx, double* ws, int splits, int n,
                               int channels, int groups, int64_t hw, nhmc_stream_t stream) {
  int rc = gn_check(x, gamma, beta, n, channels, groups, hw, splits);
  if (rc) return rc;
  if (!dy || !fwd_ws || !dx || !ws || !nhmc_aligned16(dy) || !nhmc_aligned16(dx) || !nhmc_aligned16(dx_add) || dx_add == dx)
    return NHMC_ERR_ARG;
  const GnArgs a{gamma, beta, film, film_stride, pre, pre_stride, channels, groups, hw, eps, act, splits};
  const int64_t n4 = (int64_t)(channels / groups) * hw / 4;
  hipStream_t st = nhmc_s(stream);
  NHMC_LAUNCH(k_gn_stats<true>, dim3((unsigned)splits, (unsigned)(n * groups)), dim3(NHMC_BLOCK), 0, st, (const float4*)x,
              (const float4*)dy, fwd_ws, a, ws);
  if ((rc = nhmc_launch_status())) return rc;
  NHMC_LAUNCH(k_gn_apply<true>, dim3((unsigned)((n4 + NHMC_BLOCK * 2 - 1) / (NHMC_BLOCK * 2)), (unsigned)(n * groups)),
              dim3(NHMC_BLOCK), 0, st, (const float4*)x, (const float4*)dy, fwd_ws, ws, a, (float4*)dx, (const float4*)dx_add);
  return nhmc_launch_status();
}

extern "C" int nhmc_bias_add2(const float* h, const float* bias, const float* other, float* out, int n, int channels,
                              int64_t hw, nhmc_stream_t stream) {
  if (!h || !bias || !other || !out || n <= 0 || channels <= 0 || hw <= 0) return NHMC_ERR_ARG;
  if (hw % 4) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(h) || !nhmc_aligned16(other) || !nhmc_aligned16(out)) return NHMC_ERR_ALIGN;
  const int64_t n4 = (int64_t)n * channels * hw / 4, blocks = (n4 + NHMC_BLOCK * 2 - 1) / (NHMC_BLOCK * 2);
  if (blocks > 0x7fffffff) return NHMC_ERR_SHAPE;
  NHMC_LAUNCH(k_bias_add2, dim3((unsigned)blocks), dim3(NHMC_BLOCK), 0, nhmc_s(stream), (const float4*)h, bias,
              (const float4*)other, (float4*)out, n4, hw / 4, channels);
  return nhmc_launch_status();
}
