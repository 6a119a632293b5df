// DDIM alpha/beta mixing, forward and VJP (rows a8-a11 of SURVEY.md section 8).
// Replaces algos/unconditional.py:17-28 and the autograd nodes it generates
// (main_sampling.py:695,711).  Reads the score network's [B, 2C, H, W] output in place:
// only channels [0, C) are touched, so the learned-sigma half never moves through HBM.
//
// HBM-bound: forward R xt, R e[:C], W xt_next = 3T; backward R gout, xt, e[:C], W g_xt, g_e = 5T + C zeros.
// grid = (tiles, chains); the per-chain alpha-bars come from small device arrays and the square
// roots are taken in fp32 exactly as the reference takes them on its [n,1,1,1] tensors.
#include "nhmc_common.h"
#include <cstdlib>

namespace {

struct Coef { float c1, c2, c3, c4; };

__device__ __forceinline__ Coef coef(const float* at, const float* at_next, int chain) {
  const float a = at[chain], an = at_next[chain];
  Coef c;
  c.c1 = sqrtf(1.0f - a);    // (1 - at).sqrt()
  c.c2 = sqrtf(a);           // at.sqrt()
  c.c3 = sqrtf(an);          // at_next.sqrt()
  c.c4 = sqrtf(1.0f - an);   // (1 - at_next).sqrt()
  return c;
}

template <bool W_NEXT, bool W_X0, bool W_ADD>
__global__ __launch_bounds__(NHMC_BLOCK) void k_mix_fwd(
    const float4* __restrict__ xt, const float4* __restrict__ e, int64_t e_stride4,
    const float* __restrict__ at, const float* __restrict__ at_next, int final_clip,
    float4* __restrict__ xt_next, float4* __restrict__ x0_t, float4* __restrict__ add_up, int64_t n4) {
  const int chain = blockIdx.y;
  const Coef k = coef(at, at_next, chain);
  const int64_t base = (int64_t)chain * n4, ebase = (int64_t)chain * e_stride4;
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * NHMC_VEC_PER_THREAD) + threadIdx.x;
  float4 xv[NHMC_VEC_PER_THREAD], ev[NHMC_VEC_PER_THREAD];
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q < n4) { xv[i] = nhmc_ldnt(&xt[base + q]); ev[i] = nhmc_ldnt(&e[ebase + q]); }
  }
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    float4 o_next, o_x0, o_add;
    const float* xe = reinterpret_cast<const float*>(&xv[i]);
    const float* ee = reinterpret_cast<const float*>(&ev[i]);
    float* on = reinterpret_cast<float*>(&o_next);
    float* o0 = reinterpret_cast<float*>(&o_x0);
    float* oa = reinterpret_cast<float*>(&o_add);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float u = (xe[c] - ee[c] * k.c1) / k.c2;
      const float x0 = nhmc_clip1(u);
      const float add = k.c4 * ee[c];
      float nx = k.c3 * x0 + add;
      if (final_clip) nx = nhmc_clip1(nx);
      on[c] = nx; o0[c] = x0; oa[c] = add;
    }
    if (W_NEXT) nhmc_stnt(&xt_next[base + q], o_next);
    if (W_X0) nhmc_stnt(&x0_t[base + q], o_x0);
    if (W_ADD) nhmc_stnt(&add_up[base + q], o_add);
  }
}

__global__ __launch_bounds__(NHMC_BLOCK) void k_map_back(
    const float4* __restrict__ x0_t, const float4* __restrict__ add_up, const float* __restrict__ at_next,
    float4* __restrict__ xt_next, int64_t n4) {
  const int chain = blockIdx.y;
  const float c3 = sqrtf(at_next[chain]);
  const int64_t base = (int64_t)chain * n4;
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * NHMC_VEC_PER_THREAD) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    const float4 a = nhmc_ldnt(&x0_t[base + q]), b = nhmc_ldnt(&add_up[base + q]);
    float4 o;
    o.x = c3 * a.x + b.x; o.y = c3 * a.y + b.y; o.z = c3 * a.z + b.z; o.w = c3 * a.w + b.w;
    nhmc_stnt(&xt_next[base + q], o);
  }
}

template <bool HAS_G2, bool SPLIT>
__global__ __launch_bounds__(NHMC_BLOCK) void k_mix_bwd(
    const float4* __restrict__ gout, const float4* __restrict__ gout2, const float4* __restrict__ g_x0,
    const float4* __restrict__ xt,
    const float4* __restrict__ e, int64_t e_stride4, const float* __restrict__ at,
    const float* __restrict__ at_next, int final_clip, float4* __restrict__ g_xt, float4* __restrict__ g_e,
    int64_t n4, int fill_sigma) {
  const int chain = blockIdx.y;
  const Coef k = coef(at, at_next, chain);
  const int64_t base = (int64_t)chain * n4, ebase = (int64_t)chain * e_stride4;
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * NHMC_VEC_PER_THREAD) + threadIdx.x;
  float4 gv[NHMC_VEC_PER_THREAD], xv[NHMC_VEC_PER_THREAD], ev[NHMC_VEC_PER_THREAD], sv[NHMC_VEC_PER_THREAD];
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q < n4) {
      gv[i] = nhmc_ldnt(&gout[base + q]); xv[i] = nhmc_ldnt(&xt[base + q]); ev[i] = nhmc_ldnt(&e[ebase + q]);
      if (SPLIT) sv[i] = nhmc_ldnt(&g_x0[base + q]);
      if (HAS_G2) {
        const float4 h = nhmc_ldnt(&gout2[base + q]);
        gv[i].x += h.x; gv[i].y += h.y; gv[i].z += h.z; gv[i].w += h.w;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    float4 ox, oe;
    const float* ge = reinterpret_cast<const float*>(&gv[i]);
    const float* xe = reinterpret_cast<const float*>(&xv[i]);
    const float* ee = reinterpret_cast<const float*>(&ev[i]);
    const float* se = reinterpret_cast<const float*>(&sv[i]);
    float* gx = reinterpret_cast<float*>(&ox);
    float* gee = reinterpret_cast<float*>(&oe);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float u = (xe[c] - ee[c] * k.c1) / k.c2;
      float gin = ge[c];
      if (final_clip) gin = gin * nhmc_in1(k.c3 * nhmc_clip1(u) + k.c4 * ee[c]);
      // fused step: the gradient reaching x0 is gin*c3 (map_back); split surface: it is handed in
      const float g0 = SPLIT ? se[c] : gin * k.c3;
      const float gu = (g0 * nhmc_in1(u)) / k.c2;
      gx[c] = gu;
      gee[c] = k.c4 * gin + (-gu) * k.c1;
    }
    nhmc_stnt(&g_xt[base + q], ox);
    if (g_e) nhmc_stnt(&g_e[ebase + q], oe);                 // g_e == nullptr: the score carries no gradient
  }
  // learned-sigma channels of the score gradient are zero (the forward slices them away); a caller that keeps a
  // persistent, pre-zeroed g_e buffer passes fill_sigma = 0 and saves this T of writes
  const int64_t extra = (fill_sigma && g_e) ? e_stride4 - n4 : 0;
  if (extra > 0) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
      const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
      if (q < extra) nhmc_stnt(&g_e[ebase + n4 + q], z);
    }
  }
}

// Last DDIM step VJP fused with the inpainting data term (primary BASELINE config): the clipped decode
// xt_next = clip(c3*clip(u) + c4*e) is recomputed in registers (bit-identical to k_mix_fwd), the residual
// r = y[slot] - xt_next and the upstream gradient gin = -2 r are formed on the fly, so the separate data-term pass
// (R xt_next, W g) and this kernel's read of g disappear: -3T per leapfrog step and one launch.
// PX = true: whole-pixel mask (what inpaint_random / inpaint_box build): instead of the dense CHW -> y map (a stream of
// T per chain, L2-served) the kernel reads one 32-pixel mask word and its exclusive prefix count (16 KB of tables for
// 256 x 256) and derives y index = channels * rank(pixel) + channel with a popcount.
template <bool PX>
__global__ __launch_bounds__(NHMC_BLOCK) void k_mix_bwd_inpaint(
    const float4* __restrict__ xt, const float4* __restrict__ e, int64_t e_stride4, const float* __restrict__ at,
    const float* __restrict__ at_next, const float* __restrict__ y, const int4* __restrict__ slot,
    const uint32_t* __restrict__ mask_words, const int32_t* __restrict__ prefix, int channels, int64_t hw, int64_t m,
    float4* __restrict__ g_xt, float4* __restrict__ g_e, double* __restrict__ loss_ws, int64_t n4, int fill_sigma) {
  const int chain = blockIdx.y;
  const Coef k = coef(at, at_next, chain);
  const int64_t base = (int64_t)chain * n4, ebase = (int64_t)chain * e_stride4;
  const float* yb = y + (int64_t)chain * m;
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * NHMC_VEC_PER_THREAD) + threadIdx.x;
  double acc = 0.0;                                          // fp32 squares summed in fp64: independent of the tiling
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    const float4 xv = nhmc_ldnt(&xt[base + q]), ev = nhmc_ldnt(&e[ebase + q]);
    int4 sv;
    if (PX) {
      const int64_t el = q * 4;                              // first of 4 consecutive pixels of one channel (hw % 32 == 0)
      const int ch = (int)(el / hw);
      const int64_t p0 = el - (int64_t)ch * hw;
      const uint32_t word = mask_words[p0 >> 5];
      const int b0 = (int)(p0 & 31);
      int rank = prefix[p0 >> 5] + __popc(word & ((1u << b0) - 1u));
      int* so = reinterpret_cast<int*>(&sv);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool kept = (word >> (b0 + c)) & 1u;
        so[c] = kept ? rank * channels + ch : -1;
        rank += kept ? 1 : 0;
      }
    } else {
      sv = slot[q];
    }
    const float* xe = reinterpret_cast<const float*>(&xv);
    const float* ee = reinterpret_cast<const float*>(&ev);
    const int* se = reinterpret_cast<const int*>(&sv);
    float4 ox, oe;
    float* gx = reinterpret_cast<float*>(&ox);
    float* gee = reinterpret_cast<float*>(&oe);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float u = (xe[c] - ee[c] * k.c1) / k.c2;
      const float pre = k.c3 * nhmc_clip1(u) + k.c4 * ee[c];    // decode before the final clip
      float gin = 0.0f;
      if (se[c] >= 0) {
        const float r = yb[se[c]] - nhmc_clip1(pre);
        acc += (double)(r * r);
        gin = -(2.0f * r);
      }
      gin = gin * nhmc_in1(pre);                                 // final clip mask
      const float gu = ((gin * k.c3) * nhmc_in1(u)) / k.c2;
      gx[c] = gu;
      gee[c] = k.c4 * gin + (-gu) * k.c1;
    }
    nhmc_stnt(&g_xt[base + q], ox);
    nhmc_stnt(&g_e[ebase + q], oe);
  }
  const int64_t extra = fill_sigma ? e_stride4 - n4 : 0;
  if (extra > 0) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
      const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
      if (q < extra) nhmc_stnt(&g_e[ebase + n4 + q], z);
    }
  }
  __shared__ double red[4];
  double v[1] = {acc};
  nhmc_block_sum<1>(v, red);
  if (threadIdx.x == 0) loss_ws[(int64_t)chain * gridDim.x + blockIdx.x] = v[0];
}

// Whole-pixel-mask form, restructured for latency (round 2; the form above reached 4.6 TB/s = 0.57 of the HBM peak with
// no wasted traffic, i.e. it was latency / issue bound):
//   * grid = (tiles of one channel plane, channel, chain): the channel and the pixel index come from the block index,
//     no 64-bit division per float4;
//   * every load that does not depend on data is issued up front, mask tables FIRST: hardware returns loads in issue
//     order, so waiting for the (L2-resident, 16 KB) mask words leaves the 2 x VPT streaming loads of xt / e in flight;
//   * the y gathers depend on the mask only, so they are issued next -- before any xt / e value is needed -- as
//     predicated loads (8 % of the pixels are kept); the arithmetic then finds everything resident;
//   * VPT float4 per thread of each stream (16-byte non-temporal accesses, 1 KiB contiguous per wave instruction).
// Same arithmetic and bits as the form above (g_xt, g_e); loss partials: one per (tile, channel), fp64.
template <int VPT>
__global__ __launch_bounds__(NHMC_BLOCK) void k_mix_bwd_inpaint_px(
    const float4* __restrict__ xt, const float4* __restrict__ e, int64_t e_stride4, const float* __restrict__ at,
    const float* __restrict__ at_next, const float* __restrict__ y, const uint32_t* __restrict__ mask_words,
    const int32_t* __restrict__ prefix, int channels, int64_t hw4, int64_t m, float4* __restrict__ g_xt,
    float4* __restrict__ g_e, double* __restrict__ loss_ws, int fill_sigma) {
  const int chain = blockIdx.z, ch = blockIdx.y;
  const Coef k = coef(at, at_next, chain);
  const int64_t plane = ((int64_t)chain * channels + ch) * hw4;            // float4 offset of this channel plane
  const int64_t eplane = (int64_t)chain * e_stride4 + (int64_t)ch * hw4;
  const float* yb = y + (int64_t)chain * m + ch;                            // y index = channels * rank(pixel) + channel
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * VPT) + threadIdx.x;
  uint32_t word[VPT];
  int rank0[VPT];
  float4 xv[VPT], ev[VPT];
  bool ok[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    ok[i] = q < hw4;
    word[i] = ok[i] ? mask_words[q >> 3] : 0u;                              // 8 float4 = 32 pixels per mask word
    rank0[i] = ok[i] ? prefix[q >> 3] : 0;
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (ok[i]) { xv[i] = nhmc_ldnt(&xt[plane + q]); ev[i] = nhmc_ldnt(&e[eplane + q]); }
  }
  float yv[VPT][4];
  unsigned kept[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    const int b0 = (int)(q & 7) * 4;
    kept[i] = (word[i] >> b0) & 15u;
    int rank = rank0[i] + __popc(word[i] & ((1u << b0) - 1u));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool kp = (kept[i] >> c) & 1u;
      yv[i][c] = kp ? yb[(int64_t)rank * channels] : 0.0f;
      rank += kp ? 1 : 0;
    }
  }
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    if (!ok[i]) continue;
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    const float* xe = reinterpret_cast<const float*>(&xv[i]);
    const float* ee = reinterpret_cast<const float*>(&ev[i]);
    float4 ox, oe;
    float* gx = reinterpret_cast<float*>(&ox);
    float* gee = reinterpret_cast<float*>(&oe);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float u = (xe[c] - ee[c] * k.c1) / k.c2;
      const float pre = k.c3 * nhmc_clip1(u) + k.c4 * ee[c];    // decode before the final clip
      float gin = 0.0f;
      if ((kept[i] >> c) & 1u) {
        const float r = yv[i][c] - nhmc_clip1(pre);
        acc += (double)(r * r);
        gin = -(2.0f * r);
      }
      gin = gin * nhmc_in1(pre);                                 // final clip mask
      const float gu = ((gin * k.c3) * nhmc_in1(u)) / k.c2;
      gx[c] = gu;
      gee[c] = k.c4 * gin + (-gu) * k.c1;
    }
    nhmc_stnt(&g_xt[plane + q], ox);
    nhmc_stnt(&g_e[eplane + q], oe);
    // learned-sigma plane paired with this channel (e_channels = 2 C): zero-filled on request
    if (fill_sigma && e_stride4 > (int64_t)channels * hw4)
      nhmc_stnt(&g_e[eplane + (int64_t)channels * hw4 + q], make_float4(0.f, 0.f, 0.f, 0.f));
  }
  __shared__ double red[4];
  double v[1] = {acc};
  nhmc_block_sum<1>(v, red);
  if (threadIdx.x == 0) loss_ws[((int64_t)chain * gridDim.y + ch) * gridDim.x + blockIdx.x] = v[0];
}

// Last DDIM step VJP fused with the super-resolution (r x r block mean) data term.  Work item = one float4 strip of
// one output row (as k_sr in data_term.hip): a thread walks the R rows of its strip twice -- pass 1 recomputes the
// clipped decode from (xt, e), sums it and keeps two mask bits per element; pass 2 turns the block residual into
// g_xt / g_e.  Same summation order as k_mix_fwd + k_sr + k_mix_bwd, hence the same bits, with -3T of traffic.
template <int R>
__global__ __launch_bounds__(NHMC_BLOCK) void k_mix_bwd_sr(
    const float4* __restrict__ xt, const float4* __restrict__ e, int e_channels, const float* __restrict__ at,
    const float* __restrict__ at_next, const float* __restrict__ y, float4* __restrict__ g_xt,
    float4* __restrict__ g_e, double* __restrict__ loss_ws, int dim, int channels) {
  const int chain = blockIdx.y;
  const Coef k = coef(at, at_next, chain);
  const int w4 = dim / 4, yd = dim / R;
  const int64_t items = (int64_t)channels * yd * w4;
  const int64_t item = (int64_t)blockIdx.x * NHMC_BLOCK + threadIdx.x;
  const bool live = item < items;
  const int64_t it = live ? item : 0;
  const int s = (int)(it % w4);
  const int i = (int)((it / w4) % yd);
  const int plane = (int)(it / ((int64_t)w4 * yd));
  const int64_t xrow0 = ((int64_t)chain * channels + plane) * (int64_t)dim * w4 + (int64_t)i * R * w4 + s;
  const int64_t erow0 = ((int64_t)chain * e_channels + plane) * (int64_t)dim * w4 + (int64_t)i * R * w4 + s;
  const float* yplane = y + ((int64_t)chain * channels + plane) * (int64_t)yd * yd;
  constexpr int LANES = R >= 4 ? R / 4 : 1;     // strips that share one block
  constexpr int BPS = R >= 4 ? 1 : 4 / R;       // blocks per strip (R = 2 -> 2)
  const float inv = 1.0f / (float)(R * R);

  float bs[BPS], yv[BPS];
#pragma unroll
  for (int b = 0; b < BPS; ++b) {
    bs[b] = 0.0f;
    // the observation of this thread's block(s): its address depends on indices only, so it is requested FIRST --
    // loads return in issue order, and issued after the 2R streaming loads it would be waited for behind all of them
    const int j = BPS == 1 ? (s * 4) / R : s * BPS + b;
    yv[b] = live ? yplane[(int64_t)i * yd + j] : 0.0f;
  }
  unsigned in_pre = 0u, in_u = 0u;              // bit rr*4+c: 1[-1 <= pre <= 1], 1[-1 <= u <= 1]   (R <= 8 rows per word)
  unsigned in_pre_hi = 0u, in_u_hi = 0u;        // rows 8..15 (R = 16)
  float4 dec[LANES > 1 ? R : 1];                // R >= 8: the clipped decodes, for the row-major sum across the strips
  if constexpr (LANES > 1) {
#pragma unroll
    for (int rr = 0; rr < R; ++rr) dec[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (live) {
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const float4 xv = nhmc_ldnt(&xt[xrow0 + (int64_t)rr * w4]), ev = nhmc_ldnt(&e[erow0 + (int64_t)rr * w4]);
      const float* xe = reinterpret_cast<const float*>(&xv);
      const float* ee = reinterpret_cast<const float*>(&ev);
      float cl[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float u = (xe[c] - ee[c] * k.c1) / k.c2;
        const float pre = k.c3 * nhmc_clip1(u) + k.c4 * ee[c];
        cl[c] = nhmc_clip1(pre);
        if (LANES == 1) bs[BPS == 1 ? 0 : c / R] += cl[c];
        const unsigned bit = 1u << (((rr & 7) << 2) + c);
        if (rr < 8) { if (pre >= -1.0f && pre <= 1.0f) in_pre |= bit; if (u >= -1.0f && u <= 1.0f) in_u |= bit; }
        else        { if (pre >= -1.0f && pre <= 1.0f) in_pre_hi |= bit; if (u >= -1.0f && u <= 1.0f) in_u_hi |= bit; }
      }
      if constexpr (LANES > 1) dec[rr] = make_float4(cl[0], cl[1], cl[2], cl[3]);
    }
  }
  if constexpr (LANES > 1) bs[0] = nhmc_block_sum_rowmajor<R, LANES>(dec, s % LANES);   // the reference's order (nhmc_common.h)
  float acc = 0.0f, resid[BPS];
#pragma unroll
  for (int b = 0; b < BPS; ++b) {
    resid[b] = live ? yv[b] - bs[b] * inv : 0.0f;
    if (live && (LANES == 1 || (s % LANES) == 0)) acc += resid[b] * resid[b];
  }
  if (live) {
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      float4 ox, oe;
      float* gx = reinterpret_cast<float*>(&ox);
      float* gee = reinterpret_cast<float*>(&oe);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned bit = 1u << (((rr & 7) << 2) + c);
        const float mp = ((rr < 8 ? in_pre : in_pre_hi) & bit) ? 1.0f : 0.0f;
        const float mu = ((rr < 8 ? in_u : in_u_hi) & bit) ? 1.0f : 0.0f;
        float gin = (-(2.0f * resid[BPS == 1 ? 0 : c / R])) * inv;   // data-term gradient (k_sr, no clip mask there)
        gin = gin * mp;                                               // final clip mask
        const float gu = ((gin * k.c3) * mu) / k.c2;
        gx[c] = gu;
        gee[c] = k.c4 * gin + (-gu) * k.c1;
      }
      nhmc_stnt(&g_xt[xrow0 + (int64_t)rr * w4], ox);
      nhmc_stnt(&g_e[erow0 + (int64_t)rr * w4], oe);
    }
  }
  __shared__ double red[4];
  double v[1] = {(double)acc};
  nhmc_block_sum<1>(v, red);
  if (threadIdx.x == 0) loss_ws[(int64_t)chain * gridDim.x + blockIdx.x] = v[0];
}

// The same for R = 4 (BASELINE configs[2]) with ONE float4 per thread and stream.  The form above gives a thread the 4
// rows of its strip (8 streaming loads, 16 elements with two correctly rounded divisions each before its first store):
// measured 37.9 us at 64 chains = 0.67 of the HBM peak, where the inpainting kernel, same traffic and arithmetic with
// one float4 per thread, reaches 0.74.  Here the 4 waves of a block take the 4 rows of 64 blocks-of-pixels: lane = strip
// (a wave instruction still covers 1 KiB of one image row when dim = 256), wave = row inside the block.  The clipped
// decodes meet in LDS (4 KB) and every thread adds the 16 values of its block in the order of the form above -- row by
// row, left to right -- so block means, residuals, loss and gradients are the same bits.  grid = (tiles of a plane,
// plane, chain): no 64-bit division.
__global__ __launch_bounds__(NHMC_BLOCK) void k_mix_bwd_sr4(
    const float4* __restrict__ xt, const float4* __restrict__ e, int e_channels, const float* __restrict__ at,
    const float* __restrict__ at_next, const float* __restrict__ y, float4* __restrict__ g_xt,
    float4* __restrict__ g_e, double* __restrict__ loss_ws, int dim, int channels) {
  constexpr int R = 4;
  __shared__ float4 dec[R][NHMC_WAVE];
  const int chain = blockIdx.z, plane = blockIdx.y;
  const int lane = threadIdx.x & 63, rr = threadIdx.x >> 6;
  const int w4 = dim / 4, yd = dim / R;
  const int items = yd * w4;                                   // (output row, strip) pairs of one plane
  const int item = blockIdx.x * NHMC_WAVE + lane;
  const bool live = item < items;
  const int it = live ? item : 0;
  const int i = it / w4, s = it - i * w4;                     // R = 4: strip s is output column s
  const float yv = live ? y[((int64_t)chain * channels + plane) * (int64_t)yd * yd + (int64_t)i * yd + s] : 0.0f;
  const int64_t row = (int64_t)(i * R + rr) * w4 + s;
  const int64_t xoff = ((int64_t)chain * channels + plane) * (int64_t)dim * w4 + row;
  const int64_t eoff = ((int64_t)chain * e_channels + plane) * (int64_t)dim * w4 + row;
  const Coef k = coef(at, at_next, chain);
  float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), ev = xv;
  if (live) { xv = nhmc_ldnt(&xt[xoff]); ev = nhmc_ldnt(&e[eoff]); }
  const float* xe = reinterpret_cast<const float*>(&xv);
  const float* ee = reinterpret_cast<const float*>(&ev);
  float mp[4], mu[4];
  float4 cl;
  float* cle = reinterpret_cast<float*>(&cl);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float u = (xe[c] - ee[c] * k.c1) / k.c2;
    const float pre = k.c3 * nhmc_clip1(u) + k.c4 * ee[c];
    cle[c] = nhmc_clip1(pre);
    mp[c] = nhmc_in1(pre);
    mu[c] = nhmc_in1(u);
  }
  dec[rr][lane] = cl;
  __syncthreads();
  float bs = 0.0f;
#pragma unroll
  for (int r2 = 0; r2 < R; ++r2) {
    const float4 d = dec[r2][lane];
    bs += d.x; bs += d.y; bs += d.z; bs += d.w;
  }
  const float inv = 1.0f / (float)(R * R);
  const float resid = live ? yv - bs * inv : 0.0f;
  const float acc = rr == 0 ? resid * resid : 0.0f;            // one thread per block of pixels contributes to the loss
  if (live) {
    float4 ox, oe;
    float* gx = reinterpret_cast<float*>(&ox);
    float* gee = reinterpret_cast<float*>(&oe);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float gin = (-(2.0f * resid)) * inv;
      gin = gin * mp[c];
      const float gu = ((gin * k.c3) * mu[c]) / k.c2;
      gx[c] = gu;
      gee[c] = k.c4 * gin + (-gu) * k.c1;
    }
    nhmc_stnt(&g_xt[xoff], ox);
    nhmc_stnt(&g_e[eoff], oe);
  }
  __shared__ double red[4];
  double v[1] = {(double)acc};
  nhmc_block_sum<1>(v, red);
  if (threadIdx.x == 0) loss_ws[((int64_t)chain * gridDim.y + plane) * gridDim.x + blockIdx.x] = v[0];
}

bool bad_shape(int n_chains, int channels, int64_t hw, int e_channels) {
  return n_chains <= 0 || n_chains > 65535 || channels <= 0 || hw <= 0 ||
         (e_channels != channels && e_channels != 2 * channels);
}

}  // namespace

extern "C" int nhmc_ddim_mix_fwd(const float* xt, const float* e, int e_channels, const float* at,
                                 const float* at_next, int final_clip, float* xt_next, float* x0_t,
                                 float* add_up, int n_chains, int channels, int64_t hw, nhmc_stream_t stream) {
  if (!xt || !e || !at || !at_next || (!xt_next && !x0_t && !add_up)) return NHMC_ERR_ARG;
  if (bad_shape(n_chains, channels, hw, e_channels)) return NHMC_ERR_SHAPE;
  const int64_t n_elem = (int64_t)channels * hw;
  if ((n_elem & 3) || !nhmc_aligned16(xt) || !nhmc_aligned16(e) || !nhmc_aligned16(xt_next) ||
      !nhmc_aligned16(x0_t) || !nhmc_aligned16(add_up))
    return NHMC_ERR_ALIGN;
  const int64_t n4 = n_elem / 4, es4 = (int64_t)e_channels * hw / 4;
  dim3 grid((unsigned)nhmc_leapfrog_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  hipStream_t st = nhmc_s(stream);
#define NHMC_FWD(A, B, C)                                                                             \
  NHMC_LAUNCH((k_mix_fwd<A, B, C>), grid, block, 0, st, (const float4*)xt, (const float4*)e, es4, \
                     at, at_next, final_clip, (float4*)xt_next, (float4*)x0_t, (float4*)add_up, n4)
  const int sel = (xt_next ? 4 : 0) | (x0_t ? 2 : 0) | (add_up ? 1 : 0);
  switch (sel) {
    case 4: NHMC_FWD(true, false, false); break;
    case 3: NHMC_FWD(false, true, true); break;
    case 7: NHMC_FWD(true, true, true); break;
    case 6: NHMC_FWD(true, true, false); break;
    case 5: NHMC_FWD(true, false, true); break;
    case 2: NHMC_FWD(false, true, false); break;
    default: NHMC_FWD(false, false, true); break;
  }
#undef NHMC_FWD
  return nhmc_launch_status();
}

extern "C" int nhmc_ddim_map_back(const float* x0_t, const float* add_up, const float* at_next, float* xt_next,
                                  int n_chains, int64_t n_elem, nhmc_stream_t stream) {
  if (!x0_t || !add_up || !at_next || !xt_next || n_chains <= 0 || n_elem <= 0) return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  if ((n_elem & 3) || !nhmc_aligned16(x0_t) || !nhmc_aligned16(add_up) || !nhmc_aligned16(xt_next))
    return NHMC_ERR_ALIGN;
  dim3 grid((unsigned)nhmc_leapfrog_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  NHMC_LAUNCH(k_map_back, grid, block, 0, nhmc_s(stream), (const float4*)x0_t, (const float4*)add_up,
                     at_next, (float4*)xt_next, n_elem / 4);
  return nhmc_launch_status();
}

extern "C" int nhmc_ddim_mix_bwd(const float* gout, const float* gout2, const float* g_x0, const float* xt,
                                 const float* e, int e_channels, const float* at, const float* at_next, int final_clip,
                                 float* g_xt, float* g_e, int fill_sigma, int n_chains, int channels, int64_t hw,
                                 nhmc_stream_t stream) {
  if (!gout || !xt || !e || !at || !at_next || !g_xt) return NHMC_ERR_ARG;      // g_e may be NULL (not wanted)
  if (bad_shape(n_chains, channels, hw, e_channels)) return NHMC_ERR_SHAPE;
  const int64_t n_elem = (int64_t)channels * hw;
  if (g_x0 && (gout2 || final_clip)) return NHMC_ERR_ARG;
  if ((n_elem & 3) || !nhmc_aligned16(gout) || !nhmc_aligned16(gout2) || !nhmc_aligned16(g_x0) || !nhmc_aligned16(xt) ||
      !nhmc_aligned16(e) || !nhmc_aligned16(g_xt) || !nhmc_aligned16(g_e))
    return NHMC_ERR_ALIGN;
  const int64_t n4 = n_elem / 4, es4 = (int64_t)e_channels * hw / 4;
  dim3 grid((unsigned)nhmc_leapfrog_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  hipStream_t st = nhmc_s(stream);
#define NHMC_BWD(G2, SP)                                                                                  \
  NHMC_LAUNCH((k_mix_bwd<G2, SP>), grid, block, 0, st, (const float4*)gout, (const float4*)gout2,     \
                     (const float4*)g_x0, (const float4*)xt, (const float4*)e, es4, at, at_next, final_clip, \
                     (float4*)g_xt, (float4*)g_e, n4, fill_sigma)
  if (g_x0) NHMC_BWD(false, true);
  else if (gout2) NHMC_BWD(true, false);
  else NHMC_BWD(false, false);
#undef NHMC_BWD
  return nhmc_launch_status();
}

// Last-step VJP fused with the inpainting data term; loss partials: nhmc_leapfrog_tiles(n_elem) per chain.
extern "C" int nhmc_ddim_mix_bwd_inpaint(const float* xt, const float* e, int e_channels, const float* at,
                                         const float* at_next, const float* y, const int32_t* slot, int64_t m,
                                         float* g_xt, float* g_e, int fill_sigma, double* loss_ws, int n_chains,
                                         int channels, int64_t hw, nhmc_stream_t stream) {
  if (!xt || !e || !at || !at_next || !y || !slot || !g_xt || !g_e || !loss_ws || m <= 0) return NHMC_ERR_ARG;
  if (bad_shape(n_chains, channels, hw, e_channels)) return NHMC_ERR_SHAPE;
  const int64_t n_elem = (int64_t)channels * hw;
  if ((n_elem & 3) || !nhmc_aligned16(xt) || !nhmc_aligned16(e) || !nhmc_aligned16(slot) || !nhmc_aligned16(g_xt) ||
      !nhmc_aligned16(g_e))
    return NHMC_ERR_ALIGN;
  dim3 grid((unsigned)nhmc_leapfrog_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  NHMC_LAUNCH(k_mix_bwd_inpaint<false>, grid, block, 0, nhmc_s(stream), (const float4*)xt, (const float4*)e,
              (int64_t)e_channels * hw / 4, at, at_next, y, (const int4*)slot, (const uint32_t*)nullptr,
              (const int32_t*)nullptr, channels, hw, m, (float4*)g_xt, (float4*)g_e, loss_ws, n_elem / 4, fill_sigma);
  return nhmc_launch_status();
}

constexpr int PX_VPT = 1;       // float4 per thread and stream (tools/inpaint_bench.hip on MI355X, B = 64: 34.8 us at 1, 36.6 at 2, 38.2 at 4, 39.2 at 8; round-1 form 42.3)

extern "C" int nhmc_inpaint_px_tiles(int channels, int64_t hw) {
  const int64_t hw4 = hw / 4;
  return (int)(channels * ((hw4 + NHMC_BLOCK * PX_VPT - 1) / (NHMC_BLOCK * PX_VPT)));
}

extern "C" int nhmc_ddim_mix_bwd_inpaint_px(const float* xt, const float* e, int e_channels, const float* at,
                                            const float* at_next, const float* y, const uint32_t* mask_words,
                                            const int32_t* prefix, int64_t m, float* g_xt, float* g_e, int fill_sigma,
                                            double* loss_ws, int n_chains, int channels, int64_t hw,
                                            nhmc_stream_t stream) {
  if (!xt || !e || !at || !at_next || !y || !mask_words || !prefix || !g_xt || !g_e || !loss_ws || m <= 0)
    return NHMC_ERR_ARG;
  if (bad_shape(n_chains, channels, hw, e_channels) || (hw & 31) || channels > 65535) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(xt) || !nhmc_aligned16(e) || !nhmc_aligned16(g_xt) || !nhmc_aligned16(g_e)) return NHMC_ERR_ALIGN;
  const int64_t hw4 = hw / 4;
  dim3 grid((unsigned)((hw4 + NHMC_BLOCK * PX_VPT - 1) / (NHMC_BLOCK * PX_VPT)), (unsigned)channels, (unsigned)n_chains);
  NHMC_LAUNCH(k_mix_bwd_inpaint_px<PX_VPT>, grid, dim3(NHMC_BLOCK), 0, nhmc_s(stream), (const float4*)xt,
              (const float4*)e, (int64_t)e_channels * hw / 4, at, at_next, y, mask_words, prefix, channels, hw4, m,
              (float4*)g_xt, (float4*)g_e, loss_ws, fill_sigma);
  return nhmc_launch_status();
}

extern "C" int nhmc_sr_vjp_tiles(int channels, int dim, int ratio) {
  if (ratio == 4 && channels <= 65535) return channels * (((dim / 4) * (dim / 4) + NHMC_WAVE - 1) / NHMC_WAVE);
  return nhmc_sr_tiles(channels, dim, ratio);
}

extern "C" int nhmc_ddim_mix_bwd_sr(const float* xt, const float* e, int e_channels, const float* at,
                                    const float* at_next, const float* y, int ratio, float* g_xt, float* g_e,
                                    double* loss_ws, int n_chains, int channels, int dim, nhmc_stream_t stream) {
  if (!xt || !e || !at || !at_next || !y || !g_xt || !g_e || !loss_ws) return NHMC_ERR_ARG;
  if (dim <= 0 || (dim % 4) || bad_shape(n_chains, channels, (int64_t)dim * dim, e_channels)) return NHMC_ERR_SHAPE;
  if (!(ratio == 2 || ratio == 4 || ratio == 8 || ratio == 16) || (dim % ratio)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(xt) || !nhmc_aligned16(e) || !nhmc_aligned16(g_xt) || !nhmc_aligned16(g_e)) return NHMC_ERR_ALIGN;
  if (ratio == 4 && channels <= 65535) {
    const int per_plane = ((dim / 4) * (dim / 4) + NHMC_WAVE - 1) / NHMC_WAVE;
    NHMC_LAUNCH(k_mix_bwd_sr4, dim3((unsigned)per_plane, (unsigned)channels, (unsigned)n_chains), dim3(NHMC_BLOCK), 0,
                nhmc_s(stream), (const float4*)xt, (const float4*)e, e_channels, at, at_next, y, (float4*)g_xt,
                (float4*)g_e, loss_ws, dim, channels);
    return nhmc_launch_status();
  }
  dim3 grid((unsigned)nhmc_sr_tiles(channels, dim, ratio), (unsigned)n_chains), block(NHMC_BLOCK);
#define NHMC_BSR(R)                                                                                              \
  NHMC_LAUNCH(k_mix_bwd_sr<R>, grid, block, 0, nhmc_s(stream), (const float4*)xt, (const float4*)e, e_channels, \
              at, at_next, y, (float4*)g_xt, (float4*)g_e, loss_ws, dim, channels)
  switch (ratio) {
    case 2: NHMC_BSR(2); break;
    case 4: NHMC_BSR(4); break;
    case 8: NHMC_BSR(8); break;
    default: NHMC_BSR(16); break;
  }
#undef NHMC_BSR
  return nhmc_launch_status();
}
