// Colorization operator (row f.3 of SURVEY.md section 8: remaining linear operators on the data-term interface).
// Replaces obs_functions/Hfuncs.py:655-695.  With (u, s, V) the SVD of the 1 x C row [0.3333 0.3334 0.3333] and
// v_c = V[c,0], per pixel
//     H x   = u * (s * ((v_0 x_0 + v_1 x_1) + v_2 x_2))        (Vt: a [C x C] @ [C x 1] matmul per pixel, torch sums
//                                                               its K = C products left to right, each rounded; :673-680,
//                                                               then singulars *, then U)
//     H^T y = v_c * (s * (u * y)),     H^+ y = v_c * ((u * y) * (1 / s))               (:65-90 composed with :682-695)
//     d loss / d x_c = v_c * (s * (u * (-(2 r))))               (autograd of the line above, in its order)
// in exactly these roundings, so the operator, the data term and its gradient reproduce torch's CPU bits (G15 replay).
// One thread owns a float4 of pixels and reads the C channel planes at the same offset: coalesced 16-byte accesses,
// nothing to stage or shuffle.  HBM-bound: data term R xt (T) + R y (T/C) + W g (T).
#include "nhmc_common.h"

namespace {

constexpr int MAXC = 4;
struct W { float w[MAXC]; float s, u; };     // w[c] = V[c,0]; singular value; U[0,0] (= +-1)

// MODE 0: data term, 1: H, 2: H^T, 3: H^+
template <int MODE>
__global__ __launch_bounds__(NHMC_BLOCK) void k_color(const float4* __restrict__ xin, const float4* __restrict__ yin,
                                                      float4* __restrict__ out, W wt, int channels, int apply_clip,
                                                      double* __restrict__ loss_ws, int64_t hw4) {
  const int chain = blockIdx.y;
  const int64_t q = (int64_t)blockIdx.x * NHMC_BLOCK + threadIdx.x;
  const bool live = q < hw4;
  float acc = 0.0f;
  if (live) {
    const int64_t xbase = (int64_t)chain * channels * hw4, ybase = (int64_t)chain * hw4;
    if (MODE >= 2) {
      const float4 y = nhmc_ldnt(&yin[ybase + q]);
      const float yv[4] = {y.x, y.y, y.z, y.w};
      float t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = MODE == 2 ? wt.s * (wt.u * yv[k]) : (wt.u * yv[k]) * (1.0f / wt.s);   // :80-90 multiplies by 1 / s
      for (int c = 0; c < channels; ++c) {
        float4 o;
        o.x = wt.w[c] * t[0]; o.y = wt.w[c] * t[1]; o.z = wt.w[c] * t[2]; o.w = wt.w[c] * t[3];
        nhmc_stnt(&out[xbase + (int64_t)c * hw4 + q], o);
      }
    } else {
      float4 xv[MAXC];
      float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        if (c < channels) {
          xv[c] = nhmc_ldnt(&xin[xbase + (int64_t)c * hw4 + q]);
          const float* e = reinterpret_cast<const float*>(&xv[c]);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float term = wt.w[c] * ((MODE == 0 && apply_clip) ? nhmc_clip1(e[k]) : e[k]);
            s[k] = c == 0 ? term : s[k] + term;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] = wt.u * (wt.s * s[k]);
      if (MODE == 1) {
        nhmc_stnt(&out[ybase + q], make_float4(s[0], s[1], s[2], s[3]));
      } else {
        const float4 y = nhmc_ldnt(&yin[ybase + q]);
        const float r[4] = {y.x - s[0], y.y - s[1], y.z - s[2], y.w - s[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += r[k] * r[k];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          if (c < channels) {
            const float* e = reinterpret_cast<const float*>(&xv[c]);
            float4 o;
            float* oe = reinterpret_cast<float*>(&o);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float gr = wt.w[c] * (wt.s * (wt.u * (-(2.0f * r[k]))));
              if (apply_clip) gr = gr * nhmc_in1(e[k]);
              oe[k] = gr;
            }
            nhmc_stnt(&out[xbase + (int64_t)c * hw4 + q], o);
          }
        }
      }
    }
  }
  if (MODE == 0) {
    __shared__ double red[4];
    double v[1] = {(double)acc};
    nhmc_block_sum<1>(v, red);
    if (threadIdx.x == 0) loss_ws[(int64_t)chain * gridDim.x + blockIdx.x] = v[0];
  }
}

// Last DDIM step VJP fused with the colorization data term: the clipped decode of every channel is recomputed from
// (xt, e) in registers, weighted into the grey value, and the residual goes straight through the step's VJP
// (same op order as k_mix_fwd + k_color<0> + k_mix_bwd, hence the same bits; -3T of traffic).
__global__ __launch_bounds__(NHMC_BLOCK) void k_mix_bwd_color(
    const float4* __restrict__ xt, const float4* __restrict__ e, int e_channels, const float* __restrict__ at,
    const float* __restrict__ at_next, const float4* __restrict__ yin, float4* __restrict__ g_xt,
    float4* __restrict__ g_e, W wt, int channels, double* __restrict__ loss_ws, int64_t hw4) {
  const int chain = blockIdx.y;
  const int64_t q = (int64_t)blockIdx.x * NHMC_BLOCK + threadIdx.x;
  const float a = at[chain], an = at_next[chain];
  const float c1 = sqrtf(1.0f - a), c2 = sqrtf(a), c3 = sqrtf(an), c4 = sqrtf(1.0f - an);
  float acc = 0.0f;
  if (q < hw4) {
    const int64_t xbase = (int64_t)chain * channels * hw4, ebase = (int64_t)chain * e_channels * hw4;
    float pre[MAXC][4], uu[MAXC][4];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < channels) {
        const float4 xv = nhmc_ldnt(&xt[xbase + (int64_t)c * hw4 + q]), ev = nhmc_ldnt(&e[ebase + (int64_t)c * hw4 + q]);
        const float* xe = reinterpret_cast<const float*>(&xv);
        const float* ee = reinterpret_cast<const float*>(&ev);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uu[c][k] = (xe[k] - ee[k] * c1) / c2;
          pre[c][k] = c3 * nhmc_clip1(uu[c][k]) + c4 * ee[k];
          const float term = wt.w[c] * nhmc_clip1(pre[c][k]);
          s[k] = c == 0 ? term : s[k] + term;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = wt.u * (wt.s * s[k]);
    const float4 y = nhmc_ldnt(&yin[(int64_t)chain * hw4 + q]);
    const float r[4] = {y.x - s[0], y.y - s[1], y.z - s[2], y.w - s[3]};
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += r[k] * r[k];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < channels) {
        float4 ox, oe;
        float* gx = reinterpret_cast<float*>(&ox);
        float* gee = reinterpret_cast<float*>(&oe);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float gin = wt.w[c] * (wt.s * (wt.u * (-(2.0f * r[k]))));
          gin = gin * nhmc_in1(pre[c][k]);
          const float gu = ((gin * c3) * nhmc_in1(uu[c][k])) / c2;
          gx[k] = gu;
          gee[k] = c4 * gin + (-gu) * c1;
        }
        nhmc_stnt(&g_xt[xbase + (int64_t)c * hw4 + q], ox);
        nhmc_stnt(&g_e[ebase + (int64_t)c * hw4 + q], oe);
      }
    }
  }
  __shared__ double red[4];
  double v[1] = {(double)acc};
  nhmc_block_sum<1>(v, red);
  if (threadIdx.x == 0) loss_ws[(int64_t)chain * gridDim.x + blockIdx.x] = v[0];
}

bool bad(int n_chains, int channels, int64_t hw) {
  return n_chains <= 0 || n_chains > 65535 || channels <= 0 || channels > MAXC || hw <= 0 || (hw & 3);
}

W weights(const float* w_host, int channels) {          // [v_0 .. v_{C-1}, s, u]
  W wt;
  for (int c = 0; c < MAXC; ++c) wt.w[c] = c < channels ? w_host[c] : 0.0f;
  wt.s = w_host[channels];
  wt.u = w_host[channels + 1];
  return wt;
}

template <int MODE>
int launch(const float* xin, const float* yin, float* out, const float* w_host, int channels, int apply_clip,
           double* ws, int n_chains, int64_t hw, hipStream_t st) {
  const W wt = weights(w_host, channels);
  const int64_t hw4 = hw / 4;
  dim3 grid((unsigned)((hw4 + NHMC_BLOCK - 1) / NHMC_BLOCK), (unsigned)n_chains);
  NHMC_LAUNCH((k_color<MODE>), grid, dim3(NHMC_BLOCK), 0, st, (const float4*)xin, (const float4*)yin, (float4*)out, wt,
              channels, apply_clip, ws, hw4);
  return nhmc_launch_status();
}

}  // namespace

extern "C" int nhmc_color_tiles(int64_t hw) { return (int)((hw / 4 + NHMC_BLOCK - 1) / NHMC_BLOCK); }

// w: HOST array [V[0,0] .. V[C-1,0], s, U[0,0]] of the row's SVD (channels + 2 floats; they travel as a kernel argument).
extern "C" int nhmc_data_color(const float* xt, const float* y, const float* w, int apply_clip, float* g_xt,
                               double* loss_ws, int n_chains, int channels, int64_t hw, nhmc_stream_t stream) {
  if (!xt || !y || !w || !g_xt || !loss_ws) return NHMC_ERR_ARG;
  if (bad(n_chains, channels, hw)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(xt) || !nhmc_aligned16(y) || !nhmc_aligned16(g_xt)) return NHMC_ERR_ALIGN;
  return launch<0>(xt, y, g_xt, w, channels, apply_clip, loss_ws, n_chains, hw, nhmc_s(stream));
}

extern "C" int nhmc_color_H(const float* x, const float* w, float* y, int n_chains, int channels, int64_t hw,
                            nhmc_stream_t stream) {
  if (!x || !w || !y) return NHMC_ERR_ARG;
  if (bad(n_chains, channels, hw)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(x) || !nhmc_aligned16(y)) return NHMC_ERR_ALIGN;
  return launch<1>(x, nullptr, y, w, channels, 0, nullptr, n_chains, hw, nhmc_s(stream));
}

// pinv = 0: H^T y;  1: H^+ y
extern "C" int nhmc_color_Ht(const float* y, const float* w, int pinv, float* x, int n_chains, int channels, int64_t hw,
                             nhmc_stream_t stream) {
  if (!x || !w || !y) return NHMC_ERR_ARG;
  if (bad(n_chains, channels, hw) || (pinv && w[channels] == 0.0f)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(x) || !nhmc_aligned16(y)) return NHMC_ERR_ALIGN;
  if (pinv) return launch<3>(nullptr, y, x, w, channels, 0, nullptr, n_chains, hw, nhmc_s(stream));
  return launch<2>(nullptr, y, x, w, channels, 0, nullptr, n_chains, hw, nhmc_s(stream));
}

extern "C" int nhmc_ddim_mix_bwd_color(const float* xt, const float* e, int e_channels, const float* at,
                                       const float* at_next, const float* y, const float* w, float* g_xt, float* g_e,
                                       double* loss_ws, int n_chains, int channels, int64_t hw, nhmc_stream_t stream) {
  if (!xt || !e || !at || !at_next || !y || !w || !g_xt || !g_e || !loss_ws) return NHMC_ERR_ARG;
  if (bad(n_chains, channels, hw) || (e_channels != channels && e_channels != 2 * channels)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(xt) || !nhmc_aligned16(e) || !nhmc_aligned16(y) || !nhmc_aligned16(g_xt) || !nhmc_aligned16(g_e))
    return NHMC_ERR_ALIGN;
  const W wt = weights(w, channels);
  const int64_t hw4 = hw / 4;
  dim3 grid((unsigned)((hw4 + NHMC_BLOCK - 1) / NHMC_BLOCK), (unsigned)n_chains);
  NHMC_LAUNCH(k_mix_bwd_color, grid, dim3(NHMC_BLOCK), 0, nhmc_s(stream), (const float4*)xt, (const float4*)e,
              e_channels, at, at_next, (const float4*)y, (float4*)g_xt, (float4*)g_e, wt, channels, loss_ws, hw4);
  return nhmc_launch_status();
}
