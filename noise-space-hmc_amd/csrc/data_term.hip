// Data term for the memory-op operators (rows a12-a14 of SURVEY.md section 8):
//   loss_b = sum (y_b - H clip(xt_b))^2,   g_xt = d loss / d xt = -2 H^T r (x) 1[-1 <= xt <= 1]
// Replaces main_sampling.py:693-695 / :709-711 with H = Inpainting (obs_functions/Hfuncs.py:119-154)
// or SuperResolution (:180-234), plus the bare H / H^T / H^+ the operator surface exposes (:65-90).
//
// Inpainting.  The reference permutes CHW->HWC, gathers the kept indices and lets autograd scatter
// the residual back through two index ops over the full image.  Here a dense int32 map `slot`
// (CHW order, shared by all chains, 4N bytes -> L2 resident) says for every image element which y
// entry observes it (-1: masked), so one coalesced pass writes the whole gradient, zeros included,
// with no memset and no scatter.
//
// Super-resolution.  H = r x r block mean (CHW), H^T = broadcast / r^2.  One thread owns a float4
// column strip of r rows; strips of a block are combined with wave shuffles (r/4 adjacent lanes),
// so every global access is a coalesced 16-B load/store and nothing is staged through LDS.
//
// Per-chain loss: fp32 per-thread partial -> fp64 wave shuffle -> fp64 per-tile partial in HBM ->
// nhmc_sum_partials (fixed order).  Deterministic; no float atomics.
#include "nhmc_common.h"

namespace {

__global__ __launch_bounds__(NHMC_BLOCK) void k_data_inpaint(
    const float4* __restrict__ xt, const float* __restrict__ y, const int4* __restrict__ slot, int apply_clip,
    float4* __restrict__ g_xt, double* __restrict__ loss_ws, int64_t n4, int64_t m) {
  const int chain = blockIdx.y;
  const int64_t base = (int64_t)chain * n4;
  const float* yb = y + (int64_t)chain * m;
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * NHMC_VEC_PER_THREAD) + threadIdx.x;
  double acc = 0.0;                       // fp32 squares summed in fp64 (independent of the tiling; matches the fused forms)
  int4 sv[NHMC_VEC_PER_THREAD];
  float4 xv[NHMC_VEC_PER_THREAD];
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q < n4) { sv[i] = slot[q]; xv[i] = nhmc_ldnt(&xt[base + q]); }
  }
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    const int* se = reinterpret_cast<const int*>(&sv[i]);
    const float* xe = reinterpret_cast<const float*>(&xv[i]);
    float4 o;
    float* oe = reinterpret_cast<float*>(&o);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float gr = 0.0f;
      if (se[c] >= 0) {
        const float v = apply_clip ? nhmc_clip1(xe[c]) : xe[c];
        const float r = yb[se[c]] - v;
        acc += (double)(r * r);
        gr = -(2.0f * r);
        if (apply_clip) gr = gr * nhmc_in1(xe[c]);
      }
      oe[c] = gr;
    }
    nhmc_stnt(&g_xt[base + q], o);
  }
  __shared__ double red[4];
  double v[1] = {acc};
  nhmc_block_sum<1>(v, red);
  if (threadIdx.x == 0) loss_ws[(int64_t)chain * gridDim.x + blockIdx.x] = v[0];
}

__global__ __launch_bounds__(NHMC_BLOCK) void k_inpaint_H(
    const float* __restrict__ x, const int32_t* __restrict__ kept_chw, float* __restrict__ y, int64_t n_elem,
    int64_t m) {
  const int chain = blockIdx.y;
  const int64_t k = (int64_t)blockIdx.x * NHMC_BLOCK + threadIdx.x;
  if (k < m) y[(int64_t)chain * m + k] = x[(int64_t)chain * n_elem + kept_chw[k]];
}

__global__ __launch_bounds__(NHMC_BLOCK) void k_inpaint_Ht(
    const float* __restrict__ y, const int4* __restrict__ slot, float4* __restrict__ x, int64_t n4, int64_t m) {
  const int chain = blockIdx.y;
  const float* yb = y + (int64_t)chain * m;
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * NHMC_VEC_PER_THREAD) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    const int4 s = slot[q];
    float4 o;
    o.x = s.x >= 0 ? yb[s.x] : 0.0f;
    o.y = s.y >= 0 ? yb[s.y] : 0.0f;
    o.z = s.z >= 0 ? yb[s.z] : 0.0f;
    o.w = s.w >= 0 ? yb[s.w] : 0.0f;
    nhmc_stnt(&x[(int64_t)chain * n4 + q], o);
  }
}

// ---- super-resolution ------------------------------------------------------------------------
// Work item = (block-row i, float4 column strip s) of one channel image: r rows x 4 columns.
// MODE 0: data term (loss + gradient), 1: H only, 2: H^T / H^+ (broadcast * scale).
template <int R, int MODE>
__global__ __launch_bounds__(NHMC_BLOCK) void k_sr(
    const float4* __restrict__ xt, const float* __restrict__ y_in, float* __restrict__ y_out, int apply_clip,
    float scale, float4* __restrict__ g_out, double* __restrict__ loss_ws, int dim, int planes_per_chain) {
  const int chain = blockIdx.y;
  const int w4 = dim / 4;                 // float4 strips per row
  const int yd = dim / R;                 // output rows / cols
  const int64_t items = (int64_t)planes_per_chain * yd * w4;
  const int64_t item = (int64_t)blockIdx.x * NHMC_BLOCK + threadIdx.x;
  const bool live = item < items;
  const int64_t it = live ? item : 0;
  const int s = (int)(it % w4);
  const int i = (int)((it / w4) % yd);
  const int plane = (int)(it / ((int64_t)w4 * yd));
  const int64_t img4 = ((int64_t)chain * planes_per_chain + plane) * (int64_t)dim * w4;   // float4 offset of the plane
  const int64_t row0 = img4 + (int64_t)i * R * w4 + s;
  const float* yplane_in = y_in ? y_in + ((int64_t)chain * planes_per_chain + plane) * (int64_t)yd * yd : nullptr;
  float* yplane_out = y_out ? y_out + ((int64_t)chain * planes_per_chain + plane) * (int64_t)yd * yd : nullptr;
  constexpr int LANES = R >= 4 ? R / 4 : 1;     // strips that share one block
  constexpr int BPS = R >= 4 ? 1 : 4 / R;       // blocks per strip (R = 2 -> 2)
  const float inv = 1.0f / (float)(R * R);
  float acc = 0.0f;

  if (MODE == 2) {                              // broadcast y * scale over the block
    if (live) {
      float4 o;
      float* oe = reinterpret_cast<float*>(&o);
#pragma unroll
      for (int c = 0; c < 4; ++c) oe[c] = yplane_in[(int64_t)i * yd + (s * 4 + c) / R] * scale;
#pragma unroll 4
      for (int rr = 0; rr < R; ++rr) nhmc_stnt(&g_out[row0 + (int64_t)rr * w4], o);
    }
    return;
  }

  float4 rows[R];
  float bs[BPS];
#pragma unroll
  for (int b = 0; b < BPS; ++b) bs[b] = 0.0f;
#pragma unroll
  for (int rr = 0; rr < R; ++rr) rows[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
#pragma unroll
    for (int rr = 0; rr < R; ++rr) rows[rr] = nhmc_ldnt(&xt[row0 + (int64_t)rr * w4]);
  }
  if constexpr (LANES > 1) {                    // R >= 8: the block spans LANES strips -- row-major running sum across them
    float4 vals[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      vals[rr] = rows[rr];
      if (MODE == 0 && apply_clip) {
        vals[rr].x = nhmc_clip1(vals[rr].x); vals[rr].y = nhmc_clip1(vals[rr].y);
        vals[rr].z = nhmc_clip1(vals[rr].z); vals[rr].w = nhmc_clip1(vals[rr].w);
      }
    }
    bs[0] = nhmc_block_sum_rowmajor<R, LANES>(vals, s % LANES);
  } else if (live) {
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const float* e = reinterpret_cast<const float*>(&rows[rr]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float v = (MODE == 0 && apply_clip) ? nhmc_clip1(e[c]) : e[c];
        bs[BPS == 1 ? 0 : c / R] += v;
      }
    }
  }
  if (!live) { /* fallthrough to the reduction with acc = 0 */ }
  float resid[BPS];
#pragma unroll
  for (int b = 0; b < BPS; ++b) {
    const int j = BPS == 1 ? (s * 4) / R : s * BPS + b;
    const float mean = bs[b] * inv;
    if (MODE == 1) {
      if (live && (LANES == 1 || (s % LANES) == 0)) yplane_out[(int64_t)i * yd + j] = mean;
      resid[b] = 0.0f;
    } else {
      resid[b] = live ? yplane_in[(int64_t)i * yd + j] - mean : 0.0f;
      if (live && (LANES == 1 || (s % LANES) == 0)) acc += resid[b] * resid[b];
    }
  }
  if (MODE == 0) {
    if (live) {
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        const float* e = reinterpret_cast<const float*>(&rows[rr]);
        float4 o;
        float* oe = reinterpret_cast<float*>(&o);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float gr = (-(2.0f * resid[BPS == 1 ? 0 : c / R])) * inv;
          if (apply_clip) gr = gr * nhmc_in1(e[c]);
          oe[c] = gr;
        }
        nhmc_stnt(&g_out[row0 + (int64_t)rr * w4], o);
      }
    }
    __shared__ double red[4];
    double v[1] = {(double)acc};
    nhmc_block_sum<1>(v, red);
    if (threadIdx.x == 0) loss_ws[(int64_t)chain * gridDim.x + blockIdx.x] = v[0];
  }
}

__global__ __launch_bounds__(NHMC_BLOCK) void k_sum_partials(const double* __restrict__ ws, int tiles, int stride,
                                                             int offset, int n_chains, double* __restrict__ out) {
  // one wave per chain: lanes stride over the tiles in a fixed pattern, then a shuffle tree
  const int chain = blockIdx.x * (NHMC_BLOCK / NHMC_WAVE) + (threadIdx.x >> 6);
  if (chain >= n_chains) return;
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (int t = lane; t < tiles; t += NHMC_WAVE) acc += ws[((int64_t)chain * tiles + t) * stride + offset];
  acc = nhmc_wave_sum(acc);
  if (lane == 0) out[chain] = acc;
}

int sr_tiles(int channels, int dim, int ratio) {
  const int64_t items = (int64_t)channels * (dim / ratio) * (dim / 4);
  return (int)((items + NHMC_BLOCK - 1) / NHMC_BLOCK);
}

template <int MODE>
int launch_sr(int ratio, const float* xin, const float* y_in, float* y_out, int apply_clip, float scale,
              float* g_out, double* ws, int n_chains, int channels, int dim, hipStream_t st) {
  dim3 grid((unsigned)sr_tiles(channels, dim, ratio), (unsigned)n_chains), block(NHMC_BLOCK);
#define NHMC_SR(R)                                                                                          \
  NHMC_LAUNCH((k_sr<R, MODE>), grid, block, 0, st, (const float4*)xin, y_in, y_out, apply_clip, scale, \
                     (float4*)g_out, ws, dim, channels)
  switch (ratio) {
    case 2: NHMC_SR(2); break;
    case 4: NHMC_SR(4); break;
    case 8: NHMC_SR(8); break;
    case 16: NHMC_SR(16); break;
    case 32: NHMC_SR(32); break;
    default: return NHMC_ERR_SHAPE;
  }
#undef NHMC_SR
  return nhmc_launch_status();
}

bool sr_bad(int ratio, int n_chains, int channels, int dim) {
  return n_chains <= 0 || n_chains > 65535 || channels <= 0 || dim <= 0 || (dim % 4) || (dim % ratio) ||
         !(ratio == 2 || ratio == 4 || ratio == 8 || ratio == 16 || ratio == 32);
  // the R/4 strips of one block are R/4 consecutive, R/4-aligned work items (dim/4 is a multiple of R/4),
  // hence always lanes of one wave
}

}  // namespace

extern "C" int nhmc_data_tiles(int64_t n_elem) { return (int)((n_elem + NHMC_TILE - 1) / NHMC_TILE); }

extern "C" size_t nhmc_data_ws_bytes(int n_chains, int64_t n_elem) {
  // inpaint uses n_elem/4096 tiles; sr uses at most n_elem/(4*2*256) -> size for the larger
  const int64_t sr_max = (n_elem / 8 + NHMC_BLOCK - 1) / NHMC_BLOCK + 1;
  const int64_t t = nhmc_data_tiles(n_elem) > sr_max ? nhmc_data_tiles(n_elem) : sr_max;
  return (size_t)n_chains * (size_t)t * sizeof(double);
}

extern "C" int nhmc_data_inpaint(const float* xt, const float* y, const int32_t* slot, int apply_clip, float* g_xt,
                                 double* loss_ws, int n_chains, int64_t n_elem, int64_t m, nhmc_stream_t stream) {
  if (!xt || !y || !slot || !g_xt || !loss_ws || n_chains <= 0 || n_elem <= 0 || m <= 0) return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  if ((n_elem & 3) || !nhmc_aligned16(xt) || !nhmc_aligned16(slot) || !nhmc_aligned16(g_xt)) return NHMC_ERR_ALIGN;
  dim3 grid((unsigned)nhmc_data_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  NHMC_LAUNCH(k_data_inpaint, grid, block, 0, nhmc_s(stream), (const float4*)xt, y, (const int4*)slot,
                     apply_clip, (float4*)g_xt, loss_ws, n_elem / 4, m);
  return nhmc_launch_status();
}

extern "C" int nhmc_inpaint_H(const float* x, const int32_t* kept_chw, float* y, int n_chains, int64_t n_elem,
                              int64_t m, nhmc_stream_t stream) {
  if (!x || !kept_chw || !y || n_chains <= 0 || n_elem <= 0 || m <= 0) return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  dim3 grid((unsigned)((m + NHMC_BLOCK - 1) / NHMC_BLOCK), (unsigned)n_chains), block(NHMC_BLOCK);
  NHMC_LAUNCH(k_inpaint_H, grid, block, 0, nhmc_s(stream), x, kept_chw, y, n_elem, m);
  return nhmc_launch_status();
}

extern "C" int nhmc_inpaint_Ht(const float* y, const int32_t* slot, float* x, int n_chains, int64_t n_elem,
                               int64_t m, nhmc_stream_t stream) {
  if (!y || !slot || !x || n_chains <= 0 || n_elem <= 0 || m <= 0) return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  if ((n_elem & 3) || !nhmc_aligned16(slot) || !nhmc_aligned16(x)) return NHMC_ERR_ALIGN;
  dim3 grid((unsigned)nhmc_data_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  NHMC_LAUNCH(k_inpaint_Ht, grid, block, 0, nhmc_s(stream), y, (const int4*)slot, (float4*)x, n_elem / 4, m);
  return nhmc_launch_status();
}

extern "C" int nhmc_data_sr(const float* xt, const float* y, int ratio, int apply_clip, float* g_xt,
                            double* loss_ws, int n_chains, int channels, int dim, nhmc_stream_t stream) {
  if (!xt || !y || !g_xt || !loss_ws) return NHMC_ERR_ARG;
  if (sr_bad(ratio, n_chains, channels, dim)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(xt) || !nhmc_aligned16(g_xt)) return NHMC_ERR_ALIGN;
  return launch_sr<0>(ratio, xt, y, nullptr, apply_clip, 0.0f, g_xt, loss_ws, n_chains, channels, dim, nhmc_s(stream));
}

extern "C" int nhmc_sr_tiles(int channels, int dim, int ratio) { return sr_tiles(channels, dim, ratio); }

extern "C" int nhmc_sr_H(const float* x, float* y, int ratio, int n_chains, int channels, int dim,
                         nhmc_stream_t stream) {
  if (!x || !y) return NHMC_ERR_ARG;
  if (sr_bad(ratio, n_chains, channels, dim)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(x)) return NHMC_ERR_ALIGN;
  return launch_sr<1>(ratio, x, nullptr, y, 0, 0.0f, nullptr, nullptr, n_chains, channels, dim, nhmc_s(stream));
}

extern "C" int nhmc_sr_Ht(const float* y, float* x, int ratio, float scale, int n_chains, int channels, int dim,
                          nhmc_stream_t stream) {
  if (!x || !y) return NHMC_ERR_ARG;
  if (sr_bad(ratio, n_chains, channels, dim)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(x)) return NHMC_ERR_ALIGN;
  return launch_sr<2>(ratio, nullptr, y, nullptr, 0, scale, x, nullptr, n_chains, channels, dim, nhmc_s(stream));
}

extern "C" int nhmc_sum_partials(const double* ws, int tiles, int stride, int offset, int n_chains, double* out,
                                 nhmc_stream_t stream) {
  if (!ws || !out || tiles <= 0 || stride <= 0 || offset < 0 || offset >= stride || n_chains <= 0) return NHMC_ERR_ARG;
  const int per = NHMC_BLOCK / NHMC_WAVE;
  NHMC_LAUNCH(k_sum_partials, dim3((n_chains + per - 1) / per), dim3(NHMC_BLOCK), 0, nhmc_s(stream), ws,
                     tiles, stride, offset, n_chains, out);
  return nhmc_launch_status();
}
