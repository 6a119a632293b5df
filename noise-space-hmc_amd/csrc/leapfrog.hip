// Fused leapfrog momentum + position update (rows a1-a4 of SURVEY.md section 8).
// Replaces main_sampling.py:702 (first half step), :706-707 (position), :713 (momentum),
// :715 (last half-step undo) and produces the per-chain sums the Hamiltonians at :697/:717 need.
//
// HBM-bound streaming kernel: MID reads x,p,g and writes x,p = 20 B/element (5T per chain).
// grid = (tiles, chains); 256 threads; every thread owns VPT float4 per stream (non-temporal; 2 in FIRST / LAST,
// 1 in MID), strided by the block so each wave-instruction touches 1 KiB contiguous.  Per-chain scalars are fp64 device
// values rounded once to fp32, exactly as `python_float * tensor` does in the reference.
// Compiled with -ffp-contract=off: mul/add stay separate, matching the reference's ATen op order.
#include "nhmc_common.h"

namespace {

// OOP (FIRST only): the position is read from xin and the proposal written to x, so the caller's accepted position
// survives the trajectory without a copy (same traffic as in place: R xin, W x).
//
// CACHED: the gradient cache of the sampler (two slots per chain; sel[chain] = slot that holds loss / gradient at the
// chain's ACCEPTED position).  FIRST reads its gradient from slot sel (the previous trajectory already evaluated that
// point: its LAST step if it was accepted, its FIRST step if not -- main_sampling.py:693-695 re-evaluates it); LAST
// stores the summed gradient g (+ g2) and the loss of the end point into slot 1 - sel, where they wait for the accept
// decision (nhmc_grad_cache_flip).  pair4 = distance between the two slots in float4 units.
struct LfCache {
  const int32_t* sel;       // [n_chains]
  float4* g_pair;           // slot 0 base of this launch's first chain
  int64_t pair4;
  const double* loss_in;    // LAST: loss of the end point [n_chains]
  double* loss_pair;        // LAST: slot 0 base of the loss cache
  int64_t loss_stride;
};

template <int MODE, bool HAS_G2, int VPT, bool OOP = false, bool CACHED = false>
__global__ __launch_bounds__(NHMC_BLOCK) void k_leapfrog(
    const float4* __restrict__ xin, float4* __restrict__ x, float4* __restrict__ p, const float4* __restrict__ g,
    const float4* __restrict__ g2, const double* __restrict__ eps, const double* __restrict__ sigma_y,
    double m_inv, int64_t n4, double* __restrict__ sums_ws, LfCache cache = LfCache()) {
  const int chain = blockIdx.y;
  float4* gsave = nullptr;
  if (CACHED) {
    const int s = cache.sel[chain] & 1;
    if (MODE == NHMC_LF_FIRST) g = cache.g_pair + (int64_t)s * cache.pair4;
    if (MODE == NHMC_LF_LAST) {
      gsave = cache.g_pair + (int64_t)(1 - s) * cache.pair4;
      if (blockIdx.x == 0 && threadIdx.x == 0)
        cache.loss_pair[(int64_t)(1 - s) * cache.loss_stride + chain] = cache.loss_in[chain];
    }
  }
  const double e = eps[chain], s = sigma_y[chain];
  const float kf = (float)(1.0 / (2.0 * (s * s)));   // 1/(2*sigma_y**2)
  const float ef = (float)e;                          // epsilon
  const float eh = (float)(e / 2.0);                  // epsilon / 2
  const float ex = (float)(e * m_inv);                // epsilon * m**(-1)

  const int64_t base = (int64_t)chain * n4;
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * VPT) + threadIdx.x;

  float4 xv[VPT], pv[VPT], gv[VPT];
  bool ok[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    ok[i] = q < n4;
    if (ok[i]) {
      xv[i] = nhmc_ldnt(OOP ? &xin[base + q] : &x[base + q]);
      pv[i] = nhmc_ldnt(&p[base + q]);
      gv[i] = nhmc_ldnt(&g[base + q]);
      if (HAS_G2) {
        const float4 h = nhmc_ldnt(&g2[base + q]);
        gv[i].x += h.x; gv[i].y += h.y; gv[i].z += h.z; gv[i].w += h.w;
      }
    }
  }

  float sx = 0.0f, sp = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    if (!ok[i]) continue;
    float* xe = reinterpret_cast<float*>(&xv[i]);
    float* pe = reinterpret_cast<float*>(&pv[i]);
    const float* ge = reinterpret_cast<const float*>(&gv[i]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (MODE == NHMC_LF_FIRST) { sx += xe[c] * xe[c]; sp += pe[c] * pe[c]; }
      const float G = xe[c] + kf * ge[c];
      if (MODE == NHMC_LF_FIRST) {
        pe[c] = pe[c] - eh * G;
        xe[c] = xe[c] + ex * pe[c];
      } else if (MODE == NHMC_LF_MID) {
        pe[c] = pe[c] - ef * G;
        xe[c] = xe[c] + ex * pe[c];
      } else {
        pe[c] = pe[c] - ef * G;
        pe[c] = pe[c] + eh * G;
        sx += xe[c] * xe[c];
        sp += pe[c] * pe[c];
      }
    }
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    nhmc_stnt(&p[base + q], pv[i]);
    if (MODE != NHMC_LF_LAST) nhmc_stnt(&x[base + q], xv[i]);
    if (CACHED && MODE == NHMC_LF_LAST) nhmc_stnt(&gsave[base + q], gv[i]);
  }

  if (MODE != NHMC_LF_MID) {
    __shared__ double red[8];
    double v[2] = {(double)sx, (double)sp};
    nhmc_block_sum<2>(v, red);
    if (threadIdx.x == 0) {
      double* dst = sums_ws + ((int64_t)chain * gridDim.x + blockIdx.x) * 2;
      dst[0] = v[0];
      dst[1] = v[1];
    }
  }
}

__global__ __launch_bounds__(NHMC_BLOCK) void k_copy_probe(const float4* __restrict__ src, float4* __restrict__ dst,
                                                           int64_t n4) {
  const int64_t q = (int64_t)blockIdx.x * NHMC_BLOCK + threadIdx.x;
  if (q < n4) nhmc_stnt(&dst[q], nhmc_ldnt(&src[q]));
}

template <int MODE>
int launch(float* x, float* p, const float* g, const float* g2, const double* eps, const double* sigma_y,
           double m_inv, int n_chains, int64_t n_elem, double* ws, hipStream_t st) {
  const int64_t n4 = n_elem / 4;
  // FIRST / LAST write one partial per nhmc_leapfrog_tiles() tile (2 float4 per thread); MID has no reduction and
  // runs 1 float4 per thread (measured: 41.2 us vs 41.8 us at B = 64, tools/lf_bench.hip)
  constexpr int VPT = MODE == NHMC_LF_MID ? 1 : NHMC_VEC_PER_THREAD;
  const unsigned tiles = MODE == NHMC_LF_MID ? (unsigned)((n4 + NHMC_BLOCK - 1) / NHMC_BLOCK) : (unsigned)nhmc_leapfrog_tiles(n_elem);
  dim3 grid(tiles, (unsigned)n_chains), block(NHMC_BLOCK);
  if (g2)
    NHMC_LAUNCH((k_leapfrog<MODE, true, VPT>), grid, block, 0, st, (const float4*)nullptr, (float4*)x, (float4*)p,
                (const float4*)g, (const float4*)g2, eps, sigma_y, m_inv, n4, ws, LfCache());
  else
    NHMC_LAUNCH((k_leapfrog<MODE, false, VPT>), grid, block, 0, st, (const float4*)nullptr, (float4*)x, (float4*)p,
                (const float4*)g, (const float4*)nullptr, eps, sigma_y, m_inv, n4, ws, LfCache());
  return nhmc_launch_status();
}

}  // namespace

extern "C" int nhmc_leapfrog_tiles(int64_t n_elem) { return (int)((n_elem + NHMC_TILE - 1) / NHMC_TILE); }

extern "C" size_t nhmc_leapfrog_ws_bytes(int n_chains, int64_t n_elem) {
  return (size_t)n_chains * (size_t)nhmc_leapfrog_tiles(n_elem) * 2 * sizeof(double);
}

extern "C" int nhmc_leapfrog_fused(int mode, float* x, float* p, const float* g, const float* g2,
                                   const double* eps, const double* sigma_y, double m_inv, int n_chains,
                                   int64_t n_elem, double* sums_ws, nhmc_stream_t stream) {
  if (!x || !p || !g || !eps || !sigma_y || n_chains <= 0 || n_elem <= 0) return NHMC_ERR_ARG;
  if (mode != NHMC_LF_MID && !sums_ws) return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  if ((n_elem & 3) || !nhmc_aligned16(x) || !nhmc_aligned16(p) || !nhmc_aligned16(g) || (g2 && !nhmc_aligned16(g2)))
    return NHMC_ERR_ALIGN;
  hipStream_t st = nhmc_s(stream);
  switch (mode) {
    case NHMC_LF_FIRST: return launch<NHMC_LF_FIRST>(x, p, g, g2, eps, sigma_y, m_inv, n_chains, n_elem, sums_ws, st);
    case NHMC_LF_MID:   return launch<NHMC_LF_MID>(x, p, g, g2, eps, sigma_y, m_inv, n_chains, n_elem, sums_ws, st);
    case NHMC_LF_LAST:  return launch<NHMC_LF_LAST>(x, p, g, g2, eps, sigma_y, m_inv, n_chains, n_elem, sums_ws, st);
    default: return NHMC_ERR_ARG;
  }
}

extern "C" int nhmc_leapfrog_first(const float* x_in, float* x_out, float* p, const float* g, const float* g2,
                                   const double* eps, const double* sigma_y, double m_inv, int n_chains,
                                   int64_t n_elem, double* sums_ws, nhmc_stream_t stream) {
  if (!x_in || !x_out || x_in == x_out || !p || !g || !eps || !sigma_y || !sums_ws || n_chains <= 0 || n_elem <= 0)
    return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  if ((n_elem & 3) || !nhmc_aligned16(x_in) || !nhmc_aligned16(x_out) || !nhmc_aligned16(p) || !nhmc_aligned16(g) ||
      (g2 && !nhmc_aligned16(g2)))
    return NHMC_ERR_ALIGN;
  const int64_t n4 = n_elem / 4;
  dim3 grid((unsigned)nhmc_leapfrog_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  hipStream_t st = nhmc_s(stream);
  if (g2)
    NHMC_LAUNCH((k_leapfrog<NHMC_LF_FIRST, true, NHMC_VEC_PER_THREAD, true>), grid, block, 0, st, (const float4*)x_in,
                (float4*)x_out, (float4*)p, (const float4*)g, (const float4*)g2, eps, sigma_y, m_inv, n4, sums_ws, LfCache());
  else
    NHMC_LAUNCH((k_leapfrog<NHMC_LF_FIRST, false, NHMC_VEC_PER_THREAD, true>), grid, block, 0, st, (const float4*)x_in,
                (float4*)x_out, (float4*)p, (const float4*)g, (const float4*)nullptr, eps, sigma_y, m_inv, n4, sums_ws,
                LfCache());
  return nhmc_launch_status();
}

// ---- gradient cache (one decode per leapfrog step: the FIRST half step reuses the previous evaluation) ----------
namespace {

__global__ __launch_bounds__(NHMC_BLOCK) void k_grad_cache_store(const float4* __restrict__ g,
                                                                 const float4* __restrict__ g2, LfCache cache, int flip,
                                                                 int64_t n4) {
  const int chain = blockIdx.y;
  const int s = (cache.sel[chain] ^ flip) & 1;
  if (blockIdx.x == 0 && threadIdx.x == 0) cache.loss_pair[(int64_t)s * cache.loss_stride + chain] = cache.loss_in[chain];
  float4* dst = cache.g_pair + (int64_t)s * cache.pair4 + (int64_t)chain * n4;
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * NHMC_VEC_PER_THREAD) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    float4 v = nhmc_ldnt(&g[(int64_t)chain * n4 + q]);
    if (g2) {
      const float4 h = nhmc_ldnt(&g2[(int64_t)chain * n4 + q]);
      v.x += h.x; v.y += h.y; v.z += h.z; v.w += h.w;
    }
    nhmc_stnt(&dst[q], v);
  }
}

__global__ void k_grad_cache_flip(const int32_t* __restrict__ accept, int32_t* __restrict__ sel, int n_chains) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n_chains && accept[c]) sel[c] = (sel[c] ^ 1) & 1;
}

bool cache_args_ok(const float* g_pair, const int32_t* sel, int64_t pair_stride, int n_chains, int64_t n_elem) {
  return g_pair && sel && pair_stride >= (int64_t)n_chains * n_elem;
}

}  // namespace

extern "C" int nhmc_leapfrog_first_cached(const float* x_in, float* x_out, float* p, const float* g_pair,
                                          const int32_t* sel, int64_t pair_stride, const double* eps,
                                          const double* sigma_y, double m_inv, int n_chains, int64_t n_elem,
                                          double* sums_ws, nhmc_stream_t stream) {
  if (!x_in || !x_out || x_in == x_out || !p || !eps || !sigma_y || !sums_ws || n_chains <= 0 || n_elem <= 0 ||
      !cache_args_ok(g_pair, sel, pair_stride, n_chains, n_elem))
    return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  if ((n_elem & 3) || (pair_stride & 3) || !nhmc_aligned16(x_in) || !nhmc_aligned16(x_out) || !nhmc_aligned16(p) ||
      !nhmc_aligned16(g_pair))
    return NHMC_ERR_ALIGN;
  LfCache c{sel, (float4*)g_pair, pair_stride / 4, nullptr, nullptr, 0};
  dim3 grid((unsigned)nhmc_leapfrog_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  NHMC_LAUNCH((k_leapfrog<NHMC_LF_FIRST, false, NHMC_VEC_PER_THREAD, true, true>), grid, block, 0, nhmc_s(stream),
              (const float4*)x_in, (float4*)x_out, (float4*)p, (const float4*)nullptr, (const float4*)nullptr, eps,
              sigma_y, m_inv, n_elem / 4, sums_ws, c);
  return nhmc_launch_status();
}

extern "C" int nhmc_leapfrog_last_cached(float* x, float* p, const float* g, const float* g2, float* g_pair,
                                         const int32_t* sel, int64_t pair_stride, const double* loss,
                                         double* loss_pair, int64_t loss_stride, const double* eps,
                                         const double* sigma_y, double m_inv, int n_chains, int64_t n_elem,
                                         double* sums_ws, nhmc_stream_t stream) {
  if (!x || !p || !g || !eps || !sigma_y || !sums_ws || !loss || !loss_pair || loss_stride < n_chains ||
      n_chains <= 0 || n_elem <= 0 || !cache_args_ok(g_pair, sel, pair_stride, n_chains, n_elem))
    return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  if ((n_elem & 3) || (pair_stride & 3) || !nhmc_aligned16(x) || !nhmc_aligned16(p) || !nhmc_aligned16(g) ||
      (g2 && !nhmc_aligned16(g2)) || !nhmc_aligned16(g_pair))
    return NHMC_ERR_ALIGN;
  LfCache c{sel, (float4*)g_pair, pair_stride / 4, loss, loss_pair, loss_stride};
  dim3 grid((unsigned)nhmc_leapfrog_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  if (g2)
    NHMC_LAUNCH((k_leapfrog<NHMC_LF_LAST, true, NHMC_VEC_PER_THREAD, false, true>), grid, block, 0, nhmc_s(stream),
                (const float4*)nullptr, (float4*)x, (float4*)p, (const float4*)g, (const float4*)g2, eps, sigma_y, m_inv,
                n_elem / 4, sums_ws, c);
  else
    NHMC_LAUNCH((k_leapfrog<NHMC_LF_LAST, false, NHMC_VEC_PER_THREAD, false, true>), grid, block, 0, nhmc_s(stream),
                (const float4*)nullptr, (float4*)x, (float4*)p, (const float4*)g, (const float4*)nullptr, eps, sigma_y,
                m_inv, n_elem / 4, sums_ws, c);
  return nhmc_launch_status();
}

extern "C" int nhmc_grad_cache_store(const float* g, const float* g2, const double* loss, float* g_pair,
                                     double* loss_pair, const int32_t* sel, int flip, int64_t pair_stride,
                                     int64_t loss_stride, int n_chains, int64_t n_elem, nhmc_stream_t stream) {
  if (!g || !loss || !loss_pair || loss_stride < n_chains || n_chains <= 0 || n_elem <= 0 || (flip & ~1) ||
      !cache_args_ok(g_pair, sel, pair_stride, n_chains, n_elem))
    return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  if ((n_elem & 3) || (pair_stride & 3) || !nhmc_aligned16(g) || (g2 && !nhmc_aligned16(g2)) || !nhmc_aligned16(g_pair))
    return NHMC_ERR_ALIGN;
  LfCache c{sel, (float4*)g_pair, pair_stride / 4, loss, loss_pair, loss_stride};
  dim3 grid((unsigned)nhmc_leapfrog_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  NHMC_LAUNCH(k_grad_cache_store, grid, block, 0, nhmc_s(stream), (const float4*)g, (const float4*)g2, c, flip,
              n_elem / 4);
  return nhmc_launch_status();
}

extern "C" int nhmc_grad_cache_flip(const int32_t* accept, int32_t* sel, int n_chains, nhmc_stream_t stream) {
  if (!accept || !sel || n_chains <= 0) return NHMC_ERR_ARG;
  NHMC_LAUNCH(k_grad_cache_flip, dim3((unsigned)((n_chains + 255) / 256)), dim3(256), 0, nhmc_s(stream), accept, sel,
              n_chains);
  return nhmc_launch_status();
}

extern "C" int nhmc_copy_probe(const float* src, float* dst, int64_t n_elem, nhmc_stream_t stream) {
  if (!src || !dst || n_elem <= 0) return NHMC_ERR_ARG;
  if ((n_elem & 3) || !nhmc_aligned16(src) || !nhmc_aligned16(dst)) return NHMC_ERR_ALIGN;
  const int64_t n4 = n_elem / 4, blocks = (n4 + NHMC_BLOCK - 1) / NHMC_BLOCK;
  if (blocks > 0x7fffffff) return NHMC_ERR_SHAPE;
  NHMC_LAUNCH(k_copy_probe, dim3((unsigned)blocks), dim3(NHMC_BLOCK), 0, nhmc_s(stream), (const float4*)src,
              (float4*)dst, n4);
  return nhmc_launch_status();
}
