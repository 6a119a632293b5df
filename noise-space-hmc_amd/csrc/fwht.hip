// Walsh-Hadamard compressive-sensing operator (row f.3 of SURVEY.md section 8).
// Replaces obs_functions/Hfuncs.py:611-651 (WalshHadamardCS): with F = FWHT / d (orthonormal, self-inverse) on each
// d x d channel image flattened row-major,   y[k*C + c] = F(x_c)[perm[k]]  for k < d^2/ratio,   H^T = H^+ = F(scatter).
//
// F over d^2 = 2^(2 log2 d) points factors as F_d (x) F_d: the first log2 d butterfly stages act inside rows
// (contiguous), the last log2 d across rows -- the same stage order and pairing as the reference's loop (:613-621),
// so the transform is bit-exact against it.
//   pass A (k_fwht_rows):  a lane owns V = max(1, d/64) consecutive elements of a row; stages h < V in registers,
//                          stages h >= V with __shfl_xor inside the row's lane group.  No LDS.  Optional prologue:
//                          clip (data term) or gather from y through the slot map (H^T).
//   pass B (k_fwht_cols):  a block stages a d x PC column panel in LDS (PC*4 = 128-byte row segments), runs the
//                          log2 d row-index stages there (consecutive threads -> consecutive columns: conflict-free),
//                          scales by 1/d and applies the epilogue: plain store, scatter to y (H), residual against
//                          y + loss partial (data term, writes the zero-filled spectrum for the adjoint), or
//                          -2 * clip-mask (data-term gradient).
// The data term's adjoint runs the stages in DESCENDING order -- columns first (h = d^2/2 .. d), then rows
// (h = d/2 .. 1) -- because that is what autograd does with the reference's loop: the backward of stage h is the same
// butterfly, visited last-to-first, after the backward of the final "/ img_dim" (a power of two: exact wherever it is
// applied).  Mathematically the same transform, but sums of different pairs: with the ascending order the gradient sat
// 1e-7 from the reference's and the G15 replay left the reference's run after 183 trajectories.
// HBM-bound: each pass reads and writes the image once (2T); a data term is 4 passes.
#include "nhmc_common.h"

namespace {

enum { PRO_NONE = 0, PRO_CLIP = 1, PRO_GATHER = 2 };
enum { EPI_STORE = 0, EPI_SCATTER = 1, EPI_RESID = 2, EPI_GRAD = 3, EPI_VJP = 4, EPI_RAW = 5 };
// EPI_RAW: plain store without the 1/d scale (the descending column pass; the row pass that follows scales)
// EPI_VJP: the gradient goes straight through the VJP of the last DDIM step (k_mix_bwd with final_clip = 1); `xt` is then
// the step's input and e / g_e are [chain][e_channels][d][d]
struct VjpArgs { const float* e; float* g_e; const float* at; const float* at_next; int e_channels; };
struct VjpCoef { float c1, c2, c3, c4; };
__device__ __forceinline__ VjpCoef vjp_coef(const VjpArgs& vj, int64_t chain) {
  const float a = vj.at[chain], an = vj.at_next[chain];
  VjpCoef k;
  k.c1 = sqrtf(1.0f - a); k.c2 = sqrtf(a); k.c3 = sqrtf(an); k.c4 = sqrtf(1.0f - an);
  return k;
}
// one element: data-term value v (already scaled) -> (g_xt, g_e), same op order as k_mix_bwd<false,false>
__device__ __forceinline__ void vjp_elem(const VjpCoef& k, float v, float x, float ee, float& gx, float& ge) {
  const float u = (x - ee * k.c1) / k.c2;
  float gin = -(2.0f * v);
  gin = gin * nhmc_in1(k.c3 * nhmc_clip1(u) + k.c4 * ee);
  const float gu = ((gin * k.c3) * nhmc_in1(u)) / k.c2;
  gx = gu;
  ge = k.c4 * gin + (-gu) * k.c1;
}

template <int V> struct Vec;
template <> struct Vec<1> { typedef float type; };
template <> struct Vec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec<4> { typedef float type __attribute__((ext_vector_type(4))); };

// ---- pass A: rows -------------------------------------------------------------------------------------
// DESC: stages h = d/2 .. 1 (the second half of the adjoint), then * 1/d and the gradient epilogue EPI (EPI_GRAD: -2 x
// clip mask of xt; EPI_VJP: through the last DDIM step's VJP); ascending passes store plainly (EPI_STORE).
template <int V, int PRO, bool DESC = false, int EPI = EPI_STORE>
__global__ __launch_bounds__(NHMC_BLOCK) void k_fwht_rows(const float* __restrict__ in, const float* __restrict__ y,
                                                          const int32_t* __restrict__ kslot, float* __restrict__ out,
                                                          int d, int channels, int64_t m, int64_t total_rows,
                                                          const float* __restrict__ xt, int apply_clip, VjpArgs vj) {
  const int lpr = d / V;                               // lanes per row (<= 64)
  const int rows_per_wave = NHMC_WAVE / lpr;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (NHMC_BLOCK / NHMC_WAVE) + (threadIdx.x >> 6);
  const int64_t row = wave * rows_per_wave + lane / lpr;            // global row over [chain][channel][i]
  const int j0 = (lane % lpr) * V;
  const bool live = row < total_rows;
  float v[V];
  if (live) {
    const int64_t base = row * d + j0;
    if (PRO == PRO_GATHER) {
      const int64_t plane = row / d;                                // chain*channels + c
      const int c = (int)(plane % channels);
      const int64_t chain = plane / channels;
      const int64_t q = (row % d) * d + j0;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const int k = kslot[q + e];
        v[e] = k >= 0 ? y[chain * m + (int64_t)k * channels + c] : 0.0f;
      }
    } else {
      typename Vec<V>::type w = *reinterpret_cast<const typename Vec<V>::type*>(&in[base]);   // V*4-byte aligned: d % V == 0
      const float* we = reinterpret_cast<const float*>(&w);
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] = PRO == PRO_CLIP ? nhmc_clip1(we[e]) : we[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = 0.0f;
  }
  if (!DESC) {
    // in-register stages h = 1 .. V/2
#pragma unroll
    for (int h = 1; h < V; h <<= 1) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        if ((e & h) == 0) {
          const float a = v[e], b = v[e + h];
          v[e] = a + b;
          v[e + h] = a - b;
        }
      }
    }
    // cross-lane stages h = V .. d/2  <->  lane xor (h / V)
    for (int s = 1; s < lpr; s <<= 1) {
      const bool upper = (lane & s) != 0;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float other = __shfl_xor(v[e], s, NHMC_WAVE);
        v[e] = upper ? other - v[e] : v[e] + other;
      }
    }
  } else {
    for (int s = lpr >> 1; s >= 1; s >>= 1) {                       // h = d/2 .. V
      const bool upper = (lane & s) != 0;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float other = __shfl_xor(v[e], s, NHMC_WAVE);
        v[e] = upper ? other - v[e] : v[e] + other;
      }
    }
#pragma unroll
    for (int h = V >> 1; h >= 1; h >>= 1) {                         // h = V/2 .. 1
#pragma unroll
      for (int e = 0; e < V; ++e) {
        if ((e & h) == 0) {
          const float a = v[e], b = v[e + h];
          v[e] = a + b;
          v[e + h] = a - b;
        }
      }
    }
  }
  if (live) {
    const int64_t base = row * d + j0;
    typename Vec<V>::type w;
    float* we = reinterpret_cast<float*>(&w);
    if (EPI == EPI_STORE) {
#pragma unroll
      for (int e = 0; e < V; ++e) we[e] = v[e];
    } else {
      const float scale = 1.0f / (float)d;
      const typename Vec<V>::type xw = *reinterpret_cast<const typename Vec<V>::type*>(&xt[base]);
      const float* xe = reinterpret_cast<const float*>(&xw);
      if (EPI == EPI_GRAD) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          float g = -(2.0f * (v[e] * scale));
          if (apply_clip) g = g * nhmc_in1(xe[e]);
          we[e] = g;
        }
      } else {                                                      // EPI_VJP
        const int64_t plane = row / d;
        const int c = (int)(plane % channels);
        const int64_t chain = plane / channels;
        const VjpCoef kc = vjp_coef(vj, chain);
        const int64_t eoff = (chain * vj.e_channels + c) * (int64_t)d * d + (row % d) * d + j0;
        const typename Vec<V>::type ew = *reinterpret_cast<const typename Vec<V>::type*>(&vj.e[eoff]);
        const float* ee = reinterpret_cast<const float*>(&ew);
        typename Vec<V>::type gw;
        float* ge = reinterpret_cast<float*>(&gw);
#pragma unroll
        for (int e = 0; e < V; ++e) vjp_elem(kc, v[e] * scale, xe[e], ee[e], we[e], ge[e]);
        *reinterpret_cast<typename Vec<V>::type*>(&vj.g_e[eoff]) = gw;
      }
    }
    *reinterpret_cast<typename Vec<V>::type*>(&out[base]) = w;
  }
}

// ---- pass B: columns ----------------------------------------------------------------------------------
template <int EPI, bool DESC = false>
__global__ __launch_bounds__(NHMC_BLOCK) void k_fwht_cols(const float* __restrict__ in, float* __restrict__ out,
                                                          const float* __restrict__ y, float* __restrict__ y_out,
                                                          const int32_t* __restrict__ kslot, const float* __restrict__ xt,
                                                          double* __restrict__ loss_ws, int d, int pc, int channels,
                                                          int64_t m, int apply_clip, VjpArgs vj) {
  extern __shared__ float tile[];                                   // [d][pc]
  const int64_t plane = blockIdx.y;                                 // chain*channels + c
  const int c0 = blockIdx.x * pc;
  const int c = (int)(plane % channels);
  const int64_t chain = plane / channels;
  const float* __restrict__ src = in + plane * (int64_t)d * d;
  const int n = d * pc;
  for (int idx = threadIdx.x; idx < n; idx += NHMC_BLOCK) {
    const int i = idx / pc, cc = idx % pc;
    tile[idx] = src[(int64_t)i * d + c0 + cc];
  }
  __syncthreads();
  for (int st = 1; st < d; st <<= 1) {
    const int h = DESC ? (d >> 1) / st : st;                        // ascending 1 .. d/2, or descending d/2 .. 1
    for (int idx = threadIdx.x; idx < n / 2; idx += NHMC_BLOCK) {
      const int cc = idx % pc, pr = idx / pc;                       // pair number -> lower row index
      const int i = (pr / h) * 2 * h + (pr % h);
      const float a = tile[i * pc + cc], b = tile[(i + h) * pc + cc];
      tile[i * pc + cc] = a + b;
      tile[(i + h) * pc + cc] = a - b;
    }
    __syncthreads();
  }
  const float scale = 1.0f / (float)d;                              // fwht(...) / img_dim  (:622)
  float acc = 0.0f;
  for (int idx = threadIdx.x; idx < n; idx += NHMC_BLOCK) {
    const int i = idx / pc, cc = idx % pc;
    const int64_t q = (int64_t)i * d + c0 + cc;                     // position inside the plane
    const int64_t off = plane * (int64_t)d * d + q;
    float v = tile[idx] * scale;
    if (EPI == EPI_RAW) {
      out[off] = tile[idx];
    } else if (EPI == EPI_STORE) {
      out[off] = v;
    } else if (EPI == EPI_SCATTER) {
      const int k = kslot[q];
      if (k >= 0) y_out[chain * m + (int64_t)k * channels + c] = v;
    } else if (EPI == EPI_RESID) {
      const float ys = y[off];                                      // observation in spectrum layout, NaN = not observed
      float r = 0.0f;
      if (ys == ys) { r = ys - v; acc += r * r; }
      out[off] = r;                                                 // zero-filled spectrum of the residual
    } else if (EPI == EPI_VJP) {
      const VjpCoef k = vjp_coef(vj, chain);
      const int64_t eoff = (chain * vj.e_channels + c) * (int64_t)d * d + q;
      float gx, ge;
      vjp_elem(k, v, xt[off], vj.e[eoff], gx, ge);
      out[off] = gx;
      vj.g_e[eoff] = ge;
    } else {
      v = -(2.0f * v);
      if (apply_clip) v = v * nhmc_in1(xt[off]);
      out[off] = v;
    }
  }
  if (EPI == EPI_RESID) {
    __shared__ double red[4];
    double s[1] = {(double)acc};
    nhmc_block_sum<1>(s, red);
    if (threadIdx.x == 0) loss_ws[plane * gridDim.x + blockIdx.x] = s[0];
  }
}

// ---- pass B, d = 256 fast path ---------------------------------------------------------------------------
// Panel = 256 rows x 64 columns (256-byte row segments).  Thread (rg = t/16, cg = t%16) owns one float4 column
// group and 16 rows: first the 16 CONSECUTIVE rows rg*16 + m (stages h = 1,2,4,8 in registers), one exchange through
// LDS, then the 16 STRIDED rows rg + 16 m (stages h = 16..128 in registers) -- ascending h as in the reference, one
// LDS write + read of the panel instead of eight.
template <int EPI, bool DESC = false>
__global__ __launch_bounds__(NHMC_BLOCK) void k_fwht_cols256(const float* __restrict__ in, float* __restrict__ out,
                                                             const float* __restrict__ y, float* __restrict__ y_out,
                                                             const int32_t* __restrict__ kslot, const float* __restrict__ xt,
                                                             double* __restrict__ loss_ws, int channels, int64_t m,
                                                             int apply_clip, VjpArgs vj) {
  constexpr int D = 256, PC = 64;
  __shared__ nhmc_v4f tile[D * PC / 4];                               // [row][16 column groups]
  const int64_t plane = blockIdx.y;
  const int c0 = blockIdx.x * PC;
  const int c = (int)(plane % channels);
  const int64_t chain = plane / channels;
  const int rg = threadIdx.x >> 4, cg = threadIdx.x & 15;
  const float* __restrict__ src = in + plane * (int64_t)D * D + c0 + cg * 4;
  nhmc_v4f v[16];
  if (!DESC) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const nhmc_v4f*>(&src[(int64_t)(rg * 16 + k) * D]);
#pragma unroll
    for (int h = 1; h < 16; h <<= 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if ((k & h) == 0) {
          const nhmc_v4f a = v[k], b = v[k + h];
          v[k] = a + b;
          v[k + h] = a - b;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) tile[(rg * 16 + k) * (PC / 4) + cg] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = tile[(rg + 16 * k) * (PC / 4) + cg];
#pragma unroll
    for (int h = 1; h < 16; h <<= 1) {                                // row stride 16*h
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if ((k & h) == 0) {
          const nhmc_v4f a = v[k], b = v[k + h];
          v[k] = a + b;
          v[k + h] = a - b;
        }
      }
    }
  } else {
    // descending: the STRIDED rows rg + 16 k first (row strides 128, 64, 32, 16), exchange, then the consecutive rows
    // rg*16 + k (strides 8, 4, 2, 1); the result is left in the consecutive-row ownership
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const nhmc_v4f*>(&src[(int64_t)(rg + 16 * k) * D]);
#pragma unroll
    for (int h = 8; h >= 1; h >>= 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if ((k & h) == 0) {
          const nhmc_v4f a = v[k], b = v[k + h];
          v[k] = a + b;
          v[k + h] = a - b;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) tile[(rg + 16 * k) * (PC / 4) + cg] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = tile[(rg * 16 + k) * (PC / 4) + cg];
#pragma unroll
    for (int h = 8; h >= 1; h >>= 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if ((k & h) == 0) {
          const nhmc_v4f a = v[k], b = v[k + h];
          v[k] = a + b;
          v[k + h] = a - b;
        }
      }
    }
  }
  const float scale = 1.0f / 256.0f;
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = DESC ? rg * 16 + k : rg + 16 * k;
    const int64_t q = (int64_t)i * D + c0 + cg * 4;
    const int64_t off = plane * (int64_t)D * D + q;
    nhmc_v4f val = v[k] * scale;
    if (EPI == EPI_RAW) {
      *reinterpret_cast<nhmc_v4f*>(&out[off]) = v[k];
    } else if (EPI == EPI_STORE) {
      *reinterpret_cast<nhmc_v4f*>(&out[off]) = val;
    } else if (EPI == EPI_SCATTER) {
      const int4 ks = *reinterpret_cast<const int4*>(&kslot[q]);
      if (ks.x >= 0) y_out[chain * m + (int64_t)ks.x * channels + c] = val.x;
      if (ks.y >= 0) y_out[chain * m + (int64_t)ks.y * channels + c] = val.y;
      if (ks.z >= 0) y_out[chain * m + (int64_t)ks.z * channels + c] = val.z;
      if (ks.w >= 0) y_out[chain * m + (int64_t)ks.w * channels + c] = val.w;
    } else if (EPI == EPI_RESID) {
      const nhmc_v4f ys = *reinterpret_cast<const nhmc_v4f*>(&y[off]);   // spectrum layout, NaN = not observed
      nhmc_v4f r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float re = 0.0f;
        if (ys[e] == ys[e]) { re = ys[e] - val[e]; acc += re * re; }
        r[e] = re;
      }
      *reinterpret_cast<nhmc_v4f*>(&out[off]) = r;
    } else if (EPI == EPI_VJP) {
      const VjpCoef kc = vjp_coef(vj, chain);
      const int64_t eoff = (chain * vj.e_channels + c) * (int64_t)D * D + q;
      const nhmc_v4f xv = *reinterpret_cast<const nhmc_v4f*>(&xt[off]);
      const nhmc_v4f ev = *reinterpret_cast<const nhmc_v4f*>(&vj.e[eoff]);
      nhmc_v4f gx, ge;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a, b;
        vjp_elem(kc, val[e], xv[e], ev[e], a, b);
        gx[e] = a; ge[e] = b;
      }
      *reinterpret_cast<nhmc_v4f*>(&out[off]) = gx;
      *reinterpret_cast<nhmc_v4f*>(&vj.g_e[eoff]) = ge;
    } else {
      nhmc_v4f gq = -(2.0f * val);
      if (apply_clip) {
        const nhmc_v4f xv = *reinterpret_cast<const nhmc_v4f*>(&xt[off]);
#pragma unroll
        for (int e = 0; e < 4; ++e) gq[e] = gq[e] * nhmc_in1(xv[e]);
      }
      *reinterpret_cast<nhmc_v4f*>(&out[off]) = gq;
    }
  }
  if (EPI == EPI_RESID) {
    __shared__ double red[4];
    double sacc[1] = {(double)acc};
    nhmc_block_sum<1>(sacc, red);
    if (threadIdx.x == 0) loss_ws[plane * gridDim.x + blockIdx.x] = sacc[0];
  }
}

// ---- d = 256: forward columns + residual + adjoint columns in ONE pass -------------------------------------------------
// The forward column pass ends with a thread owning the 16 STRIDED rows rg + 16 k of its column group, and the descending
// (adjoint) column pass starts in exactly that ownership: so the residual spectrum never has to leave the registers
// between them.  ascending h = 1..8 (consecutive rows) -> LDS exchange -> h = 16..128 (strided rows) -> * 1/256,
// r = y_spec - v (0 where not observed), loss partial -> descending h = 128..16 (the same strided rows) -> LDS exchange ->
// h = 8..1 (consecutive rows) -> raw store (the descending row pass that follows scales and applies the gradient epilogue).
// One read + one write of the panel instead of two of each, one launch instead of two; every butterfly is the one the
// two-kernel form computes, in the same order: the same bits.  In place is allowed (a block owns its panel).
__global__ __launch_bounds__(NHMC_BLOCK) void k_fwht_cols256_resid_adj(const float* __restrict__ in, float* __restrict__ out,
                                                                       const float* __restrict__ y_spec,
                                                                       double* __restrict__ loss_ws) {
  constexpr int D = 256, PC = 64;
  __shared__ nhmc_v4f tile[D * PC / 4];
  const int64_t plane = blockIdx.y;
  const int c0 = blockIdx.x * PC;
  const int rg = threadIdx.x >> 4, cg = threadIdx.x & 15;
  const int64_t pbase = plane * (int64_t)D * D + c0 + cg * 4;
  nhmc_v4f v[16], ys[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const nhmc_v4f*>(&in[pbase + (int64_t)(rg * 16 + k) * D]);
#pragma unroll
  for (int k = 0; k < 16; ++k) ys[k] = *reinterpret_cast<const nhmc_v4f*>(&y_spec[pbase + (int64_t)(rg + 16 * k) * D]);
#define NHMC_BFLY(H)                                                                     \
  _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                       \
    if ((k & (H)) == 0) { const nhmc_v4f a = v[k], b = v[k + (H)]; v[k] = a + b; v[k + (H)] = a - b; } \
  }
  NHMC_BFLY(1) NHMC_BFLY(2) NHMC_BFLY(4) NHMC_BFLY(8)                  // row strides 1, 2, 4, 8
#pragma unroll
  for (int k = 0; k < 16; ++k) tile[(rg * 16 + k) * (PC / 4) + cg] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = tile[(rg + 16 * k) * (PC / 4) + cg];
  NHMC_BFLY(1) NHMC_BFLY(2) NHMC_BFLY(4) NHMC_BFLY(8)                  // row strides 16, 32, 64, 128
  const float scale = 1.0f / 256.0f;
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {                                       // ascending k, as the two-kernel form's epilogue loop
    const nhmc_v4f val = v[k] * scale;
    nhmc_v4f r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float re = 0.0f;
      if (ys[k][e] == ys[k][e]) { re = ys[k][e] - val[e]; acc += re * re; }
      r[e] = re;
    }
    v[k] = r;
  }
  NHMC_BFLY(8) NHMC_BFLY(4) NHMC_BFLY(2) NHMC_BFLY(1)                  // descending: row strides 128, 64, 32, 16
  __syncthreads();                                                     // every thread has read its strided rows of the first exchange
#pragma unroll
  for (int k = 0; k < 16; ++k) tile[(rg + 16 * k) * (PC / 4) + cg] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = tile[(rg * 16 + k) * (PC / 4) + cg];
  NHMC_BFLY(8) NHMC_BFLY(4) NHMC_BFLY(2) NHMC_BFLY(1)                  // row strides 8, 4, 2, 1
#undef NHMC_BFLY
#pragma unroll
  for (int k = 0; k < 16; ++k) *reinterpret_cast<nhmc_v4f*>(&out[pbase + (int64_t)(rg * 16 + k) * D]) = v[k];
  __shared__ double red[4];
  double sacc[1] = {(double)acc};
  nhmc_block_sum<1>(sacc, red);
  if (threadIdx.x == 0) loss_ws[plane * gridDim.x + blockIdx.x] = sacc[0];
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
bool bad(int n_chains, int channels, int dim) {
  return n_chains <= 0 || channels <= 0 || !pow2(dim) || dim < 16 || dim > 1024 || (int64_t)n_chains * channels > 65535;
}
int panel_cols(int dim) {                         // d = 256: the register fast path (64 columns); else d*pc*4 B <= 32 KiB
  if (dim == 256) return 64;
  int pc = 8192 / dim;
  return pc > dim ? dim : (pc < 8 ? 8 : pc);
}

template <int PRO, bool DESC = false, int EPI = EPI_STORE>
int rows(const float* in, const float* y, const int32_t* kslot, float* out, int n_chains, int channels, int dim,
         int64_t m, hipStream_t st, const float* xt = nullptr, int apply_clip = 0, VjpArgs vj = VjpArgs{}) {
  const int64_t total_rows = (int64_t)n_chains * channels * dim;
  const int V = dim >= 256 ? 4 : (dim >= 128 ? 2 : 1);
  const int rows_per_wave = NHMC_WAVE / (dim / V);
  const int64_t waves = (total_rows + rows_per_wave - 1) / rows_per_wave;
  dim3 grid((unsigned)((waves + 3) / 4)), block(NHMC_BLOCK);
  if (dim / V > 64) return NHMC_ERR_SHAPE;
  if (V == 4) NHMC_LAUNCH((k_fwht_rows<4, PRO, DESC, EPI>), grid, block, 0, st, in, y, kslot, out, dim, channels, m, total_rows, xt, apply_clip, vj);
  else if (V == 2) NHMC_LAUNCH((k_fwht_rows<2, PRO, DESC, EPI>), grid, block, 0, st, in, y, kslot, out, dim, channels, m, total_rows, xt, apply_clip, vj);
  else NHMC_LAUNCH((k_fwht_rows<1, PRO, DESC, EPI>), grid, block, 0, st, in, y, kslot, out, dim, channels, m, total_rows, xt, apply_clip, vj);
  return nhmc_launch_status();
}

template <int EPI, bool DESC = false>
int cols(const float* in, float* out, const float* y, float* y_out, const int32_t* kslot, const float* xt, double* ws,
         int n_chains, int channels, int dim, int64_t m, int apply_clip, hipStream_t st, VjpArgs vj = VjpArgs{}) {
  const int pc = panel_cols(dim);
  dim3 grid((unsigned)(dim / pc), (unsigned)(n_chains * channels)), block(NHMC_BLOCK);
  if (dim == 256) {
    NHMC_LAUNCH((k_fwht_cols256<EPI, DESC>), grid, block, 0, st, in, out, y, y_out, kslot, xt, ws, channels, m, apply_clip, vj);
    return nhmc_launch_status();
  }
  NHMC_LAUNCH((k_fwht_cols<EPI, DESC>), grid, block, (size_t)dim * pc * sizeof(float), st, in, out, y, y_out, kslot, xt, ws,
              dim, pc, channels, m, apply_clip, vj);
  return nhmc_launch_status();
}

}  // namespace

extern "C" int nhmc_cs_tiles(int channels, int dim) { return channels * (dim / panel_cols(dim)); }

// y = H x.  kslot: int32[dim*dim], position -> k (row of y) or -1 when perm^-1(position) >= dim^2/ratio.
// tmp: float[n_chains*channels*dim*dim].
extern "C" int nhmc_cs_H(const float* x, const int32_t* kslot, float* y, float* tmp, int n_chains, int channels,
                         int dim, int64_t m, nhmc_stream_t stream) {
  if (!x || !kslot || !y || !tmp || m <= 0) return NHMC_ERR_ARG;
  if (bad(n_chains, channels, dim) || dim > 256) return NHMC_ERR_SHAPE;
  int rc = rows<PRO_NONE>(x, nullptr, nullptr, tmp, n_chains, channels, dim, m, nhmc_s(stream));
  if (rc) return rc;
  return cols<EPI_SCATTER>(tmp, nullptr, nullptr, y, kslot, nullptr, nullptr, n_chains, channels, dim, m, 0, nhmc_s(stream));
}

// x = H^T y = H^+ y.
extern "C" int nhmc_cs_Ht(const float* y, const int32_t* kslot, float* x, float* tmp, int n_chains, int channels,
                          int dim, int64_t m, nhmc_stream_t stream) {
  if (!x || !kslot || !y || !tmp || m <= 0) return NHMC_ERR_ARG;
  if (bad(n_chains, channels, dim) || dim > 256) return NHMC_ERR_SHAPE;
  int rc = rows<PRO_GATHER>(nullptr, y, kslot, tmp, n_chains, channels, dim, m, nhmc_s(stream));
  if (rc) return rc;
  return cols<EPI_STORE>(tmp, x, nullptr, nullptr, nullptr, nullptr, nullptr, n_chains, channels, dim, m, 0, nhmc_s(stream));
}

namespace {
// forward rows were written to A; residual + both column passes; leaves the raw adjoint spectrum (column stages done) in A
int cs_columns(float* A, float* B, const float* y_spec, double* loss_ws, int n_chains, int channels, int dim, hipStream_t st) {
  if (dim == 256) {
    NHMC_LAUNCH(k_fwht_cols256_resid_adj, dim3(4u, (unsigned)(n_chains * channels)), dim3(NHMC_BLOCK), 0, st, A, A, y_spec, loss_ws);
    return nhmc_launch_status();
  }
  int rc;
  if ((rc = cols<EPI_RESID>(A, B, y_spec, nullptr, nullptr, nullptr, loss_ws, n_chains, channels, dim, 0, 0, st))) return rc;
  return cols<EPI_RAW, true>(B, A, nullptr, nullptr, nullptr, nullptr, nullptr, n_chains, channels, dim, 0, 0, st);
}
}  // namespace

// loss partials (nhmc_cs_tiles per chain) and g_xt = -2 H^T (y - H clip(xt)) (x) mask.
// y_spec: the observation in SPECTRUM layout, float[n_chains][C][dim*dim], y_spec[chain][c][perm[k]] = y[chain][k*C + c]
// and NaN at the positions that are not observed (constant over a run; the host scatters it once).
// tmp: float[n_chains*C*dim*dim] for dim = 256 (the column passes run fused and in place), twice that otherwise.
extern "C" int nhmc_data_cs(const float* xt, const float* y_spec, int apply_clip, float* g_xt, double* loss_ws, float* tmp,
                            int n_chains, int channels, int dim, nhmc_stream_t stream) {
  if (!xt || !y_spec || !g_xt || !loss_ws || !tmp) return NHMC_ERR_ARG;
  if (bad(n_chains, channels, dim) || dim > 256) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(xt) || !nhmc_aligned16(y_spec) || !nhmc_aligned16(g_xt) || !nhmc_aligned16(tmp)) return NHMC_ERR_ALIGN;
  hipStream_t st = nhmc_s(stream);
  float* A = tmp;
  float* B = tmp + (int64_t)n_chains * channels * dim * dim;
  int rc;
  if (apply_clip) { if ((rc = rows<PRO_CLIP>(xt, nullptr, nullptr, A, n_chains, channels, dim, 0, st))) return rc; }
  else            { if ((rc = rows<PRO_NONE>(xt, nullptr, nullptr, A, n_chains, channels, dim, 0, st))) return rc; }
  // adjoint in autograd's order: column stages descending, then row stages descending with the gradient epilogue
  if ((rc = cs_columns(A, B, y_spec, loss_ws, n_chains, channels, dim, st))) return rc;
  return rows<PRO_NONE, true, EPI_GRAD>(A, nullptr, nullptr, g_xt, n_chains, channels, dim, 0, st, xt, apply_clip);
}

// nhmc_data_cs on xt_next (the clipped decode of the LAST DDIM step) with that step's VJP applied in the last row
// pass: writes g_xt and channels [0, channels) of g_e.
extern "C" int nhmc_data_cs_vjp(const float* xt_next, const float* y_spec, const float* xt, const float* e, int e_channels,
                                const float* at, const float* at_next, float* g_xt, float* g_e, double* loss_ws, float* tmp,
                                int n_chains, int channels, int dim, nhmc_stream_t stream) {
  if (!xt_next || !y_spec || !xt || !e || !at || !at_next || !g_xt || !g_e || !loss_ws || !tmp) return NHMC_ERR_ARG;
  if (bad(n_chains, channels, dim) || dim > 256 || (e_channels != channels && e_channels != 2 * channels))
    return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(xt_next) || !nhmc_aligned16(xt) || !nhmc_aligned16(e) || !nhmc_aligned16(g_xt) ||
      !nhmc_aligned16(g_e) || !nhmc_aligned16(tmp) || !nhmc_aligned16(y_spec))
    return NHMC_ERR_ALIGN;
  hipStream_t st = nhmc_s(stream);
  float* A = tmp;
  float* B = tmp + (int64_t)n_chains * channels * dim * dim;
  int rc;
  if ((rc = rows<PRO_NONE>(xt_next, nullptr, nullptr, A, n_chains, channels, dim, 0, st))) return rc;
  if ((rc = cs_columns(A, B, y_spec, loss_ws, n_chains, channels, dim, st))) return rc;
  const VjpArgs vj{e, g_e, at, at_next, e_channels};
  return rows<PRO_NONE, true, EPI_VJP>(A, nullptr, nullptr, g_xt, n_chains, channels, dim, 0, st, xt, 0, vj);
}
