// Codebook lookup of the VQ-f4 first stage on decode (latent variant).
//
// Reference: `differentiable_decode_first_stage` (ldm/models/diffusion/ddpm.py:766-820) calls
// `VQModelInterface.decode(z, force_not_quantize=False)` (ldm/models/autoencoder.py:274-283), whose first step is
// taming-transformers' `VectorQuantizer2.forward` (environment.yml:24 pins taming-transformers==0.0.1; the package is
// not vendored in the reference).  Its published arithmetic, per latent pixel z in R^D against the codebook e_k:
//     d_k = (sum_c z_c^2 + sum_c e_kc^2) - 2 * sum_c z_c e_kc ;  k* = argmin_k d_k (first minimum)
//     (the two norms are sequential fp32 sums, the inner product an FMA chain over c: torch's CPU kernels, bit for bit)
//     z_q = z + (e_k* - z)            (the straight-through form: value of z + (z_q - z).detach())
// torch materialises d as an [n_pixels, n_embed] fp32 matrix (4096 x 8192 x 4 B = 134 MB per chain, written once and
// re-read by the argmin).  Here the codebook is staged through LDS in chunks of VQ_CHUNK codes as (e_0..e_{D-1}, |e|^2)
// and four lanes (one per wave of the block) share a pixel, each walking a quarter of every chunk: every lane of a wave
// reads the same LDS address (broadcast, no bank conflicts), nothing but z, idx and z_q touches HBM (algorithmic bytes:
// 2T_z + 4 B per pixel).
#include "nhmc_common.h"

namespace {

constexpr int VQ_CHUNK = 2048;          // codes per LDS chunk: 2048 x 16 B = 32 KB (+8 KB of norms for D = 4)

constexpr int VQ_PIX = 64;              // latent pixels per block: one per lane; the block's 4 waves split every codebook chunk

template <int D>
__global__ __launch_bounds__(NHMC_BLOCK) void k_vq_nearest(
    const float* __restrict__ z, const float* __restrict__ codebook, float* __restrict__ z_q,
    int32_t* __restrict__ idx_out, int64_t n_pix, int64_t hw, int n_embed) {
  __shared__ float4 code[VQ_CHUNK];
  __shared__ float norm[D == 4 ? VQ_CHUNK : 1];
  __shared__ float part_d[4][VQ_PIX];
  __shared__ int part_k[4][VQ_PIX];
  // 16 x 4096 latent pixels are only one wave per SIMD if a thread owns a pixel and walks the whole codebook (then VALU
  // and LDS latencies are exposed: 247 us measured); so a pixel is shared by 4 lanes, one in each wave of the block, and
  // wave w walks quarter w of every chunk -- 4 waves per SIMD, same broadcast LDS reads.
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int64_t p = (int64_t)blockIdx.x * VQ_PIX + lane;
  const bool live = p < n_pix;
  const int64_t b = live ? p / hw : 0, pos = live ? p % hw : 0;
  const float* zp = z + b * D * hw + pos;
  float zc[D];
  float zz = 0.0f;
#pragma unroll
  for (int c = 0; c < D; ++c) {
    zc[c] = live ? zp[(int64_t)c * hw] : 0.0f;
    zz = c == 0 ? zc[0] * zc[0] : zz + zc[c] * zc[c];
  }
  // torch.argmin's answers for non-finite rows (a divergent trajectory decoded through the first stage): a NaN distance
  // counts as the minimum and the FIRST one wins; an all-inf row has its first minimum at index 0.  `seen_nan` freezes
  // the first NaN; the sentinel index of a wave that never updated is resolved after the merge.
  float best = INFINITY;
  int best_k = 0x7fffffff;
  bool seen_nan = false;
  constexpr int QUARTER = VQ_CHUNK / 4;
  for (int k0 = 0; k0 < n_embed; k0 += VQ_CHUNK) {
    __syncthreads();
    for (int k = threadIdx.x; k < VQ_CHUNK; k += NHMC_BLOCK) {
      float e[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      float ee = INFINITY;                                          // padding codes can never win
      if (k0 + k < n_embed) {
#pragma unroll
        for (int c = 0; c < D; ++c) e[c] = codebook[(int64_t)(k0 + k) * D + c];
        ee = e[0] * e[0];
#pragma unroll
        for (int c = 1; c < D; ++c) ee = ee + e[c] * e[c];
      }
      if constexpr (D == 4) { code[k] = make_float4(e[0], e[1], e[2], e[3]); norm[k] = ee; }
      else                  { code[k] = make_float4(e[0], e[1], e[2], ee); }
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < QUARTER; ++kk) {
      const int k = part * QUARTER + kk;
      const float4 c4 = code[k];
      float dot = zc[0] * c4.x;                          // the -2 z.e term is a [n,D] x [D,n_embed] sgemm in the reference:
      dot = __fmaf_rn(zc[1], c4.y, dot);                 // a k-ordered FMA chain (bit-identical to torch's CPU sgemm,
      dot = __fmaf_rn(zc[2], c4.z, dot);                 // measured), so the argmin takes the same decisions
      float ee = c4.w;
      if constexpr (D == 4) { dot = __fmaf_rn(zc[3], c4.w, dot); ee = norm[k]; }
      const float d = (zz + ee) - 2.0f * dot;
      const bool dn = d != d;
      if ((d < best) || (dn && !seen_nan)) {             // within a wave the code index only grows: first minimum kept
        best = d; best_k = k0 + k; seen_nan = seen_nan || dn;
      }
    }
  }
  part_d[part][lane] = best;
  part_k[part][lane] = best_k;
  __syncthreads();
  if (part != 0 || !live) return;
#pragma unroll
  for (int w = 1; w < 4; ++w) {                          // first minimum over the whole codebook: smallest d, then smallest index
    const float d = part_d[w][lane];
    const int k = part_k[w][lane];
    const bool dn = d != d, bn = best != best;
    if (dn ? (!bn || k < best_k) : (!bn && (d < best || (d == best && k < best_k)))) { best = d; best_k = k; }
  }
  if ((unsigned)best_k >= (unsigned)n_embed) best_k = 0;   // no distance below +inf anywhere: argmin of an all-inf row
  if (idx_out) idx_out[p] = best_k;
  float* qp = z_q + b * D * hw + pos;
#pragma unroll
  for (int c = 0; c < D; ++c) {
    const float e = codebook[(int64_t)best_k * D + c];
    qp[(int64_t)c * hw] = zc[c] + (e - zc[c]);
  }
}

}  // namespace

extern "C" int nhmc_vq_nearest(const float* z, const float* codebook, float* z_q, int32_t* idx, int n_chains,
                               int channels, int64_t hw, int n_embed, nhmc_stream_t stream) {
  if (!z || !codebook || !z_q || n_chains <= 0 || hw <= 0 || n_embed <= 0) return NHMC_ERR_ARG;
  if (channels != 3 && channels != 4) return NHMC_ERR_SHAPE;
  const int64_t n_pix = (int64_t)n_chains * hw;
  const dim3 grid((unsigned)((n_pix + VQ_PIX - 1) / VQ_PIX));
  if (channels == 3)
    NHMC_LAUNCH(k_vq_nearest<3>, grid, dim3(NHMC_BLOCK), 0, nhmc_s(stream), z, codebook, z_q, idx, n_pix, hw, n_embed);
  else
    NHMC_LAUNCH(k_vq_nearest<4>, grid, dim3(NHMC_BLOCK), 0, nhmc_s(stream), z, codebook, z_q, idx, n_pix, hw, n_embed);
  return nhmc_launch_status();
}
