// Spectral (anisotropic-blur) operator as a chain of fp32-MFMA GEMMs (row a15 of SURVEY.md section 8).
// Replaces obs_functions/Hfuncs.py:448-523 (Deblurring2D.H / Ht / H_pinv) and, fused with the residual
// and the clip mask, the data term main_sampling.py:693-695 / :709-711 for deg = deblur_aniso.
//
// The reference operator is NOT a separable stencil: its tiled-singulars / interleaved-Vt layout makes
//     out_c = Lo (D_c o (L^T X_c R)) Ro^T          (d x d factors, per-channel multiplier map D_c)
// i.e. four dense d^3 products per channel image (d = 256: 134 MFLOP each way).  That is MFMA work.
// 1e-4 relative parity rules out bf16, and gfx950 has no xf32, so the products run on
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 64 FLOP/clk/SIMD).
//
// One kernel, C_img = A * B per d x d image, with one operand the shared factor and the other the image:
//   LEFT  : C = S^T-layout * X     A(i,k) = S[k][i]   B(k,j) = X[k][j]
//   RIGHT : C = X * S              A(i,k) = X[i][k]   B(k,j) = S[k][j]
// (so "multiply by M from the left" passes M^T's memory, "by M from the right" passes M's memory; the host
// keeps both orientations of the four factors resident -- 2 MB).
// Block tile T x T (T = 128: 4 waves as 2x2, each 2x2 MFMA tiles; 64: 4 waves, 1 tile each; 32: one wave),
// K step 32 staged through LDS: k-major tiles so every fragment read is 32 consecutive floats per half-wave
// (conflict-free ds_read_b32); the RIGHT form's image tile is kept row-major with a +1 pad instead.
// Epilogues fuse the spectral multiplier, the residual + per-tile loss partial, and the -2 * mask scaling.
#include "nhmc_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_NONE = 0, EPI_MULD = 1, EPI_RESID = 2, EPI_GRAD = 3 };
constexpr int BK = 32;

template <int T, int NW, bool LEFT, int EPI, bool PRECLIP>
__global__ __launch_bounds__(64 * NW * NW) void k_sgemm(
    const float* __restrict__ X, const float* __restrict__ S, float* __restrict__ Cout,
    const float* __restrict__ Dmap, const float* __restrict__ aux, double* __restrict__ ws, int d, int channels) {
  constexpr int NT = 64 * NW * NW;
  constexpr int FR = T / (32 * NW);
  constexpr int ALD = LEFT ? T : BK + 1;
  __shared__ float As[LEFT ? BK * T : T * (BK + 1)];
  __shared__ float Bs[BK * T];

  const int img = blockIdx.z, ti = blockIdx.y * T, tj = blockIdx.x * T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave / NW, wj = wave % NW;
  const int lr = lane & 31, lh = lane >> 5;
  const float* __restrict__ Ximg = X + (int64_t)img * d * d;

  f32x16 acc[FR][FR];
#pragma unroll
  for (int a = 0; a < FR; ++a)
#pragma unroll
    for (int b = 0; b < FR; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  for (int k0 = 0; k0 < d; k0 += BK) {
    // ---- stage A ----
    if (LEFT) {
#pragma unroll
      for (int v = 0; v < (BK * T / 4) / NT; ++v) {
        const int idx = tid + v * NT, kk = idx / (T / 4), c4 = idx % (T / 4);
        *reinterpret_cast<float4*>(&As[kk * T + c4 * 4]) =
            *reinterpret_cast<const float4*>(&S[(int64_t)(k0 + kk) * d + ti + c4 * 4]);
      }
    } else {
#pragma unroll
      for (int v = 0; v < (BK * T / 4) / NT; ++v) {
        const int idx = tid + v * NT, r = idx / (BK / 4), c4 = idx % (BK / 4);
        const float4 val = *reinterpret_cast<const float4*>(&Ximg[(int64_t)(ti + r) * d + k0 + c4 * 4]);
        float* dst = &As[r * (BK + 1) + c4 * 4];
        dst[0] = val.x; dst[1] = val.y; dst[2] = val.z; dst[3] = val.w;
      }
    }
    // ---- stage B ----
    {
      const float* __restrict__ src = LEFT ? Ximg : S;
#pragma unroll
      for (int v = 0; v < (BK * T / 4) / NT; ++v) {
        const int idx = tid + v * NT, kk = idx / (T / 4), c4 = idx % (T / 4);
        float4 val = *reinterpret_cast<const float4*>(&src[(int64_t)(k0 + kk) * d + tj + c4 * 4]);
        if (LEFT && PRECLIP) {
          val.x = nhmc_clip1(val.x); val.y = nhmc_clip1(val.y); val.z = nhmc_clip1(val.z); val.w = nhmc_clip1(val.w);
        }
        *reinterpret_cast<float4*>(&Bs[kk * T + c4 * 4]) = val;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < BK; kk += 2) {
      float a[FR], b[FR];
#pragma unroll
      for (int f = 0; f < FR; ++f) {
        const int row = (wi * FR + f) * 32 + lr;
        a[f] = LEFT ? As[(kk + lh) * ALD + row] : As[row * ALD + kk + lh];
        b[f] = Bs[(kk + lh) * T + (wj * FR + f) * 32 + lr];
      }
#pragma unroll
      for (int fa = 0; fa < FR; ++fa)
#pragma unroll
        for (int fb = 0; fb < FR; ++fb)
          acc[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fa], b[fb], acc[fa][fb], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D map col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  const int c = img % channels;
  float lsum = 0.0f;
#pragma unroll
  for (int fa = 0; fa < FR; ++fa)
#pragma unroll
    for (int fb = 0; fb < FR; ++fb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = ti + (wi * FR + fa) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = tj + (wj * FR + fb) * 32 + lr;
        const int64_t off = (int64_t)row * d + col;
        float v = acc[fa][fb][r];
        if (EPI == EPI_MULD) v = v * Dmap[(int64_t)c * d * d + off];
        if (EPI == EPI_RESID) {
          v = aux[(int64_t)img * d * d + off] - v;        // r = y - H x
          lsum += v * v;
        }
        if (EPI == EPI_GRAD) {
          v = -(2.0f * v);
          if (aux) v = v * nhmc_in1(aux[(int64_t)img * d * d + off]);
        }
        Cout[(int64_t)img * d * d + off] = v;
      }
  if (EPI == EPI_RESID) {
    __shared__ double red[NW * NW];
    double s = nhmc_wave_sum((double)lsum);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
      for (int w = 0; w < NW * NW; ++w) tot += red[w];
      // tiles of one chain are contiguous: [chain][channel][tile_row][tile_col]
      const int tiles_side = d / T;
      ws[((int64_t)img * tiles_side + blockIdx.y) * tiles_side + blockIdx.x] = tot;
    }
  }
}

template <bool LEFT, int EPI, bool PRECLIP>
int gemm(const float* X, const float* S, float* Cout, const float* Dmap, const float* aux, double* ws, int n_img,
         int channels, int d, hipStream_t st) {
  if (d % 128 == 0) {
    dim3 grid(d / 128, d / 128, n_img);
    NHMC_LAUNCH((k_sgemm<128, 2, LEFT, EPI, PRECLIP>), grid, dim3(256), 0, st, X, S, Cout, Dmap, aux, ws, d, channels);
  } else if (d % 64 == 0) {
    dim3 grid(d / 64, d / 64, n_img);
    NHMC_LAUNCH((k_sgemm<64, 2, LEFT, EPI, PRECLIP>), grid, dim3(256), 0, st, X, S, Cout, Dmap, aux, ws, d, channels);
  } else {
    dim3 grid(d / 32, d / 32, n_img);
    NHMC_LAUNCH((k_sgemm<32, 1, LEFT, EPI, PRECLIP>), grid, dim3(64), 0, st, X, S, Cout, Dmap, aux, ws, d, channels);
  }
  return nhmc_launch_status();
}

int tile_of(int d) { return d % 128 == 0 ? 128 : (d % 64 == 0 ? 64 : 32); }

bool bad(int n_chains, int channels, int dim) {
  return n_chains <= 0 || channels <= 0 || dim <= 0 || (dim % 32) || (int64_t)n_chains * channels > 65535;
}

}  // namespace

extern "C" int nhmc_spectral_tiles(int channels, int dim) {
  const int t = dim / tile_of(dim);
  return channels * t * t;
}

// out_c = Lo (D_c o (L^T X_c R)) Ro^T.  Arguments: L and R as stored ([d][d] row-major), LoT = Lo^T and
// RoT = Ro^T as stored (see the layout note at the top).
extern "C" int nhmc_spectral_apply(const float* x, const float* L, const float* R, const float* Dmap,
                                   const float* LoT, const float* RoT, float* out, float* tmp, int n_chains,
                                   int channels, int dim, nhmc_stream_t stream) {
  if (!x || !L || !R || !Dmap || !LoT || !RoT || !out || !tmp) return NHMC_ERR_ARG;
  if (bad(n_chains, channels, dim)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(x) || !nhmc_aligned16(L) || !nhmc_aligned16(R) || !nhmc_aligned16(LoT) ||
      !nhmc_aligned16(RoT) || !nhmc_aligned16(out) || !nhmc_aligned16(tmp))
    return NHMC_ERR_ALIGN;
  hipStream_t st = nhmc_s(stream);
  const int n = n_chains * channels;
  int rc;
  if ((rc = gemm<true, EPI_NONE, false>(x, L, tmp, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc;
  if ((rc = gemm<false, EPI_MULD, false>(tmp, R, out, Dmap, nullptr, nullptr, n, channels, dim, st))) return rc;
  if ((rc = gemm<true, EPI_NONE, false>(out, LoT, tmp, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc;
  return gemm<false, EPI_NONE, false>(tmp, RoT, out, nullptr, nullptr, nullptr, n, channels, dim, st);
}

// Data term for the spectral operator.  `factors` is the packed resident block [8][d][d] (row-major):
// U1, U2, V1, V2, U1^T, U2^T, V1^T, V2^T.
extern "C" int nhmc_data_spectral(const float* xt, const float* y, const float* factors, const float* Dmap,
                                  int apply_clip, float* g_xt, double* loss_ws, float* tmp, int n_chains,
                                  int channels, int dim, nhmc_stream_t stream) {
  if (!xt || !y || !factors || !Dmap || !g_xt || !loss_ws || !tmp) return NHMC_ERR_ARG;
  if (bad(n_chains, channels, dim)) return NHMC_ERR_SHAPE;
  const int64_t dd = (int64_t)dim * dim;
  if (!nhmc_aligned16(xt) || !nhmc_aligned16(y) || !nhmc_aligned16(factors) || !nhmc_aligned16(g_xt) ||
      !nhmc_aligned16(tmp))
    return NHMC_ERR_ALIGN;
  const float *U1 = factors, *U2 = factors + dd, *V1 = factors + 2 * dd, *V2 = factors + 3 * dd;
  const float *U1T = U1 + 4 * dd, *U2T = U1 + 5 * dd, *V1T = U1 + 6 * dd, *V2T = U1 + 7 * dd;
  hipStream_t st = nhmc_s(stream);
  const int n = n_chains * channels;
  float* A = tmp;
  float* B = tmp + (int64_t)n * dd;
  int rc;
  // r = y - U1 (D o (V1^T clip(xt) V2)) U2^T
  if (apply_clip) { if ((rc = gemm<true, EPI_NONE, true>(xt, V1, A, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc; }
  else            { if ((rc = gemm<true, EPI_NONE, false>(xt, V1, A, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc; }
  if ((rc = gemm<false, EPI_MULD, false>(A, V2, B, Dmap, nullptr, nullptr, n, channels, dim, st))) return rc;
  if ((rc = gemm<true, EPI_NONE, false>(B, U1T, A, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc;
  if ((rc = gemm<false, EPI_RESID, false>(A, U2T, B, nullptr, y, loss_ws, n, channels, dim, st))) return rc;
  // g = -2 V1 (D o (U1^T r U2)) V2^T  (x) mask
  if ((rc = gemm<true, EPI_NONE, false>(B, U1, A, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc;
  if ((rc = gemm<false, EPI_MULD, false>(A, U2, B, Dmap, nullptr, nullptr, n, channels, dim, st))) return rc;
  if ((rc = gemm<true, EPI_NONE, false>(B, V1T, A, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc;
  return gemm<false, EPI_GRAD, false>(A, V2T, g_xt, nullptr, apply_clip ? xt : nullptr, nullptr, n, channels, dim, st);
}
