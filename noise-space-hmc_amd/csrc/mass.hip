// Diagonal-mass HMC (row f.4 of SURVEY.md section 8): `hmc_test_conditioning`, main_sampling.py:776-894.
//   * fused leapfrog update with a per-element mass: p = z*std (momentum draw, :819), kinetic sum p^2/M (:824,852),
//     p -= eps G (:841), x += eps p / M (:834), plus the Welford mean / M2 update of the trajectory positions
//     (:843-847) folded into the same pass (x is in registers anyway);
//   * mass rebuild from the rank transform of the trajectory variance (:857-870): variance = M2/(L-1), ascending
//     sort per chain (rocPRIM segmented radix sort through hipCUB), M = exp(2 rank/(N-1) - 1);
//   * its sigma_y schedule (:808-816).
// HBM-bound: MID reads x,p,g,inv_M (+mean,M2) and writes x,p (+mean,M2): 6T (10T with Welford).
#include <hipcub/hipcub.hpp>

#include "nhmc_common.h"

namespace {

template <int MODE, bool HAS_G2>
__global__ __launch_bounds__(NHMC_BLOCK) void k_leapfrog_mass(
    float4* __restrict__ x, float4* __restrict__ p, const float4* __restrict__ z, const float4* __restrict__ g,
    const float4* __restrict__ g2, const float4* __restrict__ inv_m, const float4* __restrict__ std_m,
    const double* __restrict__ eps, const double* __restrict__ sigma_y, const int32_t* __restrict__ welford_on,
    float4* __restrict__ mean, float4* __restrict__ m2, int l, int64_t n4, double* __restrict__ sums_ws) {
  const int chain = blockIdx.y;
  const double e = eps[chain], s = sigma_y[chain];
  const float kf = (float)(1.0 / (2.0 * (s * s))), ef = (float)e, eh = (float)(e / 2.0);
  const bool wel = MODE != NHMC_LF_FIRST && welford_on && welford_on[chain] != 0;
  const float inv_cnt_div = (float)(l + 1);
  const int64_t base = (int64_t)chain * n4;
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * NHMC_VEC_PER_THREAD) + threadIdx.x;
  float sx = 0.0f, sp = 0.0f;
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    float4 xv = nhmc_ldnt(&x[base + q]), gv = nhmc_ldnt(&g[base + q]), iv = nhmc_ldnt(&inv_m[base + q]);
    float4 pv = MODE == NHMC_LF_FIRST ? nhmc_ldnt(&z[base + q]) : nhmc_ldnt(&p[base + q]);
    if (HAS_G2) {
      const float4 h = nhmc_ldnt(&g2[base + q]);
      gv.x += h.x; gv.y += h.y; gv.z += h.z; gv.w += h.w;
    }
    float4 sv = make_float4(1.f, 1.f, 1.f, 1.f), mv = make_float4(0.f, 0.f, 0.f, 0.f), qv = mv;
    if (MODE == NHMC_LF_FIRST) sv = nhmc_ldnt(&std_m[base + q]);
    if (wel && l > 0) { mv = nhmc_ldnt(&mean[base + q]); qv = nhmc_ldnt(&m2[base + q]); }
    float* xe = reinterpret_cast<float*>(&xv);
    float* pe = reinterpret_cast<float*>(&pv);
    const float* ge = reinterpret_cast<const float*>(&gv);
    const float* ie = reinterpret_cast<const float*>(&iv);
    const float* se = reinterpret_cast<const float*>(&sv);
    float* me = reinterpret_cast<float*>(&mv);
    float* qe = reinterpret_cast<float*>(&qv);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (MODE == NHMC_LF_FIRST) {
        pe[c] = pe[c] * se[c];                                   // randn * std_diag
        sx += xe[c] * xe[c];
        sp += ie[c] * (pe[c] * pe[c]);                           // inv_M * p**2
      }
      const float G = xe[c] + kf * ge[c];
      if (MODE == NHMC_LF_FIRST) {
        pe[c] = pe[c] - eh * G;
      } else {
        pe[c] = pe[c] - ef * G;
        if (MODE == NHMC_LF_LAST) pe[c] = pe[c] + eh * G;
      }
      if (wel) {                                                 // Welford on the CURRENT position (:843-847)
        const float delta = xe[c] - me[c];
        me[c] = me[c] + delta / inv_cnt_div;
        const float delta2 = xe[c] - me[c];
        qe[c] = qe[c] + delta * delta2;
      }
      if (MODE == NHMC_LF_LAST) {
        sx += xe[c] * xe[c];
        sp += ie[c] * (pe[c] * pe[c]);
      } else {
        xe[c] = xe[c] + (ef * pe[c]) * ie[c];                    // x + epsilon * p * inv_M
      }
    }
    nhmc_stnt(&p[base + q], pv);
    if (MODE != NHMC_LF_LAST) nhmc_stnt(&x[base + q], xv);
    if (wel) { nhmc_stnt(&mean[base + q], mv); nhmc_stnt(&m2[base + q], qv); }
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

__global__ void k_variance_keys(const float* __restrict__ m2, float div, float* __restrict__ keys, int32_t* __restrict__ vals,
                                int32_t* __restrict__ offsets, int n_chains, int64_t n_elem) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)n_chains * n_elem;
  if (i < total) { keys[i] = m2[i] / div; vals[i] = (int32_t)(i % n_elem); }
  if (i <= n_chains) offsets[i] = (int32_t)(i * n_elem);
}

// The mass of the element of rank r is a constant of (r, N): the caller hands in std_table[r] = sqrt(M_r) and
// inv_table[r] = 1 / M_r with M_r = exp(2 r/(N-1) - 1), evaluated ON THE HOST with the reference's own tensor
// expressions (main_sampling.py:863-868), so the transcendental is the reference host's libm and not this GPU's expf
// (whose last bit differs and which a 183-trajectory run amplifies to 1e-3).
__global__ void k_mass_from_ranks(const int32_t* __restrict__ sorted_idx, const int32_t* __restrict__ flags,
                                  const float* __restrict__ std_table, const float* __restrict__ inv_table,
                                  float* __restrict__ inv_m, float* __restrict__ std_m, int64_t n_elem) {
  const int chain = blockIdx.y;
  if (flags && !flags[chain]) return;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_elem) return;
  const int64_t idx = sorted_idx[(int64_t)chain * n_elem + r];
  std_m[(int64_t)chain * n_elem + idx] = std_table[r];
  inv_m[(int64_t)chain * n_elem + idx] = inv_table[r];
}

__global__ void k_schedule_begin_mass(const int32_t* __restrict__ epoch, double* __restrict__ tau, double* __restrict__ eps,
                                      double* __restrict__ sigma_y, double* __restrict__ eps_eff,
                                      int32_t* __restrict__ active, int32_t* __restrict__ welford_on,
                                      const double* __restrict__ sigma_table, int burn, int epochs, int sampling,
                                      int n_chains) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chains) return;
  const int ep = epoch[c];
  const bool act = ep < burn + epochs + 4 * sampling;
  if (act) {
    // sigma_table[e], e = 0..epochs: the host evaluates :808-813 in Python floats (the `** 3` there is libm pow)
    if (ep < epochs) {
      sigma_y[c] = sigma_table[ep];
    } else if (ep == epochs) {
      sigma_y[c] = sigma_table[epochs];
      if (tau[c] > 0.1) { tau[c] = 0.1; eps[c] = 0.01; }
    }
  }
  active[c] = act ? 1 : 0;
  eps_eff[c] = act ? eps[c] : 0.0;
  welford_on[c] = (act && (ep - burn) > epochs / 3) ? 1 : 0;
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

size_t cub_temp_bytes(int n_chains, int64_t n_elem) {
  size_t bytes = 0;
  (void)hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, bytes, (const float*)nullptr, (float*)nullptr,
                                                    (const int32_t*)nullptr, (int32_t*)nullptr, (int)(n_chains * n_elem),
                                                    n_chains, (const int32_t*)nullptr, (const int32_t*)nullptr);
  return bytes;
}

}  // namespace

extern "C" int nhmc_leapfrog_mass(int mode, float* x, float* p, const float* z, const float* g, const float* g2,
                                  const float* inv_m, const float* std_m, const double* eps, const double* sigma_y,
                                  const int32_t* welford_on, float* mean, float* m2, int l, int n_chains,
                                  int64_t n_elem, double* sums_ws, nhmc_stream_t stream) {
  if (!x || !p || !g || !inv_m || !eps || !sigma_y || n_chains <= 0 || n_elem <= 0 || l < 0) return NHMC_ERR_ARG;
  if (mode == NHMC_LF_FIRST && (!z || !std_m)) return NHMC_ERR_ARG;
  if (mode != NHMC_LF_MID && !sums_ws) return NHMC_ERR_ARG;
  if (welford_on && (!mean || !m2)) return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  if ((n_elem & 3) || !nhmc_aligned16(x) || !nhmc_aligned16(p) || !nhmc_aligned16(z) || !nhmc_aligned16(g) ||
      !nhmc_aligned16(g2) || !nhmc_aligned16(inv_m) || !nhmc_aligned16(std_m) || !nhmc_aligned16(mean) || !nhmc_aligned16(m2))
    return NHMC_ERR_ALIGN;
  dim3 grid((unsigned)nhmc_leapfrog_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  hipStream_t st = nhmc_s(stream);
#define NHMC_LM(MODE, G2)                                                                                          \
  NHMC_LAUNCH((k_leapfrog_mass<MODE, G2>), grid, block, 0, st, (float4*)x, (float4*)p, (const float4*)z, (const float4*)g, \
              (const float4*)g2, (const float4*)inv_m, (const float4*)std_m, eps, sigma_y, welford_on, (float4*)mean,       \
              (float4*)m2, l, n_elem / 4, sums_ws)
  switch (mode) {
    case NHMC_LF_FIRST: if (g2) NHMC_LM(NHMC_LF_FIRST, true); else NHMC_LM(NHMC_LF_FIRST, false); break;
    case NHMC_LF_MID:   if (g2) NHMC_LM(NHMC_LF_MID, true);   else NHMC_LM(NHMC_LF_MID, false);   break;
    case NHMC_LF_LAST:  if (g2) NHMC_LM(NHMC_LF_LAST, true);  else NHMC_LM(NHMC_LF_LAST, false);  break;
    default: return NHMC_ERR_ARG;
  }
#undef NHMC_LM
  return nhmc_launch_status();
}

extern "C" size_t nhmc_mass_sort_ws_bytes(int n_chains, int64_t n_elem) {
  if (n_chains <= 0 || n_elem <= 0) return 0;
  const size_t tot = (size_t)n_chains * (size_t)n_elem;
  return 4 * align_up(tot * 4) + align_up((size_t)(n_chains + 1) * 4) + align_up(cub_temp_bytes(n_chains, n_elem));
}

extern "C" int nhmc_mass_from_variance(const float* m2, int L, const int32_t* flags, const float* std_table,
                                       const float* inv_table, float* inv_m, float* std_m, void* ws, size_t ws_bytes,
                                       int n_chains, int64_t n_elem, nhmc_stream_t stream) {
  if (!m2 || !std_table || !inv_table || !inv_m || !std_m || !ws || n_chains <= 0 || n_elem <= 1) return NHMC_ERR_ARG;
  if (n_chains > 65535 || (int64_t)n_chains * n_elem > 0x7fffffffLL) return NHMC_ERR_SHAPE;
  if (ws_bytes < nhmc_mass_sort_ws_bytes(n_chains, n_elem)) return NHMC_ERR_ARG;
  hipStream_t st = nhmc_s(stream);
  const size_t tot = (size_t)n_chains * (size_t)n_elem, blk = align_up(tot * 4);
  char* base = static_cast<char*>(ws);
  float* keys_in = reinterpret_cast<float*>(base);
  float* keys_out = reinterpret_cast<float*>(base + blk);
  int32_t* vals_in = reinterpret_cast<int32_t*>(base + 2 * blk);
  int32_t* vals_out = reinterpret_cast<int32_t*>(base + 3 * blk);
  int32_t* offsets = reinterpret_cast<int32_t*>(base + 4 * blk);
  void* temp = base + 4 * blk + align_up((size_t)(n_chains + 1) * 4);
  size_t temp_bytes = cub_temp_bytes(n_chains, n_elem);
  NHMC_LAUNCH(k_variance_keys, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, m2, (float)(L - 1), keys_in, vals_in,
              offsets, n_chains, n_elem);
  if (nhmc_launch_status()) return NHMC_ERR_LAUNCH;
  (void)hipGetLastError();
  if (hipcub::DeviceSegmentedRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)tot, n_chains,
                                                  offsets, offsets + 1, 0, 32, st) != hipSuccess)
    return NHMC_ERR_LAUNCH;
  NHMC_LAUNCH(k_mass_from_ranks, dim3((unsigned)((n_elem + 255) / 256), (unsigned)n_chains), dim3(256), 0, st, vals_out,
              flags, std_table, inv_table, inv_m, std_m, n_elem);
  return nhmc_launch_status();
}

extern "C" int nhmc_schedule_begin_mass(const int32_t* epoch, double* tau, double* eps, double* sigma_y, double* eps_eff,
                                        int32_t* active, int32_t* welford_on, const double* sigma_table, int burn,
                                        int epochs, int sampling, int n_chains, nhmc_stream_t stream) {
  if (!epoch || !tau || !eps || !sigma_y || !eps_eff || !active || !welford_on || !sigma_table || epochs <= 0 || n_chains <= 0)
    return NHMC_ERR_ARG;
  NHMC_LAUNCH(k_schedule_begin_mass, dim3((unsigned)((n_chains + 255) / 256)), dim3(256), 0, nhmc_s(stream), epoch, tau, eps,
              sigma_y, eps_eff, active, welford_on, sigma_table, burn, epochs, sampling, n_chains);
  return nhmc_launch_status();
}
