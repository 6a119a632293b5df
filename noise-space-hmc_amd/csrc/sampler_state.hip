// Per-chain sampler state on the device (rows a1, a5, a6, a7 of SURVEY.md section 8):
// Hamiltonian, Metropolis test, accept/reject bookkeeping + sigma_y / eps / tau schedules,
// sample collection, PSNR, and the counter-based momentum / accept noise.
// Replaces main_sampling.py:683-689 (schedule), :692 (momentum draw), :697,717-720 (H, dH, accept),
// :721-749 (bookkeeping), :738-739 (PSNR).  The reference does all of this with host Python scalars
// and a `.item()` sync per trajectory; here it is per chain and stays on the stream.
#include "nhmc_common.h"

namespace {

// ---- a5 -------------------------------------------------------------------------------------
__global__ void k_hamiltonian(const double* __restrict__ sums_ws, int tiles, const double* __restrict__ loss,
                              const double* __restrict__ sigma_y, double m_inv, float* __restrict__ H_out,
                              double* __restrict__ terms, int n_chains, const int32_t* __restrict__ sel = nullptr,
                              int64_t loss_stride = 0) {
  const int chain = blockIdx.x * (blockDim.x / NHMC_WAVE) + (threadIdx.x >> 6);
  if (chain >= n_chains) return;
  const int lane = threadIdx.x & 63;
  double sx = 0.0, sp = 0.0;
  for (int t = lane; t < tiles; t += NHMC_WAVE) {
    sx += sums_ws[((int64_t)chain * tiles + t) * 2 + 0];
    sp += sums_ws[((int64_t)chain * tiles + t) * 2 + 1];
  }
  sx = nhmc_wave_sum(sx);
  sp = nhmc_wave_sum(sp);
  if (lane == 0) {
    const double s = sigma_y[chain];
    const float kf = (float)(1.0 / (2.0 * (s * s)));
    const double lossv = sel ? loss[(int64_t)(sel[chain] & 1) * loss_stride + chain] : loss[chain];
    const float Sx = (float)sx, Sp = (float)sp, L = (float)lossv, mi = (float)m_inv;
    // (1/2)*sum(x^2) + k*loss + (1/2)*sum(p*p)*m^-1, left to right in fp32 (main_sampling.py:697)
    const float a = 0.5f * Sx;
    const float b = kf * L;
    const float c = (0.5f * Sp) * mi;
    H_out[chain] = (a + b) + c;
    if (terms) { terms[chain * 3 + 0] = sx; terms[chain * 3 + 1] = sp; terms[chain * 3 + 2] = lossv; }
  }
}

// ---- a6 -------------------------------------------------------------------------------------
__global__ void k_metropolis(const float* __restrict__ H0, const float* __restrict__ H1, const float* __restrict__ u,
                             const int32_t* __restrict__ active, int32_t* __restrict__ accept,
                             float* __restrict__ dH, int n_chains) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chains) return;
  const float d = H1[c] - H0[c];
  // min(1, exp(-dH)).  Deliberate deviation: a NaN dH REJECTS here.  (In the reference Python's
  // min(tensor(1), tensor(nan)) returns 1, so a NaN proposal would be accepted and poison the chain.)
  const float ratio = (d != d) ? 0.0f : fminf(1.0f, expf(-d));
  const bool act = active ? active[c] != 0 : true;
  accept[c] = (act && (u[c] < ratio)) ? 1 : 0;
  if (dH) dH[c] = d;
}

// ---- a7 -------------------------------------------------------------------------------------
__global__ void k_schedule_begin(const int32_t* __restrict__ epoch, double* __restrict__ tau, double* __restrict__ eps,
                                 double* __restrict__ sigma_y, double* __restrict__ eps_eff,
                                 int32_t* __restrict__ active, double sigma_0, int epochs, int sampling,
                                 int n_chains) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chains) return;
  const int ep = epoch[c];
  const bool act = ep < epochs + 2 * sampling;
  if (act) {
    if (ep < epochs) {
      const double f = 1.0 - (double)ep / (double)epochs;
      sigma_y[c] = sigma_0 + 1.6 * (f * f);             // sigma_0 + 1.6*(1 - epoch/epochs)**2
    } else if (ep == epochs) {
      sigma_y[c] = sigma_0;
      if (tau[c] > 0.1) { tau[c] = 0.1; eps[c] = 0.01; }
    }
  }
  active[c] = act ? 1 : 0;
  eps_eff[c] = act ? eps[c] : 0.0;
}

__global__ void k_schedule_end(const int32_t* __restrict__ accept, const int32_t* __restrict__ active,
                               int32_t* __restrict__ epoch, int32_t* __restrict__ rejected, double* __restrict__ tau,
                               double* __restrict__ eps, int32_t* __restrict__ n_accept,
                               int32_t* __restrict__ n_reject, int n_chains) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chains || !active[c]) return;
  if (accept[c]) {
    rejected[c] = 0;
    epoch[c] += 1;
    if (n_accept) n_accept[c] += 1;
  } else {
    rejected[c] += 1;
    if (n_reject) n_reject[c] += 1;
    if (rejected[c] >= 2) { tau[c] = tau[c] * 0.95; eps[c] = eps[c] * 0.95; }
  }
}

__global__ __launch_bounds__(NHMC_BLOCK) void k_accept_commit(
    const int32_t* __restrict__ accept, const int32_t* __restrict__ epoch, float4* __restrict__ x,
    const float4* __restrict__ x_prop, const float4* __restrict__ xt_prop, float4* __restrict__ samples, int epochs,
    int sampling, int64_t n4) {
  const int chain = blockIdx.y;
  if (!accept[chain]) return;                      // wave-uniform: whole block leaves
  const int slot = epoch[chain] - (epochs + sampling);
  const bool keep = samples && slot >= 0 && slot < sampling;
  const int64_t base = (int64_t)chain * n4;
  const int64_t sbase = ((int64_t)chain * sampling + (keep ? slot : 0)) * n4;
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * NHMC_VEC_PER_THREAD) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    nhmc_stnt(&x[base + q], nhmc_ldnt(&x_prop[base + q]));
    if (keep) nhmc_stnt(&samples[sbase + q], nhmc_ldnt(&xt_prop[base + q]));
  }
}

// ---- latent variant (main_sampling_latent.py:691-733) ------------------------------------------
// Accepted chains: in the final phase the PREVIOUS accepted decode is appended to a ring of the last `keep`
// samples (:709 appends x_accept before :713 reassigns it), then xt_last <- xt_prop and x <- x_prop.
__global__ __launch_bounds__(NHMC_BLOCK) void k_latent_commit(
    const int32_t* __restrict__ accept, const int32_t* __restrict__ has_prev, const int32_t* __restrict__ count,
    int final_phase, int keep, float4* __restrict__ x, const float4* __restrict__ x_prop, float4* __restrict__ xt_last,
    const float4* __restrict__ xt_prop, float4* __restrict__ samples, int64_t n4) {
  const int chain = blockIdx.y;
  if (!accept[chain]) return;
  const bool push = final_phase && has_prev[chain] && samples;
  const int64_t base = (int64_t)chain * n4;
  const int64_t sbase = ((int64_t)chain * keep + (push ? count[chain] % keep : 0)) * n4;
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * NHMC_VEC_PER_THREAD) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    if (push) nhmc_stnt(&samples[sbase + q], nhmc_ldnt(&xt_last[base + q]));
    nhmc_stnt(&xt_last[base + q], nhmc_ldnt(&xt_prop[base + q]));
    nhmc_stnt(&x[base + q], nhmc_ldnt(&x_prop[base + q]));
  }
}

__global__ void k_schedule_end_latent(const int32_t* __restrict__ accept, int32_t* __restrict__ rejected,
                                      double* __restrict__ tau, double* __restrict__ eps, double* __restrict__ sigma_y,
                                      int32_t* __restrict__ count, int32_t* __restrict__ has_prev,
                                      int32_t* __restrict__ n_accept, double sigma_y_on_accept, int final_phase,
                                      int n_chains) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chains) return;
  if (accept[c]) {
    rejected[c] = 0;
    sigma_y[c] = sigma_y_on_accept;                       // :693-695 (annealing) or :705-706 (sigma_0)
    if (final_phase) {
      tau[c] = 0.1; eps[c] = 0.01;                        // :707-708
      if (has_prev[c]) count[c] += 1;                     // :709
    }
    has_prev[c] = 1;
    if (n_accept) n_accept[c] += 1;
  } else {
    rejected[c] += 1;
    if (rejected[c] >= 2) { tau[c] = tau[c] * 0.9; eps[c] = eps[c] * 0.9; rejected[c] = 0; }   // :728-732
  }
}

// ---- PSNR -----------------------------------------------------------------------------------
__global__ __launch_bounds__(NHMC_BLOCK) void k_psnr_partial(const float4* __restrict__ xt,
                                                             const float4* __restrict__ xo, double* __restrict__ ws,
                                                             int64_t n4) {
  const int chain = blockIdx.y;
  const int64_t base = (int64_t)chain * n4;
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * NHMC_VEC_PER_THREAD) + threadIdx.x;
  float acc = 0.0f;
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    const float4 a = nhmc_ldnt(&xt[base + q]), b = nhmc_ldnt(&xo[base + q]);
    const float* ae = reinterpret_cast<const float*>(&a);
    const float* be = reinterpret_cast<const float*>(&b);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float ua = fminf(fmaxf((ae[c] + 1.0f) / 2.0f, 0.0f), 1.0f);   // inverse_data_transform
      const float ub = fminf(fmaxf((be[c] + 1.0f) / 2.0f, 0.0f), 1.0f);
      const float d = ua - ub;
      acc += d * d;
    }
  }
  __shared__ double red[4];
  double v[1] = {(double)acc};
  nhmc_block_sum<1>(v, red);
  if (threadIdx.x == 0) ws[(int64_t)chain * gridDim.x + blockIdx.x] = v[0];
}

__global__ void k_psnr_final(const double* __restrict__ ws, int tiles, int64_t n_elem, float* __restrict__ psnr,
                             int n_chains) {
  const int chain = blockIdx.x * (blockDim.x / NHMC_WAVE) + (threadIdx.x >> 6);
  if (chain >= n_chains) return;
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (int t = lane; t < tiles; t += NHMC_WAVE) acc += ws[(int64_t)chain * tiles + t];
  acc = nhmc_wave_sum(acc);
  if (lane == 0) {
    const float mse = (float)(acc / (double)n_elem);
    psnr[chain] = 10.0f * log10f(1.0f / mse);
  }
}

// ---- a1: Philox4x32-10 ------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                             uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__device__ __forceinline__ void philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                              uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__device__ __forceinline__ float unit24(uint32_t r) { return ((float)(r >> 8) + 0.5f) * 5.9604644775390625e-08f; }

__global__ __launch_bounds__(NHMC_BLOCK) void k_randn(float4* __restrict__ out, uint32_t k0, uint32_t k1,
                                                      uint32_t chain_id0, uint32_t draw, float scale, int64_t n4) {
  const int chain = blockIdx.y;
  const int64_t t0 = (int64_t)blockIdx.x * (NHMC_BLOCK * NHMC_VEC_PER_THREAD) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < NHMC_VEC_PER_THREAD; ++i) {
    const int64_t q = t0 + (int64_t)i * NHMC_BLOCK;
    if (q >= n4) continue;
    uint32_t c0 = (uint32_t)q, c1 = chain_id0 + (uint32_t)chain, c2 = draw, c3 = 0u;
    philox4x32_10(c0, c1, c2, c3, k0, k1);
    const float ra = sqrtf(-2.0f * logf(unit24(c0))), aa = 6.283185307179586f * unit24(c1);
    const float rb = sqrtf(-2.0f * logf(unit24(c2))), ab = 6.283185307179586f * unit24(c3);
    float4 o;
    o.x = (ra * cosf(aa)) * scale; o.y = (ra * sinf(aa)) * scale;
    o.z = (rb * cosf(ab)) * scale; o.w = (rb * sinf(ab)) * scale;
    nhmc_stnt(&out[(int64_t)chain * n4 + q], o);
  }
}

__global__ void k_uniform(float* __restrict__ out, uint32_t k0, uint32_t k1, uint32_t chain_id0, uint32_t draw,
                          int n_chains) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chains) return;
  uint32_t c0 = 0u, c1 = chain_id0 + (uint32_t)c, c2 = draw, c3 = 1u;
  philox4x32_10(c0, c1, c2, c3, k0, k1);
  out[c] = unit24(c0);
}

inline dim3 small_grid(int n) { return dim3((unsigned)((n + 255) / 256)); }
inline dim3 wave_grid(int n) { return dim3((unsigned)((n + 3) / 4)); }

}  // namespace

extern "C" int nhmc_abi_version(void) { return NHMC_ABI_VERSION; }

extern "C" const char* nhmc_last_launch_error(void) { return hipGetErrorString(nhmc_last_hip_error); }

extern "C" const char* nhmc_status_string(int status) {
  switch (status) {
    case NHMC_OK: return "ok";
    case NHMC_ERR_ARG: return "bad argument (null pointer, non-positive size or unknown mode)";
    case NHMC_ERR_ALIGN: return "pointer not 16-byte aligned or element count not a multiple of 4";
    case NHMC_ERR_SHAPE: return "unsupported shape";
    case NHMC_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown status";
  }
}

extern "C" int nhmc_hamiltonian(const double* sums_ws, int tiles, const double* loss, const double* sigma_y,
                                double m_inv, float* H_out, double* terms, int n_chains, nhmc_stream_t stream) {
  if (!sums_ws || !loss || !sigma_y || !H_out || tiles <= 0 || n_chains <= 0) return NHMC_ERR_ARG;
  NHMC_LAUNCH(k_hamiltonian, wave_grid(n_chains), dim3(256), 0, nhmc_s(stream), sums_ws, tiles, loss, sigma_y,
                     m_inv, H_out, terms, n_chains, (const int32_t*)nullptr, (int64_t)0);
  return nhmc_launch_status();
}

// H at the accepted position with the loss taken from the gradient cache: loss_pair[sel[chain]][chain]
extern "C" int nhmc_hamiltonian_cached(const double* sums_ws, int tiles, const double* loss_pair, const int32_t* sel,
                                       int64_t loss_stride, const double* sigma_y, double m_inv, float* H_out,
                                       double* terms, int n_chains, nhmc_stream_t stream) {
  if (!sums_ws || !loss_pair || !sel || !sigma_y || !H_out || tiles <= 0 || n_chains <= 0 || loss_stride < n_chains)
    return NHMC_ERR_ARG;
  NHMC_LAUNCH(k_hamiltonian, wave_grid(n_chains), dim3(256), 0, nhmc_s(stream), sums_ws, tiles, loss_pair, sigma_y,
              m_inv, H_out, terms, n_chains, sel, loss_stride);
  return nhmc_launch_status();
}

extern "C" int nhmc_metropolis(const float* H0, const float* H1, const float* u, const int32_t* active,
                               int32_t* accept, float* dH, int n_chains, nhmc_stream_t stream) {
  if (!H0 || !H1 || !u || !accept || n_chains <= 0) return NHMC_ERR_ARG;
  NHMC_LAUNCH(k_metropolis, small_grid(n_chains), dim3(256), 0, nhmc_s(stream), H0, H1, u, active, accept, dH,
                     n_chains);
  return nhmc_launch_status();
}

extern "C" int nhmc_schedule_begin(const int32_t* epoch, double* tau, double* eps, double* sigma_y, double* eps_eff,
                                   int32_t* active, double sigma_0, int epochs, int sampling, int n_chains,
                                   nhmc_stream_t stream) {
  if (!epoch || !tau || !eps || !sigma_y || !eps_eff || !active || epochs <= 0 || sampling < 0 || n_chains <= 0)
    return NHMC_ERR_ARG;
  NHMC_LAUNCH(k_schedule_begin, small_grid(n_chains), dim3(256), 0, nhmc_s(stream), epoch, tau, eps, sigma_y,
                     eps_eff, active, sigma_0, epochs, sampling, n_chains);
  return nhmc_launch_status();
}

extern "C" int nhmc_schedule_end(const int32_t* accept, const int32_t* active, int32_t* epoch, int32_t* rejected,
                                 double* tau, double* eps, int32_t* n_accept, int32_t* n_reject, int n_chains,
                                 nhmc_stream_t stream) {
  if (!accept || !active || !epoch || !rejected || !tau || !eps || n_chains <= 0) return NHMC_ERR_ARG;
  NHMC_LAUNCH(k_schedule_end, small_grid(n_chains), dim3(256), 0, nhmc_s(stream), accept, active, epoch,
                     rejected, tau, eps, n_accept, n_reject, n_chains);
  return nhmc_launch_status();
}

extern "C" int nhmc_accept_commit(const int32_t* accept, const int32_t* epoch, float* x, const float* x_prop,
                                  const float* xt_prop, float* samples, int epochs, int sampling, int n_chains,
                                  int64_t n_elem, nhmc_stream_t stream) {
  if (!accept || !epoch || !x || !x_prop || n_chains <= 0 || n_elem <= 0) return NHMC_ERR_ARG;
  if (samples && !xt_prop) return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  if ((n_elem & 3) || !nhmc_aligned16(x) || !nhmc_aligned16(x_prop) || !nhmc_aligned16(xt_prop) ||
      !nhmc_aligned16(samples))
    return NHMC_ERR_ALIGN;
  dim3 grid((unsigned)nhmc_leapfrog_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  NHMC_LAUNCH(k_accept_commit, grid, block, 0, nhmc_s(stream), accept, epoch, (float4*)x,
                     (const float4*)x_prop, (const float4*)xt_prop, (float4*)samples, epochs, sampling, n_elem / 4);
  return nhmc_launch_status();
}

extern "C" int nhmc_latent_commit(const int32_t* accept, const int32_t* has_prev, const int32_t* count,
                                  int final_phase, int keep, float* x, const float* x_prop, float* xt_last,
                                  const float* xt_prop, float* samples, int n_chains, int64_t n_elem,
                                  nhmc_stream_t stream) {
  if (!accept || !has_prev || !count || !x || !x_prop || !xt_last || !xt_prop || keep <= 0 || n_chains <= 0 ||
      n_elem <= 0)
    return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  if ((n_elem & 3) || !nhmc_aligned16(x) || !nhmc_aligned16(x_prop) || !nhmc_aligned16(xt_last) ||
      !nhmc_aligned16(xt_prop) || !nhmc_aligned16(samples))
    return NHMC_ERR_ALIGN;
  dim3 grid((unsigned)nhmc_leapfrog_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  NHMC_LAUNCH(k_latent_commit, grid, block, 0, nhmc_s(stream), accept, has_prev, count, final_phase, keep, (float4*)x,
              (const float4*)x_prop, (float4*)xt_last, (const float4*)xt_prop, (float4*)samples, n_elem / 4);
  return nhmc_launch_status();
}

extern "C" int nhmc_schedule_end_latent(const int32_t* accept, int32_t* rejected, double* tau, double* eps,
                                        double* sigma_y, int32_t* count, int32_t* has_prev, int32_t* n_accept,
                                        double sigma_y_on_accept, int final_phase, int n_chains,
                                        nhmc_stream_t stream) {
  if (!accept || !rejected || !tau || !eps || !sigma_y || !count || !has_prev || n_chains <= 0) return NHMC_ERR_ARG;
  NHMC_LAUNCH(k_schedule_end_latent, small_grid(n_chains), dim3(256), 0, nhmc_s(stream), accept, rejected, tau, eps,
              sigma_y, count, has_prev, n_accept, sigma_y_on_accept, final_phase, n_chains);
  return nhmc_launch_status();
}

extern "C" int nhmc_psnr(const float* xt, const float* x_orig, float* psnr, double* ws, int n_chains,
                         int64_t n_elem, nhmc_stream_t stream) {
  if (!xt || !x_orig || !psnr || !ws || n_chains <= 0 || n_elem <= 0) return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  if ((n_elem & 3) || !nhmc_aligned16(xt) || !nhmc_aligned16(x_orig)) return NHMC_ERR_ALIGN;
  const int tiles = nhmc_data_tiles(n_elem);
  NHMC_LAUNCH(k_psnr_partial, dim3((unsigned)tiles, (unsigned)n_chains), dim3(NHMC_BLOCK), 0, nhmc_s(stream),
                     (const float4*)xt, (const float4*)x_orig, ws, n_elem / 4);
  NHMC_LAUNCH(k_psnr_final, wave_grid(n_chains), dim3(256), 0, nhmc_s(stream), ws, tiles, n_elem, psnr,
                     n_chains);
  return nhmc_launch_status();
}

extern "C" int nhmc_randn_philox(float* out, uint64_t seed, uint32_t chain_id0, uint32_t draw, float scale,
                                 int n_chains, int64_t n_elem, nhmc_stream_t stream) {
  if (!out || n_chains <= 0 || n_elem <= 0) return NHMC_ERR_ARG;
  if (n_chains > 65535) return NHMC_ERR_SHAPE;
  if ((n_elem & 3) || !nhmc_aligned16(out)) return NHMC_ERR_ALIGN;
  dim3 grid((unsigned)nhmc_leapfrog_tiles(n_elem), (unsigned)n_chains), block(NHMC_BLOCK);
  NHMC_LAUNCH(k_randn, grid, block, 0, nhmc_s(stream), (float4*)out, (uint32_t)(seed & 0xffffffffu),
                     (uint32_t)(seed >> 32), chain_id0, draw, scale, n_elem / 4);
  return nhmc_launch_status();
}

extern "C" int nhmc_uniform_philox(float* out, uint64_t seed, uint32_t chain_id0, uint32_t draw, int n_chains,
                                   nhmc_stream_t stream) {
  if (!out || n_chains <= 0) return NHMC_ERR_ARG;
  NHMC_LAUNCH(k_uniform, small_grid(n_chains), dim3(256), 0, nhmc_s(stream), out, (uint32_t)(seed & 0xffffffffu),
                     (uint32_t)(seed >> 32), chain_id0, draw, n_chains);
  return nhmc_launch_status();
}
