/* nhmc.h -- C ABI of the MI355X-native noise-space HMC hot path (libnhmc.so).
 *
 * Every entry point
 *   - takes raw DEVICE pointers, sizes, per-chain scalar arrays and a hipStream_t
 *     (passed as void*; NULL = the null stream);
 *   - is asynchronous on that stream, allocates nothing, never synchronises;
 *   - returns an int status (NHMC_OK = 0); no C++ exception crosses the boundary;
 *   - may be called concurrently on different streams / devices.
 *
 * "Reference" below is Sunsett5/Noise-space-HMC; file:line are relative to its root.
 * The reference has no FFI: its hot path is Python issuing ATen ops.  Each function
 * names the reference lines whose arithmetic it replaces; INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Layout conventions
 *   images   : fp32, [n_chains][C][H][W] contiguous (NCHW), n_elem = C*H*W per chain,
 *              n_elem % 4 == 0 and base pointers 16-byte aligned (NHMC_ERR_ALIGN otherwise)
 *   score    : fp32, [n_chains][e_channels][H][W], e_channels in {C, 2C}; only the first
 *              C channels are read (learned-sigma channels ignored, algos/unconditional.py:17-18)
 *   y        : fp32, [n_chains][M]
 *   per-chain scalars : device arrays of length n_chains (double for the sampler's
 *              Python-float quantities eps / sigma_y, float for alpha-bar)
 *   partial-sum workspaces : double, sized by the matching *_ws_bytes(); reductions are
 *              two-pass and deterministic (no float atomics)
 */
#ifndef NHMC_H
#define NHMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nhmc_stream_t; /* hipStream_t */

enum {
  NHMC_OK = 0,
  NHMC_ERR_ARG = 1,    /* null pointer / non-positive size / unknown mode */
  NHMC_ERR_ALIGN = 2,  /* pointer not 16-byte aligned or n_elem % 4 != 0 */
  NHMC_ERR_SHAPE = 3,  /* shape combination the kernel does not cover */
  NHMC_ERR_LAUNCH = 4  /* hipGetLastError() after launch != hipSuccess */
};

#define NHMC_ABI_VERSION 2
int nhmc_abi_version(void);
const char* nhmc_status_string(int status);
/* HIP's message for the last NHMC_ERR_LAUNCH raised on the calling thread (diagnostics only). */
const char* nhmc_last_launch_error(void);

/* ------------------------------------------------------------------------------------
 * a1-a4  Fused leapfrog update            main_sampling.py:702,706-707,713,715
 *
 *   G  = x + (1/(2 sigma_y^2)) * (g [+ g2])
 *   NHMC_LF_FIRST : Sx,Sp partials of (x,p) BEFORE the update (for H at :697);
 *                   p -= (eps/2) G ; x += (eps/m) p
 *   NHMC_LF_MID   : p -= eps G     ; x += (eps/m) p          <- the north-star kernel, 5T/chain
 *   NHMC_LF_LAST  : p -= eps G ; p += (eps/2) G ; x unchanged;
 *                   Sx,Sp partials AFTER the update (for H at :717)
 * fp32 op order and scalar rounding follow the reference (scalars are fp64 products
 * rounded once to fp32).  g2 (nullable) is a second gradient contribution added to g
 * before use (the U-Net input-gradient of the first DDIM step), saving a separate add pass.
 * sums_ws: double[n_chains][nhmc_leapfrog_tiles(n_elem)][2]; written in FIRST/LAST, may be
 * NULL in MID.
 * ---------------------------------------------------------------------------------- */
enum { NHMC_LF_FIRST = 0, NHMC_LF_MID = 1, NHMC_LF_LAST = 2 };
int nhmc_leapfrog_tiles(int64_t n_elem);
size_t nhmc_leapfrog_ws_bytes(int n_chains, int64_t n_elem);
int nhmc_leapfrog_fused(int mode, float* x, float* p, const float* g, const float* g2,
                        const double* eps, const double* sigma_y, double m_inv,
                        int n_chains, int64_t n_elem, double* sums_ws, nhmc_stream_t stream);

/* NHMC_LF_FIRST out of place in x: reads x_in, writes the proposal to x_out (x_in != x_out) and updates p in place.
 * The caller's accepted position survives the trajectory, so a reject needs no saved copy (main_sampling.py:699
 * clones x for that).  Same traffic as the in-place form. */
int nhmc_leapfrog_first(const float* x_in, float* x_out, float* p, const float* g, const float* g2,
                        const double* eps, const double* sigma_y, double m_inv,
                        int n_chains, int64_t n_elem, double* sums_ws, nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Gradient cache: one decode + gradient per leapfrog step, none for the first half step.
 *
 * The reference evaluates the decode, loss and gradient at the accepted position x at the top of every outer
 * iteration (main_sampling.py:693-695), although that point was already evaluated: by the previous iteration's last
 * leapfrog step if its proposal was accepted (x is that proposal, :709-711,731), by the previous iteration's own
 * :693-695 if it was rejected (x did not move).  Loss and gradient do not depend on sigma_y / eps (those enter inside
 * the update), so the values are identical and reusing them changes no bit while removing 1 of L+1 score-network
 * ladders per trajectory.
 *
 * Layout: g_pair = float[2][pair_stride] (pair_stride >= n_chains_total * n_elem, % 4 == 0), slot s of chain c at
 * g_pair + s*pair_stride + c*n_elem, holding the SUMMED gradient g + g2 (the fp32 add the update kernels perform
 * first anyway); loss_pair = double[2][loss_stride]; sel = int32[n_chains], the slot that belongs to the accepted
 * position.  All pointers are those of the launch's first chain (chunk views just offset them).
 *   nhmc_grad_cache_store        slot (sel ^ flip) <- (g + g2, loss)          (the once-per-run evaluation at x0)
 *   nhmc_leapfrog_first_cached   NHMC_LF_FIRST (out of place) with g read from slot sel
 *   nhmc_leapfrog_last_cached    NHMC_LF_LAST, and slot 1 - sel <- (g + g2, loss) of the end point
 *   nhmc_hamiltonian_cached      nhmc_hamiltonian with loss = loss_pair[sel]
 *   nhmc_grad_cache_flip         sel ^= accept                                 (after the Metropolis test)
 * ---------------------------------------------------------------------------------- */
int nhmc_grad_cache_store(const float* g, const float* g2, const double* loss, float* g_pair, double* loss_pair,
                          const int32_t* sel, int flip, int64_t pair_stride, int64_t loss_stride,
                          int n_chains, int64_t n_elem, nhmc_stream_t stream);
int nhmc_leapfrog_first_cached(const float* x_in, float* x_out, float* p, const float* g_pair, const int32_t* sel,
                               int64_t pair_stride, const double* eps, const double* sigma_y, double m_inv,
                               int n_chains, int64_t n_elem, double* sums_ws, nhmc_stream_t stream);
int nhmc_leapfrog_last_cached(float* x, float* p, const float* g, const float* g2, float* g_pair, const int32_t* sel,
                              int64_t pair_stride, const double* loss, double* loss_pair, int64_t loss_stride,
                              const double* eps, const double* sigma_y, double m_inv,
                              int n_chains, int64_t n_elem, double* sums_ws, nhmc_stream_t stream);
int nhmc_grad_cache_flip(const int32_t* accept, int32_t* sel, int n_chains, nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a9-a10  DDIM mix, forward               algos/unconditional.py:17-28, main_sampling.py:709
 *   u = (xt - e*sqrt(1-at)) / sqrt(at);  x0 = clip(u,-1,1);  add = sqrt(1-at_next)*e;
 *   xt_next = sqrt(at_next)*x0 + add;    if final_clip: xt_next = clip(xt_next,-1,1)
 * Any of xt_next / x0_t / add_up may be NULL (not written): the fused sampler writes only
 * xt_next (3T), the plugin surface's cal_x0 writes x0_t and add_up.
 * at / at_next: float[n_chains] (the reference's [n,1,1,1] alpha-bar tensors).
 * ---------------------------------------------------------------------------------- */
int nhmc_ddim_mix_fwd(const float* xt, const float* e, int e_channels,
                      const float* at, const float* at_next, int final_clip,
                      float* xt_next, float* x0_t, float* add_up,
                      int n_chains, int channels, int64_t hw, nhmc_stream_t stream);

/* map_back alone: xt_next = sqrt(at_next)*x0_t + add_up      algos/unconditional.py:26-28 */
int nhmc_ddim_map_back(const float* x0_t, const float* add_up, const float* at_next,
                       float* xt_next, int n_chains, int64_t n_elem, nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a11  DDIM mix, backward (the VJP autograd builds at main_sampling.py:695,711)
 *   gin = gout (+ gout2);  if final_clip: gin *= 1[-1 <= xt_next <= 1]
 *   g_xt = ((gin*sqrt(at_next)) * 1[-1<=u<=1]) / sqrt(at)
 *   g_e[:, :C] = sqrt(1-at_next)*gin + (-g_xt)*sqrt(1-at);   g_e[:, C:] = 0
 * g_e has e_channels channels (what the score network's backward consumes).  fill_sigma = 1 writes the zero
 * sigma-channels; fill_sigma = 0 leaves channels [C, e_channels) untouched (caller keeps a pre-zeroed buffer: -T).
 * g_e may be NULL: the score-path gradient is then not formed (a score evaluated without gradient, as the latent
 * model's apply_model is: ldm/models/diffusion/ddpm.py:892), -T of writes.
 * g_x0 (nullable; then gout2 == NULL and final_clip == 0): split form for the plugin surface,
 * where cal_x0 and map_back are differentiated separately -- gout is then d/d add_up and g_x0
 * replaces gin*sqrt(at_next) as the gradient reaching x0_t.
 * ---------------------------------------------------------------------------------- */
int nhmc_ddim_mix_bwd(const float* gout, const float* gout2, const float* g_x0, const float* xt,
                      const float* e, int e_channels, const float* at, const float* at_next, int final_clip,
                      float* g_xt, float* g_e, int fill_sigma,
                      int n_chains, int channels, int64_t hw, nhmc_stream_t stream);

/* a11 + a12/a13 fused: VJP of the LAST DDIM step (final clip included) with the inpainting data term computed on the
 * fly from (xt, e): r = y[slot] - clip(xt_next), loss partials (nhmc_leapfrog_tiles(n_elem) per chain), gin = -2 r.
 * Replaces nhmc_data_inpaint + nhmc_ddim_mix_bwd(final_clip = 1) for deg = inpaint_* (same bits, -3T of traffic). */
int nhmc_ddim_mix_bwd_inpaint(const float* xt, const float* e, int e_channels, const float* at,
                              const float* at_next, const float* y, const int32_t* slot, int64_t m,
                              float* g_xt, float* g_e, int fill_sigma, double* loss_ws, int n_chains,
                              int channels, int64_t hw, nhmc_stream_t stream);

/* The same for a WHOLE-PIXEL mask (every pixel keeps all its channels or none, as main_sampling.py:290-305 builds them):
 * mask_words[hw/32] has bit (p % 32) of word p / 32 set when pixel p (row-major) is kept, prefix[hw/32] is the number of
 * kept pixels before each word, and y index = channels * rank(p) + channel.  Replaces the T-per-chain slot stream by
 * 16 KB of tables; hw % 32 == 0.  Same bits as nhmc_ddim_mix_bwd_inpaint (g_xt, g_e; the loss to fp64 rounding).
 * loss_ws: nhmc_inpaint_px_tiles(channels, hw) partials per chain (one per tile of a channel plane). */
int nhmc_inpaint_px_tiles(int channels, int64_t hw);
int nhmc_ddim_mix_bwd_inpaint_px(const float* xt, const float* e, int e_channels, const float* at,
                                 const float* at_next, const float* y, const uint32_t* mask_words,
                                 const int32_t* prefix, int64_t m, float* g_xt, float* g_e, int fill_sigma,
                                 double* loss_ws, int n_chains, int channels, int64_t hw, nhmc_stream_t stream);

/* a11 + a12/a14 fused: the same for the super-resolution operator (obs_functions/Hfuncs.py:180-234), ratio in
 * {2,4,8,16}: r = y - blockmean(clip(xt_next)), loss partials (nhmc_sr_vjp_tiles(channels, dim, ratio) per chain),
 * gin = -2 r / ratio^2.  Replaces nhmc_data_sr + nhmc_ddim_mix_bwd(final_clip = 1) (same bits, -3T of traffic).
 * Writes channels [0, channels) of g_e only: the caller keeps the sigma-channels of g_e zero.
 * ratio 4 runs one float4 per thread and stream with the four rows of a block of pixels on the four waves of a
 * workgroup (nhmc_sr_vjp_tiles partials per chain: more, smaller tiles than nhmc_sr_tiles). */
int nhmc_sr_vjp_tiles(int channels, int dim, int ratio);
int nhmc_ddim_mix_bwd_sr(const float* xt, const float* e, int e_channels, const float* at, const float* at_next,
                         const float* y, int ratio, float* g_xt, float* g_e, double* loss_ws, int n_chains,
                         int channels, int dim, nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a12-a14  Data term: loss_b = sum (y_b - H clip(xt_b))^2 and d loss / d xt
 *                                          main_sampling.py:693-695,709-711
 * All write g_xt = -2 H^T r (masked by 1[-1<=xt<=1] when apply_clip) densely (no memset
 * needed) and per-tile fp64 partials of sum r^2 into loss_ws; finish with
 * nhmc_sum_partials(loss_ws, tiles, n_chains, loss).
 *
 * inpaint (obs_functions/Hfuncs.py:119-154): slot[j], j in CHW order, is the index into y of
 *   pixel-element j, or -1 if that element is masked out.
 * sr (Hfuncs.py:180-234): ratio r in {2,4,8,16,32}, W % 4 == 0, y laid out [C][H/r][W/r].
 * ---------------------------------------------------------------------------------- */
int nhmc_data_tiles(int64_t n_elem);                 /* tiles written by nhmc_data_inpaint / nhmc_psnr */
int nhmc_sr_tiles(int channels, int dim, int ratio); /* tiles written by nhmc_data_sr */
size_t nhmc_data_ws_bytes(int n_chains, int64_t n_elem);
int nhmc_data_inpaint(const float* xt, const float* y, const int32_t* slot, int apply_clip,
                      float* g_xt, double* loss_ws,
                      int n_chains, int64_t n_elem, int64_t m, nhmc_stream_t stream);
int nhmc_data_sr(const float* xt, const float* y, int ratio, int apply_clip,
                 float* g_xt, double* loss_ws,
                 int n_chains, int channels, int dim, nhmc_stream_t stream);
int nhmc_sum_partials(const double* ws, int tiles, int stride, int offset, int n_chains,
                      double* out, nhmc_stream_t stream);

/* Operator surface H / H^T / H^+ (Hfuncs.py:65-90) for the same two operators.
 * inpaint: kept_chw[k] = CHW address of y entry k.  scale: 1/r^2 for H^T, 1 for H^+. */
int nhmc_inpaint_H(const float* x, const int32_t* kept_chw, float* y,
                   int n_chains, int64_t n_elem, int64_t m, nhmc_stream_t stream);
int nhmc_inpaint_Ht(const float* y, const int32_t* slot, float* x,
                    int n_chains, int64_t n_elem, int64_t m, nhmc_stream_t stream);
int nhmc_sr_H(const float* x, float* y, int ratio, int n_chains, int channels, int dim,
              nhmc_stream_t stream);
int nhmc_sr_Ht(const float* y, float* x, int ratio, float scale, int n_chains, int channels,
               int dim, nhmc_stream_t stream);

/* Colorization (Hfuncs.py:655-695).  w: HOST array of channels + 2 floats (channels <= 4; passed by value to the
 * kernel): the SVD (u, s, V) of the 1 x C grey row as the reference holds it -- V[0,0] .. V[C-1,0], s, U[0,0].
 *   H x = u * (s * ((v_0 x_0 + v_1 x_1) + v_2 x_2)),  H^T y = v_c * (s * (u * y)),  H^+ y = v_c * ((u * y) * (1 / s)),
 * each product and sum rounded in this order (torch's CPU ops on the reference's composition: same bits).
 * loss partials: nhmc_color_tiles(hw) per chain. */
int nhmc_color_tiles(int64_t hw);
int nhmc_data_color(const float* xt, const float* y, const float* w, int apply_clip, float* g_xt,
                    double* loss_ws, int n_chains, int channels, int64_t hw, nhmc_stream_t stream);
/* nhmc_data_color fused with the VJP of the LAST DDIM step (as nhmc_ddim_mix_bwd_inpaint / _sr): writes g_xt and
 * channels [0, channels) of g_e. */
int nhmc_ddim_mix_bwd_color(const float* xt, const float* e, int e_channels, const float* at, const float* at_next,
                            const float* y, const float* w, float* g_xt, float* g_e, double* loss_ws, int n_chains,
                            int channels, int64_t hw, nhmc_stream_t stream);
int nhmc_color_H(const float* x, const float* w, float* y, int n_chains, int channels, int64_t hw,
                 nhmc_stream_t stream);
int nhmc_color_Ht(const float* y, const float* w, int pinv, float* x, int n_chains, int channels, int64_t hw,
                  nhmc_stream_t stream);                                    /* pinv: 0 = H^T, 1 = H^+ */

/* Walsh-Hadamard compressive sensing (Hfuncs.py:611-651): y[k*C + c] = (FWHT(x_c) / d)[perm[k]], k < d*d/ratio;
 * H^T = H^+.  kslot: int32[d*d], position -> k or -1 (inverse of perm restricted to the kept rows).  m = row length
 * of y (= C*d*d/ratio).  dim: power of two, 16..256.  tmp: float[n_chains*C*d*d].
 * loss partials: nhmc_cs_tiles(channels, dim) per chain.
 * The data term takes the observation in SPECTRUM layout: y_spec = float[n_chains][C][d*d] with
 * y_spec[chain][c][perm[k]] = y[chain][k*C + c] and NaN at the positions that are not observed -- constant over a run,
 * scattered once by the host -- so that the residual is a coalesced read instead of a gather per element, and at d = 256
 * the forward column pass, the residual and the adjoint's column pass run as ONE kernel on a register-resident panel
 * (three passes over the image instead of four).  tmp for the data term: float[n_chains*C*d*d] at d = 256, twice that
 * otherwise. */
int nhmc_cs_tiles(int channels, int dim);
int nhmc_cs_H(const float* x, const int32_t* kslot, float* y, float* tmp, int n_chains, int channels, int dim,
              int64_t m, nhmc_stream_t stream);
int nhmc_cs_Ht(const float* y, const int32_t* kslot, float* x, float* tmp, int n_chains, int channels, int dim,
               int64_t m, nhmc_stream_t stream);
int nhmc_data_cs(const float* xt, const float* y_spec, int apply_clip, float* g_xt, double* loss_ws, float* tmp,
                 int n_chains, int channels, int dim, nhmc_stream_t stream);
/* nhmc_data_cs on xt_next (the clipped decode of the LAST DDIM step) with that step's VJP applied in the last row
 * pass: writes g_xt and channels [0, channels) of g_e (the caller keeps the sigma-channels zero). */
int nhmc_data_cs_vjp(const float* xt_next, const float* y_spec, const float* xt, const float* e, int e_channels,
                     const float* at, const float* at_next, float* g_xt, float* g_e, double* loss_ws, float* tmp,
                     int n_chains, int channels, int dim, nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a15  Spectral (anisotropic-blur) operator   Hfuncs.py:448-523
 *   out_c = Lo (D_c o (L^T X_c R)) Ro^T      (H: L,R = V1,V2, Lo,Ro = U1,U2; H^T swaps them)
 * as a chain of fp32-MFMA GEMMs against the four shared d x d factor matrices; d % 64 == 0.
 * "Multiply by M from the left" consumes M^T's memory and "from the right" M's memory (k-major
 * tiles for the MFMA fragments), so the host keeps both orientations of the factors resident.
 * nhmc_spectral_apply   : one sandwich; L, R as stored, LoT = Lo^T, RoT = Ro^T as stored.
 *                         tmp: float[n_chains*C*d*d] scratch.
 * nhmc_data_spectral    : r = y - H clip(xt); loss partials (nhmc_spectral_tiles per chain);
 *                         g_xt = -2 H^T r (masked).  factors: packed [8][d][d] =
 *                         U1,U2,V1,V2,U1^T,U2^T,V1^T,V2^T.  tmp: float[2*n_chains*C*d*d] scratch.
 *                         The forward multiplies by the left factor first (Hfuncs.py:493-509), the adjoint by the
 *                         RIGHT factor first -- the order autograd differentiates those matmuls in at
 *                         main_sampling.py:695,711 -- and every product is an exact k-ascending FMA chain, as
 *                         torch's CPU sgemm is: the data term is the reference's bits.  The right-first order runs on
 *                         transposed operands, hence two constant inputs in transposed form:
 *                         yT = the observation with every channel plane transposed ([n_chains][C][d][d]),
 *                         DmapT = Dmap with every channel plane transposed.
 * ---------------------------------------------------------------------------------- */
int nhmc_spectral_apply(const float* x, const float* L, const float* R, const float* Dmap,
                        const float* LoT, const float* RoT, float* out, float* tmp,
                        int n_chains, int channels, int dim, nhmc_stream_t stream);
int nhmc_spectral_tiles(int channels, int dim);
int nhmc_data_spectral(const float* xt, const float* yT, const float* factors, const float* Dmap,
                       const float* DmapT, int apply_clip, float* g_xt, double* loss_ws, float* tmp,
                       int n_chains, int channels, int dim, nhmc_stream_t stream);
/* a11 + a12/a15 fused: data term on xt_next (the clipped decode of the LAST DDIM step, as nhmc_ddim_mix_fwd wrote it)
 * with the step's VJP applied in the last product's epilogue: writes g_xt and channels [0, channels) of g_e (the caller
 * keeps its sigma-channels zero).  Replaces nhmc_data_spectral(apply_clip = 0) + nhmc_ddim_mix_bwd(final_clip = 1). */
int nhmc_data_spectral_vjp(const float* xt_next, const float* yT, const float* factors, const float* Dmap,
                           const float* DmapT, const float* xt, const float* e, int e_channels, const float* at,
                           const float* at_next,
                           float* g_xt, float* g_e, double* loss_ws, float* tmp, int n_chains, int channels, int dim,
                           nhmc_stream_t stream);

/* The same data term with the residual taken in the operator's left singular basis (U1, U2 orthogonal: full SVDs,
 * Hfuncs.py:473-474):  y - H x = U1 (y^ - D o (V1^T x V2)) U2^T  with  y^ = U1^T y U2, so
 *   |y - H x|^2 = |y^ - D o S|^2   and   H^T (y - H x) = V1 (D o (y^ - D o S)) V2^T      (S = V1^T x V2):
 * four d^3 products per evaluation instead of eight.  y^ is constant over a run (main_sampling.py:694-695,710-711 use
 * the same y_0 in every leapfrog step): compute it once with nhmc_spectral_project.
 * nhmc_spectral_project      : out = L^T Y R per channel image (L = U1, R = U2 as stored); tmp: float[n_chains*C*d*d].
 * nhmc_data_spectral_proj    : as nhmc_data_spectral with y_proj in place of y; tmp: float[n_chains*C*d*d].
 * nhmc_data_spectral_proj_vjp: as nhmc_data_spectral_vjp with y_proj in place of y; same tmp. */
int nhmc_spectral_project(const float* y, const float* L, const float* R, float* out, float* tmp, int n_chains,
                          int channels, int dim, nhmc_stream_t stream);
int nhmc_data_spectral_proj(const float* xt, const float* y_proj, const float* factors, const float* Dmap,
                            int apply_clip, float* g_xt, double* loss_ws, float* tmp, int n_chains, int channels,
                            int dim, nhmc_stream_t stream);
int nhmc_data_spectral_proj_vjp(const float* xt_next, const float* y_proj, const float* factors, const float* Dmap,
                                const float* xt, const float* e, int e_channels, const float* at,
                                const float* at_next, float* g_xt, float* g_e, double* loss_ws, float* tmp,
                                int n_chains, int channels, int dim, nhmc_stream_t stream);

/* Separable strided convolution (SRConv / sr_bicubic, Hfuncs.py:527-607), in the reference's stage order: with
 * V1 = V_small[:, :sd] ([d][sd]), U = U_small ([sd][sd]), S[i][j] = fl(s_i * s_j) ([sd][sd], thresholded singular values):
 *   H(X) = U (S o (V1^T X V1)) U^T,  H^T(Y) = V1 (S o (U^T Y U)) V1^T,  H^+(Y) = V1 (S+ o (U^T Y U)) V1^T  (S+ = 1/S where != 0)
 * every product and the multiplication by S a rounded fp32 stage (H at Hfuncs.py:65-71).  Same MFMA kernel, rectangular.
 * nhmc_sandwich_rect: out = (S1^T in S2) o mul per image, as t = in^T S1, out = (t^T S2) o mul, with in [K1][R1],
 *   S1 [K1][C1], S2 [R1][C2], mul [C1][C2] (nullable: no multiplication), out [C1][C2] (all dims % 32 == 0);
 *   tmp: float[n_img*R1*C1].   V1^T X V1: in = X, S1 = S2 = V1, mul = S;   U Z U^T: in = Z, S1 = S2 = U^T.
 * nhmc_data_srconv: r = y - H(clip(xt)), loss partials (nhmc_srconv_tiles per chain), g = -2 H^T r (masked): eight
 *   products, the adjoint right factor first as in nhmc_data_spectral (autograd's order), so `y` is the observation
 *   with every channel plane TRANSPOSED ([n_chains][C][sd][sd]).  V1T = V1^T as stored [sd][d], UT = U^T [sd][sd];
 *   tmp: float[n_chains*C*(d*sd + 3*sd*sd)]. */
int nhmc_sandwich_rect(const float* in, const float* S1, const float* S2, const float* mul, float* out, float* tmp,
                       int n_img, int K1, int R1, int C1, int C2, nhmc_stream_t stream);
int nhmc_srconv_tiles(int channels, int small_dim);
int nhmc_data_srconv(const float* xt, const float* y, const float* V1, const float* V1T, const float* U, const float* UT,
                     const float* S, int apply_clip, float* g_xt, double* loss_ws, float* tmp, int n_chains,
                     int channels, int dim, int small_dim, nhmc_stream_t stream);
/* nhmc_data_srconv on xt_next (the clipped decode of the LAST DDIM step) with that step's VJP applied in the final
 * product's epilogue (as nhmc_data_spectral_vjp): writes g_xt and channels [0, channels) of g_e. */
int nhmc_data_srconv_vjp(const float* xt_next, const float* y, const float* V1, const float* V1T, const float* U,
                         const float* UT, const float* S, const float* xt, const float* e, int e_channels,
                         const float* at, const float* at_next, float* g_xt, float* g_e, double* loss_ws, float* tmp,
                         int n_chains, int channels, int dim, int small_dim, nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a5  Hamiltonian                          main_sampling.py:697,717-718
 *   H = (0.5*Sx + (1/(2 sigma_y^2))*loss) + (0.5*Sp)*m^-1, per chain, in the reference's fp32
 *   op order on fp32-rounded sums; Sx,Sp summed (fixed order, fp64) from the leapfrog partials.
 * terms (nullable): double[n_chains][3] = Sx, Sp, loss as summed.
 * ---------------------------------------------------------------------------------- */
int nhmc_hamiltonian(const double* sums_ws, int tiles, const double* loss, const double* sigma_y,
                     double m_inv, float* H_out, double* terms, int n_chains,
                     nhmc_stream_t stream);
/* the same with the loss taken from the gradient cache: loss_pair[sel[chain]*loss_stride + chain] */
int nhmc_hamiltonian_cached(const double* sums_ws, int tiles, const double* loss_pair, const int32_t* sel,
                            int64_t loss_stride, const double* sigma_y, double m_inv, float* H_out, double* terms,
                            int n_chains, nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a6  Metropolis test, per chain, on the device (no host sync)   main_sampling.py:718-720
 *   dH = H1 - H0;  accept = active && (u < min(1, exp(-dH)))
 * ---------------------------------------------------------------------------------- */
int nhmc_metropolis(const float* H0, const float* H1, const float* u, const int32_t* active,
                    int32_t* accept, float* dH, int n_chains, nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a7  Accept/reject bookkeeping and schedules   main_sampling.py:683-689,721-749
 * nhmc_schedule_begin: per chain, active = epoch < epochs+2*sampling; sigma_y(epoch) anneal;
 *   at epoch == epochs: sigma_y = sigma_0 and (tau > 0.1 -> tau = 0.1, eps = 0.01).
 *   Inactive chains get eps_eff = 0 (frozen), active ones eps_eff = eps.
 * nhmc_accept_commit: accepted chains copy x_prop -> x and, when epoch >= epochs+sampling,
 *   xt_prop -> samples[chain][epoch-(epochs+sampling)] (samples: [n_chains][sampling][n_elem]).
 *   Uses the PRE-increment epoch (:724-727); call before nhmc_schedule_end.
 * nhmc_schedule_end: accept -> epoch += 1, rejected = 0; reject -> rejected += 1 and, from the
 *   second consecutive reject on, tau *= 0.95, eps *= 0.95.
 * ---------------------------------------------------------------------------------- */
int nhmc_schedule_begin(const int32_t* epoch, double* tau, double* eps, double* sigma_y,
                        double* eps_eff, int32_t* active, double sigma_0, int epochs, int sampling,
                        int n_chains, nhmc_stream_t stream);
int nhmc_accept_commit(const int32_t* accept, const int32_t* epoch, float* x, const float* x_prop,
                       const float* xt_prop, float* samples, int epochs, int sampling,
                       int n_chains, int64_t n_elem, nhmc_stream_t stream);
int nhmc_schedule_end(const int32_t* accept, const int32_t* active, int32_t* epoch,
                      int32_t* rejected, double* tau, double* eps, int32_t* n_accept,
                      int32_t* n_reject, int n_chains, nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Latent variant (hmc_latent)                   main_sampling_latent.py:691-733
 * The epoch index is the shared loop counter there (a reject consumes its epoch), so the host passes
 * what depends on it: final_phase = (epoch >= epochs) and sigma_y_on_accept =
 * sigma_y0*(sigma_0/sigma_y0)**(epoch/epochs) (annealing) or sigma_0 (final phase).
 * nhmc_latent_commit: accepted chains push their PREVIOUS accepted decode xt_last into a ring of the last
 *   `keep` samples when final_phase && has_prev (samples: [n_chains][keep][n_elem], slot count % keep;
 *   :709 runs before :713), then xt_last <- xt_prop, x <- x_prop.  Call before nhmc_schedule_end_latent.
 * nhmc_schedule_end_latent: accept -> rejected = 0, sigma_y = sigma_y_on_accept, final phase also tau = 0.1,
 *   eps = 0.01 and count += has_prev; has_prev = 1.  reject -> rejected += 1; at 2: tau *= 0.9, eps *= 0.9,
 *   rejected = 0.
 * ---------------------------------------------------------------------------------- */
int nhmc_latent_commit(const int32_t* accept, const int32_t* has_prev, const int32_t* count,
                       int final_phase, int keep, float* x, const float* x_prop, float* xt_last,
                       const float* xt_prop, float* samples, int n_chains, int64_t n_elem,
                       nhmc_stream_t stream);
int nhmc_schedule_end_latent(const int32_t* accept, int32_t* rejected, double* tau, double* eps,
                             double* sigma_y, int32_t* count, int32_t* has_prev, int32_t* n_accept,
                             double sigma_y_on_accept, int final_phase, int n_chains,
                             nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Diagonal-mass variant (hmc_test_conditioning)     main_sampling.py:776-894
 * nhmc_leapfrog_mass: the fused update with a per-element mass (inv_m = 1/M, std_m = sqrt(M), [n_chains][n_elem]):
 *   FIRST: p = z*std_m (z: raw N(0,1) draw, :819); partials Sx = sum x^2, Sp = sum inv_m p^2 (:824);
 *          p -= (eps/2) G; x += (eps p) inv_m (:834)
 *   MID  : p -= eps G (:841); [Welford]; x += (eps p) inv_m
 *   LAST : p -= eps G; p += (eps/2) G (:849); [Welford]; partials of the outputs (:852)
 *   [Welford] (:843-847), for chains with welford_on: delta = x - mean; mean += delta/(l+1); M2 += delta (x - mean),
 *   l = 0-based leapfrog index; at l = 0 mean and M2 are taken as zero (not read).  H = nhmc_hamiltonian with m_inv = 1.
 * nhmc_mass_from_variance (:857-870): variance = M2/(L-1); ascending sort per chain, ties by index (a stable sort:
 *   the reference's default torch.sort leaves the order of equal variances to the sort implementation); the element
 *   of rank r gets std_m = std_table[r], inv_m = inv_table[r], the host-evaluated sqrt(M_r) and 1/M_r of
 *   M_r = exp(2 r/(N-1) - 1) (float[n_elem] device arrays: the transcendental is the reference host's, :863-868);
 *   written for the chains whose flag is set.  ws: nhmc_mass_sort_ws_bytes(n_chains, n_elem) bytes.
 * nhmc_schedule_begin_mass (:808-816): sigma_table[e], e = 0..epochs, are the host-evaluated sigma_y values;
 *   active = epoch < burn+epochs+4*sampling; welford_on = active && (epoch-burn) > epochs/3.
 * ---------------------------------------------------------------------------------- */
int nhmc_leapfrog_mass(int mode, float* x, float* p, const float* z, const float* g, const float* g2,
                       const float* inv_m, const float* std_m, const double* eps, const double* sigma_y,
                       const int32_t* welford_on, float* mean, float* m2, int l, int n_chains,
                       int64_t n_elem, double* sums_ws, nhmc_stream_t stream);
size_t nhmc_mass_sort_ws_bytes(int n_chains, int64_t n_elem);
int nhmc_mass_from_variance(const float* m2, int L, const int32_t* flags, const float* std_table,
                            const float* inv_table, float* inv_m, float* std_m,
                            void* ws, size_t ws_bytes, int n_chains, int64_t n_elem, nhmc_stream_t stream);
int nhmc_schedule_begin_mass(const int32_t* epoch, double* tau, double* eps, double* sigma_y, double* eps_eff,
                             int32_t* active, int32_t* welford_on, const double* sigma_table, int burn, int epochs,
                             int sampling, int n_chains, nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * f.1  Codebook lookup of the VQ first stage on decode (latent variant)
 *      ldm/models/autoencoder.py:274-279 -> taming VectorQuantizer2.forward (not vendored; taming-transformers==0.0.1)
 *   per latent pixel: k* = argmin_k (|z|^2 + |e_k|^2) - 2 z.e_k (first minimum), z_q = z + (e_k* - z)
 * z, z_q: [n_chains][channels][hw] (channels = embed_dim in {3, 4}); codebook: [n_embed][channels];
 * idx (nullable): int32 [n_chains][hw].  The backward of this step is the identity (straight-through).
 * ---------------------------------------------------------------------------------- */
int nhmc_vq_nearest(const float* z, const float* codebook, float* z_q, int32_t* idx, int n_chains,
                    int channels, int64_t hw, int n_embed, nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a16 glue  Fused GroupNorm (+ FiLM scale/shift) (+ SiLU) of the score networks, forward and input gradient
 *      guided_diffusion/unet_ffhq.py:310-321 (GroupNorm32, scale-shift norm, SiLU); ldm/modules/diffusionmodules/model.py:38-39
 *   u = ((x - mean_g) rstd_g gamma_c + beta_c) (1 + scale_bc) + shift_bc ;   y = act ? u sigmoid(u) : u
 * x, y, dy, dx: [n][channels][hw] contiguous fp32 (hw % 4 == 0); gamma, beta: [channels];
 * film (nullable): [n][film_stride] with scale at [c] and shift at [channels + c] (the reference's emb_out.chunk(2)).
 * pre (nullable): x + pre[b * pre_stride + c] is what gets normalised -- the bias of the convolution that produced x
 *   (pre_stride = 0) or bias + a per-sample embedding term ([n][pre_stride], openaimodel.py:257 `h = h + emb_out`), so the
 *   producer runs without its broadcast add pass.
 * ws: double[n * groups][splits][2] partial sums (splits = nhmc_gn_splits(...)); the forward's ws is an input of the
 * backward (mean / rstd are re-derived from it; nothing else is saved).  Only dx is produced: the networks are frozen.
 * ---------------------------------------------------------------------------------- */
int nhmc_gn_splits(int n, int channels, int groups, int64_t hw);
int nhmc_gn_act_fwd(const float* x, const float* gamma, const float* beta, const float* film, int64_t film_stride,
                    const float* pre, int64_t pre_stride, float eps, int act, float* y, double* ws, int splits,
                    int n, int channels, int groups, int64_t hw, nhmc_stream_t stream);
/* dx_add (optional, same shape as x, must not alias dx): a second gradient of x -- in a ResBlock the block input feeds the
 * GroupNorm AND the skip path -- added to the result in the same pass: dx = fl(dx_groupnorm) + dx_add, the bits of
 * autograd's separate accumulation add. */
int nhmc_gn_act_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* film,
                    int64_t film_stride, const float* pre, int64_t pre_stride, float eps, int act,
                    const double* fwd_ws, const float* dx_add, float* dx, double* ws, int splits, int n, int channels,
                    int groups, int64_t hw, nhmc_stream_t stream);
/* out = (h + bias_c) + other, [n][channels][hw]: a convolution's bias folded into the residual add that follows it
 * (unet_ffhq.py:321).  Its backward is the identity towards both h and other. */
int nhmc_bias_add2(const float* h, const float* bias, const float* other, float* out, int n, int channels,
                   int64_t hw, nhmc_stream_t stream);

/* PSNR of clamp((xt+1)/2,0,1) against clamp((x_orig+1)/2,0,1)   main_sampling.py:738-739
 * ws: double[n_chains][nhmc_data_tiles(n_elem)]. */
int nhmc_psnr(const float* xt, const float* x_orig, float* psnr, double* ws,
              int n_chains, int64_t n_elem, nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a1  Momentum / accept noise: Philox4x32-10 keyed by seed, counted by
 *     (element quad, chain_id0 + chain, draw, tag) -- independent of how chains are sharded.
 *     Replaces torch.randn_like (main_sampling.py:692) and torch.rand(1) (:720).
 * ---------------------------------------------------------------------------------- */
int nhmc_randn_philox(float* out, uint64_t seed, uint32_t chain_id0, uint32_t draw, float scale,
                      int n_chains, int64_t n_elem, nhmc_stream_t stream);
int nhmc_uniform_philox(float* out, uint64_t seed, uint32_t chain_id0, uint32_t draw,
                        int n_chains, nhmc_stream_t stream);

/* ------------------------------------------------------------------------------------
 * (d) Measurement aid: a streaming copy dst = src with the access pattern of the fused update (256-thread blocks,
 *     one non-temporal float4 per thread): the "copy ceiling" bench.py quotes beside the 8 TB/s HBM peak
 *     (SURVEY.md section 8d).  n_elem % 4 == 0, 16-byte aligned.
 * ---------------------------------------------------------------------------------- */
int nhmc_copy_probe(const float* src, float* dst, int64_t n_elem, nhmc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NHMC_H */
