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
// ONE kernel form serves all eight products of a data-term evaluation:
//         OUT = IN^T * S          OUT[r][c] = sum_k IN[k][r] * S[k][c]
// Both operands are k-major in memory (r resp. c contiguous), so global->LDS staging is a straight
// 16-byte copy and every MFMA fragment read is 32 consecutive floats per half-wave (conflict-free
// ds_read_b32), with no transposing store and no padding.  Chaining it four times gives the sandwich:
//     X1 = X^T V1,   X2 = (X1^T V2) o D = (V1^T X V2) o D,   X3 = X2^T U1^T,   X4 = X3^T U2^T = U1 X2 U2^T
// (each product undoes the transpose of the one before).  "S" is V1, V2, U1^T, U2^T for H and
// U1, U2, V1^T, V2^T for H^T: the host keeps both orientations of the four factors resident (2 MB).
//
// Block tile T x T (T = 128: 4 waves as 2x2, each 2x2 MFMA tiles of 32x32; 64: 4 waves, 1 tile each; 32: one
// wave), K step 32, synchronous global->LDS staging (latency covered by the other resident blocks).  Epilogues fuse the spectral multiplier, the residual + per-tile loss partial, and the
// -2 * clip-mask scaling; PRECLIP clamps IN on the fly (the data term's final clip).
#include "nhmc_common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_NONE = 0, EPI_MULD = 1, EPI_RESID = 2, EPI_GRAD = 3, EPI_VJP = 4, EPI_SRES = 5 };
// EPI_SRES: the residual taken in the operator's left singular basis: r^ = y^ - D o acc with y^ = U1^T y U2 (aux), loss
// partial += r^2, out = D o r^ -- the input of the adjoint's last two products (see nhmc_data_spectral_proj).
// EPI_VJP: the gradient tile goes straight through the VJP of the last DDIM step (k_mix_bwd, final clip included)
struct VjpArgs { const float* e; float* g_e; const float* at; const float* at_next; int e_channels; };
constexpr int BK = 32;

// TOUT: the product is STORED TRANSPOSED, OUT_T[c][r] = sum_k IN[k][r] S[k][c], and the epilogue (Dmap / aux / e / g_e
// reads, the store) works in the coordinates of that transposed matrix.  It is what lets a chain of this one kernel form
// multiply from the RIGHT first: autograd differentiates U (D o (V^T x V)) U^T as  r U, then U^T (.), then (.) V^T, then
// V (.) -- right factor first at both sandwiches -- and in the space of transposed operands OUT = IN^T S is a right
// multiplication (IN = M^T gives M S).  So the residual is written transposed by the forward's last product and the
// gradient is transposed back by the adjoint's last product, both inside their epilogues; the accumulator slab is
// XOR-swizzled in LDS so that the column-wise reads of the transposing epilogue stay conflict-free.
template <int T, int NW, int EPI, bool PRECLIP, bool TOUT = false>
__global__ __launch_bounds__(64 * NW * NW, (NW == 2 && EPI != EPI_NONE) ? 4 : 1) void k_sgemm(
    const float* __restrict__ IN, const float* __restrict__ S, float* __restrict__ OUT,
    const float* __restrict__ Dmap, const float* __restrict__ aux, double* __restrict__ ws, int K, int R, int C,
    int channels, VjpArgs vj) {
  constexpr int NT = 64 * NW * NW;
  constexpr int FR = T / (32 * NW);
  constexpr int NV = (BK * T / 4) / NT;          // float4 per thread per operand tile
  __shared__ float L[2 * BK * T];        // operand tiles in the main loop, one accumulator row-slab in the epilogue
  float* const As = L;
  float* const Bs = L + BK * T;

  // 1-D grid, XCD-aware: hardware block id L runs on XCD L % 8, so the blocks of one XCD are made consecutive in
  // the logical order [image][tile row][tile col] -- the tiles of an image (which share its k-slabs pairwise, and all
  // share S) then fill the same L2 instead of four different ones.
  const int tiles_c = C / T, tiles_img = (R / T) * tiles_c;
  const int total = gridDim.x;
  int logical = blockIdx.x;
  if ((total & 7) == 0) logical = (logical & 7) * (total >> 3) + (logical >> 3);
  const int img = logical / tiles_img, tile = logical % tiles_img;
  const int tr = (tile / tiles_c) * T, tc = (tile % tiles_c) * T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave / NW, wj = wave % NW;
  const int lr = lane & 31, lh = lane >> 5;
  const float* __restrict__ Ximg = IN + (int64_t)img * K * R;       // IN: [K][R], S: [K][C], OUT: [R][C]

  f32x16 acc[FR][FR];
#pragma unroll
  for (int a = 0; a < FR; ++a)
#pragma unroll
    for (int b = 0; b < FR; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // Synchronous staging: with 3-4 blocks resident per CU the other blocks' MFMAs cover the global-load
  // latency; register prefetch and LDS double buffering measured 5-10 % SLOWER here (tools/gemm_bench.hip:
  // 57 us vs 61-69 us per 192-image product), because they cost registers and a second barrier.
  for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = tid + v * NT, kk = idx / (T / 4), c4 = idx % (T / 4);
      nhmc_v4f pa = *reinterpret_cast<const nhmc_v4f*>(&Ximg[(int64_t)(k0 + kk) * R + tr + c4 * 4]);
      const nhmc_v4f pb = *reinterpret_cast<const nhmc_v4f*>(&S[(int64_t)(k0 + kk) * C + tc + c4 * 4]);
      if (PRECLIP) {
        pa.x = nhmc_clip1(pa.x); pa.y = nhmc_clip1(pa.y); pa.z = nhmc_clip1(pa.z); pa.w = nhmc_clip1(pa.w);
      }
      *reinterpret_cast<nhmc_v4f*>(&As[kk * T + c4 * 4]) = pa;
      *reinterpret_cast<nhmc_v4f*>(&Bs[kk * T + c4 * 4]) = pb;
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < BK; kk += 2) {
      float a[FR], b[FR];
#pragma unroll
      for (int f = 0; f < FR; ++f) {
        a[f] = As[(kk + lh) * T + (wi * FR + f) * 32 + lr];
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

  // ---- epilogue ----
  // C/D map of the 32x32 MFMA tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  The accumulators go through
  // LDS one row-slab (T/NW rows x T columns <= the 2*BK*T floats of the operand tiles) at a time, so that every
  // global access of the epilogue -- the store and the Dmap / y / xt / e reads of the fused variants -- is a 16-byte
  // row access (the direct per-lane 4-byte form cost 45 us in the VJP variant, as much as the kernel it replaced).
  static_assert((T / NW) * T <= 2 * BK * T, "accumulator slab must fit the operand tiles");
  const int c = img % channels;
  float* __restrict__ out_img = OUT + (int64_t)img * R * C;
  const float* __restrict__ dm_img = (EPI == EPI_MULD || EPI == EPI_SRES) ? Dmap + (int64_t)c * R * C : nullptr;
  const float* __restrict__ aux_img = aux ? aux + (int64_t)img * R * C : nullptr;
  float lsum = 0.0f;
  constexpr int SR = T / NW;                            // rows of one accumulator slab
  // EPI_VJP: aux = the DDIM step's input xt; e / g_e are [chain][e_channels][R][C]
  const float* __restrict__ e_img = nullptr;
  float* __restrict__ ge_img = nullptr;
  float c1 = 0.f, c2 = 1.f, c3 = 0.f, c4 = 0.f;
  if (EPI == EPI_VJP) {
    const int chain = img / channels;
    const int64_t eoff = ((int64_t)chain * vj.e_channels + c) * R * C;
    e_img = vj.e + eoff;
    ge_img = vj.g_e + eoff;
    const float a = vj.at[chain], an = vj.at_next[chain];
    c1 = sqrtf(1.0f - a); c2 = sqrtf(a); c3 = sqrtf(an); c4 = sqrtf(1.0f - an);
  }
  constexpr int SLAB_V = ((T / NW) * T / 4) / NT;       // float4 per thread per slab
  for (int h = 0; h < NW; ++h) {
    if (wi == h) {
#pragma unroll
      for (int fa = 0; fa < FR; ++fa)
#pragma unroll
        for (int fb = 0; fb < FR; ++fb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = fa * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = (wj * FR + fb) * 32 + lr;
            L[row * T + (TOUT ? (col ^ ((row >> 2) & 31)) : col)] = acc[fa][fb][r];
          }
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < SLAB_V; ++v) {
      const int idx = tid + v * NT;
      int off;
      nhmc_v4f q;
      if (TOUT) {                                       // 4 consecutive ROWS of one column = 4 consecutive floats of OUT_T
        const int r4 = idx % (SR / 4), col = idx / (SR / 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = L[(r4 * 4 + j) * T + (col ^ (r4 & 31))];
        off = (tc + col) * R + tr + h * SR + r4 * 4;
      } else {
        const int row = idx / (T / 4), c4i = idx % (T / 4);
        off = (tr + h * SR + row) * C + tc + c4i * 4;
        q = *reinterpret_cast<const nhmc_v4f*>(&L[row * T + c4i * 4]);
      }
      if (EPI == EPI_MULD) {
        const nhmc_v4f d = *reinterpret_cast<const nhmc_v4f*>(&dm_img[off]);
        q = q * d;
      }
      if (EPI == EPI_RESID) {
        const nhmc_v4f yv = *reinterpret_cast<const nhmc_v4f*>(&aux_img[off]);
        q = yv - q;                                       // r = y - H x
        lsum += q.x * q.x; lsum += q.y * q.y; lsum += q.z * q.z; lsum += q.w * q.w;
      }
      if (EPI == EPI_SRES) {
        const nhmc_v4f d = *reinterpret_cast<const nhmc_v4f*>(&dm_img[off]);
        const nhmc_v4f yv = *reinterpret_cast<const nhmc_v4f*>(&aux_img[off]);
        q = yv - q * d;                                   // r^ = U1^T (y - H x) U2
        lsum += q.x * q.x; lsum += q.y * q.y; lsum += q.z * q.z; lsum += q.w * q.w;
        q = q * d;
      }
      if (EPI == EPI_GRAD) {
        q = -(2.0f * q);
        if (aux_img) {
          const nhmc_v4f xv = *reinterpret_cast<const nhmc_v4f*>(&aux_img[off]);
          q.x = q.x * nhmc_in1(xv.x); q.y = q.y * nhmc_in1(xv.y); q.z = q.z * nhmc_in1(xv.z); q.w = q.w * nhmc_in1(xv.w);
        }
      }
      if (EPI == EPI_VJP) {                               // same op order as k_mix_bwd<false,false>, final_clip = 1
        const nhmc_v4f xv = *reinterpret_cast<const nhmc_v4f*>(&aux_img[off]);
        const nhmc_v4f ev = *reinterpret_cast<const nhmc_v4f*>(&e_img[off]);
        nhmc_v4f ge;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          const float ee = ev[k4];
          const float u = (xv[k4] - ee * c1) / c2;
          float gin = -(2.0f * q[k4]);
          gin = gin * nhmc_in1(c3 * nhmc_clip1(u) + c4 * ee);
          const float gu = ((gin * c3) * nhmc_in1(u)) / c2;
          ge[k4] = c4 * gin + (-gu) * c1;
          q[k4] = gu;
        }
        *reinterpret_cast<nhmc_v4f*>(&ge_img[off]) = ge;
      }
      *reinterpret_cast<nhmc_v4f*>(&out_img[off]) = q;
    }
    __syncthreads();                                      // the slab is rewritten by the next row of waves
  }
  if (EPI == EPI_RESID || EPI == EPI_SRES) {
    __shared__ double red[NW * NW];
    double sw = nhmc_wave_sum((double)lsum);
    if (lane == 0) red[wave] = sw;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
      for (int w = 0; w < NW * NW; ++w) tot += red[w];
      // tiles of one chain are contiguous: [chain][channel][tile_row][tile_col]
      ws[(int64_t)img * tiles_img + tile] = tot;
    }
  }
}

// ---- two products per launch (d = 256): OUT = (IN^T S1)^T S2, the intermediate never leaves the CU --------------------
// Round 2.  The chain above moves every intermediate through HBM (8 launches, ~16T of intermediates per chain against 3T
// algorithmic) and, because all 768 blocks of a product are co-resident and in lock-step, its store epilogues and first
// loads are never overlapped with another block's MFMAs.  Here one block owns a 64-column slab of the intermediate
//     T1[:, slab] = IN^T S1[:, slab]      (256 x 64 fp32 = 64 KB, kept in LDS)
// and immediately consumes it as the k-major left operand of the second product
//     OUT[slab rows][:] = T1[:, slab]^T S2 (64 x 256),
// so a pair of products costs one launch, one prologue and one epilogue, and T1 is never written.  4 blocks per image
// (768 per pair at B = 64 = 3 per CU; 64 KB of LDS each, so two are resident and drift apart: one block's slab staging
// and epilogue overlap the other's MFMAs), 8 waves per block.
// The main loops have NO block barrier: the operand every wave shares is resident in LDS for the whole phase (the S1
// slab in phase 1, staged once; the T1 slab in phase 2), and the operand a wave needs alone (its 32 rows of IN, its 32
// columns of S2) is read straight from global / L2 into MFMA fragments, NHMC_PAIR_CHK (8) k-pairs ahead in registers, two sets, so waves drift
// freely and the two waves of a SIMD fill each other's latencies.  MFMA count, k order and epilogues are those of the
// two-launch form: the same bits.  Measured at B = 64 (data term + last VJP, tools/pair_bench.py): one product per
// launch 525 us; this form 522 us (98.7 TFLOP/s); with per-K-step LDS staging of both operands + a block barrier 562 us
// (8 waves) / 643 us (4 waves); with the S1 slab read from global instead of LDS (64 KB, two blocks per CU) 676 us.
// So the pair form does not beat the chain on time -- each block still serialises ~16 us of slab staging, barriers and
// the 64 KB epilogue against 27 us of MFMA work, and one block fits a CU -- but it halves the launches and keeps
// half of the intermediates (8T of 16T per chain) out of HBM.
// NHMC_PAIR_STAMPS (tools/pair_stamps.hip only; never defined in the product build): thread 0 of every workgroup records
// clock64() at the phase boundaries, for the per-phase attribution in profiles/.
#ifdef NHMC_PAIR_STAMPS
__device__ long long* nhmc_pair_stamps = nullptr;               // [workgroup][8]
#define NHMC_STAMP(i) do { if (threadIdx.x == 0 && nhmc_pair_stamps) nhmc_pair_stamps[(long long)blockIdx.x * 8 + (i)] = clock64(); } while (0)
#else
// Without the stamps every phase boundary still gets a basic-block split: a branch on a condition the compiler cannot
// evaluate (channels is never INT_MAX: n_chains * channels <= 65535 is checked on the host) around a store it cannot
// drop.  The machine scheduler works per basic block; with the whole kernel in one block it hoists the next phase's global
// loads over the phase boundaries and allocates 156-179 VGPRs (25-48 spilled once the launch bounds ask for 128); split
// where the stamps split it, the same code takes 103-124 and two blocks are resident per CU.  One scalar compare per boundary.
#define NHMC_STAMP(i) do { if (channels == 0x7fffffff) OUT[0] = (float)clock64(); } while (0)
#endif

#ifndef NHMC_LDS_AHEAD
#define NHMC_LDS_AHEAD 2      // k-pairs by which the LDS operand read leads its MFMAs
#endif
#ifndef NHMC_PAIR_CHK
#define NHMC_PAIR_CHK 8       // k-pairs per register set of the global operand ring (measured with two blocks resident: 4 / 8 / 16 / 32 -> 458.9 / 458.8 / 465.0 / 478.3 us per data term + last VJP)
#endif
// TOUT: as in k_sgemm -- the pair's result is stored transposed (and its epilogue runs in the transposed coordinates).
template <int EPI, bool PRECLIP, int NW, bool TOUT = false>
// __launch_bounds__' second argument is waves per SIMD (not blocks per CU): two resident blocks of NW waves on 4 SIMDs are
// NW / 2 waves per SIMD, i.e. at most 128 VGPRs for NW = 8.  Written as "2" (rounds 2 and 3 until the phase-stamp harness,
// whose stamps happened to hold the allocation at 104, ran 13 % faster than the product build of the same source) the
// compiler took 156-179 registers and ONE block fitted a CU.
__global__ __launch_bounds__(64 * NW, NW / 2) void k_pair256(
    const float* __restrict__ IN, const float* __restrict__ S1, const float* __restrict__ S2, float* __restrict__ OUT,
    const float* __restrict__ Dmap, const float* __restrict__ aux, double* __restrict__ ws, int channels, VjpArgs vj) {
  constexpr int D = 256, SL = 64, NT = 64 * NW, CHK = NHMC_PAIR_CHK;   // CHK: k-pairs fetched ahead per register set
  constexpr int F = 8 / NW;                                  // 32-wide fragments of the wave's private operand (NW = 4: 2 x 2 MFMA tiles)
  extern __shared__ float lds[];
  // ONE 64 KB region, three lives: S1[:, slab] during phase 1, T1[:, slab] during phase 2 (it lives in the accumulators
  // until every wave has finished reading S1), the [64][256] output staging of the epilogue -> two blocks fit a CU
  float* const slab = lds;                                  // [256][64]  T1[k'][r'] / [64][256] epilogue staging
  float* const s1s = lds;                                   // [256][64]  S1[k][slab columns]
  const int total = gridDim.x;
  int logical = blockIdx.x;
  if ((total & 7) == 0) logical = (logical & 7) * (total >> 3) + (logical >> 3);   // the 4 slabs of an image share an XCD
  const int img = logical >> 2, q = logical & 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const float* __restrict__ Ximg = IN + (int64_t)img * D * D;
#ifdef NHMC_PAIR_DELAY                                       // experiment (tools/pair_stamps.hip only): start the second resident workgroup of a CU late
  if (blockIdx.x >= 256 && blockIdx.x < 512)
    for (int i = 0; i < NHMC_PAIR_DELAY; ++i) __builtin_amdgcn_s_sleep(127);
#endif
  NHMC_STAMP(0);
#ifdef NHMC_PAIR_STAMPS
  if (threadIdx.x == 0 && nhmc_pair_stamps) nhmc_pair_stamps[(long long)blockIdx.x * 8 + 7] = wall_clock64();   // constant-rate, chip-wide
#endif

#pragma unroll
  for (int v = 0; v < D * SL / 4 / NT; ++v) {               // S1[:, slab] -> LDS, once
    const int idx = tid + v * NT, kk = idx / (SL / 4), c4 = idx % (SL / 4);
    *reinterpret_cast<nhmc_v4f*>(&s1s[kk * SL + c4 * 4]) = *reinterpret_cast<const nhmc_v4f*>(&S1[(int64_t)kk * D + q * SL + c4 * 4]);
  }
  f32x16 acc[F][2];                                         // phase 1: [private row fragment][slab column fragment]; phase 2: [private column fragment][slab row fragment]
#pragma unroll
  for (int f = 0; f < F; ++f)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][b][r] = 0.0f;

  // ------------------------------------------------ phase 1: T1[wave rows][slab] = IN^T S1[:, slab] ------------------
  {
    const float* __restrict__ ap = Ximg + wave * (32 * F) + lr + (int64_t)lh * D;  // a(k-pair j, f) = ap[2 j D + 32 f]: IN[2j + lh][row]
    float f0[CHK][F], f1[CHK][F];
    auto fetch = [&](float (&fa)[CHK][F], int c) {
#pragma unroll
      for (int j = 0; j < CHK; ++j)
#pragma unroll
        for (int f = 0; f < F; ++f) fa[j][f] = ap[(int64_t)(2 * (c + j)) * D + 32 * f];
    };
    // The LDS operand of k-pair j + LDS_AHEAD is read BEFORE the MFMAs of k-pair j are issued.  Written naively (read b, use
    // b) the compiler emits ds_read2 -> s_waitcnt lgkmcnt(0) -> 2 x v_mfma per k-pair, so a wave's matrix pipe idles for the
    // LDS latency in front of every pair of MFMAs: phase stamps (tools/pair_stamps.hip, r3) showed a workgroup running
    // alone at 39 % of the pipe rate in this loop and two co-resident ones at 62 %.
    constexpr int LDS_AHEAD = NHMC_LDS_AHEAD;
    auto compute = [&](const float (&fa)[CHK][F], int c) {
      float b0[CHK + LDS_AHEAD], b1[CHK + LDS_AHEAD];
#pragma unroll
      for (int j = 0; j < LDS_AHEAD; ++j) {
        const int kk = 2 * (c + j);
        b0[j] = s1s[(kk + lh) * SL + lr];
        b1[j] = s1s[(kk + lh) * SL + 32 + lr];
      }
#pragma unroll
      for (int j = 0; j < CHK; ++j) {
        if (j + LDS_AHEAD < CHK) {
          const int kk = 2 * (c + j + LDS_AHEAD);
          b0[j + LDS_AHEAD] = s1s[(kk + lh) * SL + lr];
          b1[j + LDS_AHEAD] = s1s[(kk + lh) * SL + 32 + lr];
        }
        __builtin_amdgcn_sched_barrier(0);                  // keep that read in front of this k-pair's MFMAs (the scheduler sinks it to its use otherwise)
#pragma unroll
        for (int f = 0; f < F; ++f) {
          const float a = PRECLIP ? nhmc_clip1(fa[j][f]) : fa[j][f];
          acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0[j], acc[f][0], 0, 0, 0);
          acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1[j], acc[f][1], 0, 0, 0);
        }
      }
    };
    fetch(f0, 0);
    __syncthreads();                                        // S1 slab resident
    NHMC_STAMP(1);
    for (int c0 = 0; c0 < D / 2; c0 += 2 * CHK) {
      fetch(f1, c0 + CHK);
      compute(f0, c0);
      if (c0 + 2 * CHK < D / 2) fetch(f0, c0 + 2 * CHK);
      compute(f1, c0 + CHK);
    }
  }
  // accumulators -> slab[k' = row of T1][r' = slab column]   (C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 lh)
  NHMC_STAMP(2);
  __syncthreads();                                          // every wave is done with the S1 slab the T1 slab overwrites
#pragma unroll
  for (int f = 0; f < F; ++f)
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        slab[((wave * F + f) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * SL + fb * 32 + lr] = acc[f][fb][r];
        acc[f][fb][r] = 0.0f;
      }

  // ------------------------------------------------ phase 2: OUT[slab][wave columns] = T1[:, slab]^T S2 --------------
  {
    const float* __restrict__ bp = S2 + wave * (32 * F) + lr + (int64_t)lh * D;    // b(k-pair j, f) = bp[2 j D + 32 f]: S2[2j + lh][col]
    float f0[CHK][F], f1[CHK][F];
    auto fetch = [&](float (&fb)[CHK][F], int c) {
#pragma unroll
      for (int j = 0; j < CHK; ++j)
#pragma unroll
        for (int f = 0; f < F; ++f) fb[j][f] = bp[(int64_t)(2 * (c + j)) * D + 32 * f];
    };
    constexpr int LDS_AHEAD = NHMC_LDS_AHEAD;                            // as in phase 1: the T1 slab operand is read two k-pairs ahead
    auto compute = [&](const float (&fb)[CHK][F], int c) {
      float a0[CHK + LDS_AHEAD], a1[CHK + LDS_AHEAD];
#pragma unroll
      for (int j = 0; j < LDS_AHEAD; ++j) {
        const int kk = 2 * (c + j);
        a0[j] = slab[(kk + lh) * SL + lr];
        a1[j] = slab[(kk + lh) * SL + 32 + lr];
      }
#pragma unroll
      for (int j = 0; j < CHK; ++j) {
        if (j + LDS_AHEAD < CHK) {
          const int kk = 2 * (c + j + LDS_AHEAD);
          a0[j + LDS_AHEAD] = slab[(kk + lh) * SL + lr];
          a1[j + LDS_AHEAD] = slab[(kk + lh) * SL + 32 + lr];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < F; ++f) {
          acc[f][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], fb[j][f], acc[f][0], 0, 0, 0);
          acc[f][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], fb[j][f], acc[f][1], 0, 0, 0);
        }
      }
    };
    fetch(f0, 0);
    __syncthreads();                                        // T1 slab complete
    NHMC_STAMP(3);
    for (int c0 = 0; c0 < D / 2; c0 += 2 * CHK) {
      fetch(f1, c0 + CHK);
      compute(f0, c0);
      if (c0 + 2 * CHK < D / 2) fetch(f0, c0 + 2 * CHK);
      compute(f1, c0 + CHK);
    }
  }

  // ------------------------------------------------ epilogue: [64 rows][256 cols] through the slab's 64 KB -----------
  NHMC_STAMP(4);
  __syncthreads();                                          // every wave is done reading the T1 slab
#pragma unroll
  for (int f = 0; f < F; ++f)
#pragma unroll
    for (int fa = 0; fa < 2; ++fa)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = fa * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = (wave * F + f) * 32 + lr;
        slab[row * D + (TOUT ? (col ^ (row >> 2)) : col)] = acc[f][fa][r];       // 64 rows: row >> 2 < 16, stays in the 32-group
      }
  const int c = img % channels;
  float* __restrict__ out_img = OUT + (int64_t)img * D * D;
  const float* __restrict__ dm_img = (EPI == EPI_MULD || EPI == EPI_SRES) ? Dmap + (int64_t)c * D * D : nullptr;
  const float* __restrict__ aux_img = aux ? aux + (int64_t)img * D * D : nullptr;
  const float* __restrict__ e_img = nullptr;
  float* __restrict__ ge_img = nullptr;
  float c1 = 0.f, c2 = 1.f, c3 = 0.f, c4 = 0.f;
  if (EPI == EPI_VJP) {
    const int chain = img / channels;
    const int64_t eoff = ((int64_t)chain * vj.e_channels + c) * D * D;
    e_img = vj.e + eoff;
    ge_img = vj.g_e + eoff;
    const float a = vj.at[chain], an = vj.at_next[chain];
    c1 = sqrtf(1.0f - a); c2 = sqrtf(a); c3 = sqrtf(an); c4 = sqrtf(1.0f - an);
  }
  // the epilogue's global operands (multiplier map / observation / clip-mask source / the VJP's xt and e) are requested HERE,
  // before the barrier that ends the staging: the accumulators and the operand ring are dead, so their registers hold the
  // 8 (16 for the VJP) float4 per thread, and the loads' latency runs under the barrier instead of inside the store loop
  constexpr int NV = SL * D / 4 / NT;
  constexpr bool USE_DM = EPI == EPI_MULD || EPI == EPI_SRES;
  constexpr bool USE_AUX = EPI == EPI_RESID || EPI == EPI_SRES || EPI == EPI_GRAD || EPI == EPI_VJP;
  auto offset_of = [&](int v) {
    const int idx = tid + v * NT;
    if (TOUT) { const int r4 = idx % (SL / 4), col = idx / (SL / 4); return col * D + q * SL + r4 * 4; }
    const int row = idx / (D / 4), c4i = idx % (D / 4);
    return (q * SL + row) * D + c4i * 4;
  };
  nhmc_v4f pre_dm[USE_DM ? NV : 1], pre_aux[USE_AUX ? NV : 1], pre_e[EPI == EPI_VJP ? NV : 1];
  const bool have_aux = USE_AUX && aux_img != nullptr;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int off = offset_of(v);
    if (USE_DM) pre_dm[v] = *reinterpret_cast<const nhmc_v4f*>(&dm_img[off]);
    if (USE_AUX) { if (have_aux) pre_aux[v] = *reinterpret_cast<const nhmc_v4f*>(&aux_img[off]); }
    if (EPI == EPI_VJP) pre_e[v] = *reinterpret_cast<const nhmc_v4f*>(&e_img[off]);
  }
  __syncthreads();
  NHMC_STAMP(5);
  float lsum = 0.0f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int idx = tid + v * NT;
    const int off = offset_of(v);
    nhmc_v4f o;
    if (TOUT) {                                             // 4 consecutive rows of one column = 4 consecutive floats of OUT_T
      const int r4 = idx % (SL / 4), col = idx / (SL / 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = slab[(r4 * 4 + j) * D + (col ^ r4)];
    } else {
      const int row = idx / (D / 4), c4i = idx % (D / 4);
      o = *reinterpret_cast<const nhmc_v4f*>(&slab[row * D + c4i * 4]);
    }
    if (EPI == EPI_MULD) o = o * pre_dm[v];
    if (EPI == EPI_RESID) {
      o = pre_aux[v] - o;                                   // r = y - H x
      lsum += o.x * o.x; lsum += o.y * o.y; lsum += o.z * o.z; lsum += o.w * o.w;
    }
    if (EPI == EPI_SRES) {
      const nhmc_v4f d = pre_dm[v];
      o = pre_aux[v] - o * d;                               // r^ = U1^T (y - H x) U2
      lsum += o.x * o.x; lsum += o.y * o.y; lsum += o.z * o.z; lsum += o.w * o.w;
      o = o * d;
    }
    if (EPI == EPI_GRAD) {
      o = -(2.0f * o);
      if (have_aux) {
        const nhmc_v4f xv = pre_aux[v];
        o.x = o.x * nhmc_in1(xv.x); o.y = o.y * nhmc_in1(xv.y); o.z = o.z * nhmc_in1(xv.z); o.w = o.w * nhmc_in1(xv.w);
      }
    }
    if (EPI == EPI_VJP) {                                   // same op order as k_mix_bwd<false,false>, final_clip = 1
      const nhmc_v4f xv = pre_aux[v];
      const nhmc_v4f ev = pre_e[v];
      nhmc_v4f ge;
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const float ee = ev[k4];
        const float u = (xv[k4] - ee * c1) / c2;
        float gin = -(2.0f * o[k4]);
        gin = gin * nhmc_in1(c3 * nhmc_clip1(u) + c4 * ee);
        const float gu = ((gin * c3) * nhmc_in1(u)) / c2;
        ge[k4] = c4 * gin + (-gu) * c1;
        o[k4] = gu;
      }
      *reinterpret_cast<nhmc_v4f*>(&ge_img[off]) = ge;
    }
    __builtin_nontemporal_store(o, reinterpret_cast<nhmc_v4f*>(&out_img[off]));
  }
  NHMC_STAMP(6);
  if (EPI == EPI_RESID || EPI == EPI_SRES) {
    __shared__ double red[NW];
    double sw = nhmc_wave_sum((double)lsum);
    if (lane == 0) red[wave] = sw;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
      for (int w = 0; w < NW; ++w) tot += red[w];
      ws[(int64_t)img * 4 + q] = tot;
    }
  }
}

constexpr int PAIR_LDS_BYTES = 256 * 64 * 4;                // one 64 KB slab region (S1 slab, then T1 slab, then output staging)

template <int EPI, bool PRECLIP, int NW, bool TOUT>
int pair256_nw(const float* IN, const float* S1, const float* S2, float* OUT, const float* Dmap, const float* aux, double* ws,
               int n_img, int channels, hipStream_t st, VjpArgs vj) {
  static bool attr_set[64] = {};                             // raise the dynamic-LDS limit of this instantiation once per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pair256<EPI, PRECLIP, NW, TOUT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS_BYTES) != hipSuccess)
      return NHMC_ERR_LAUNCH;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  NHMC_LAUNCH((k_pair256<EPI, PRECLIP, NW, TOUT>), dim3((unsigned)(4 * n_img)), dim3(64 * NW), PAIR_LDS_BYTES, st, IN, S1, S2,
              OUT, Dmap, aux, ws, channels, vj);
  return nhmc_launch_status();
}

template <int EPI, bool PRECLIP, bool TOUT = false>
int pair256(const float* IN, const float* S1, const float* S2, float* OUT, const float* Dmap, const float* aux, double* ws,
            int n_img, int channels, hipStream_t st, VjpArgs vj = VjpArgs{}) {
  return pair256_nw<EPI, PRECLIP, 8, TOUT>(IN, S1, S2, OUT, Dmap, aux, ws, n_img, channels, st, vj);
}

int tile_of2(int R, int C) { return (R % 128 == 0 && C % 128 == 0) ? 128 : ((R % 64 == 0 && C % 64 == 0) ? 64 : 32); }

// OUT[R][C] = IN[K][R]^T * S[K][C] per image; K % 32 == 0, R % 32 == 0, C % 32 == 0.
template <int EPI, bool PRECLIP, bool TOUT = false>
int gemm_krc(const float* IN, const float* S, float* OUT, const float* Dmap, const float* aux, double* ws, int n_img,
             int channels, int K, int R, int C, hipStream_t st, VjpArgs vj = VjpArgs{}) {
  const int T = tile_of2(R, C);
  dim3 grid((unsigned)((C / T) * (R / T) * n_img));
  if (T == 128)
    NHMC_LAUNCH((k_sgemm<128, 2, EPI, PRECLIP, TOUT>), grid, dim3(256), 0, st, IN, S, OUT, Dmap, aux, ws, K, R, C, channels, vj);
  else if (T == 64)
    NHMC_LAUNCH((k_sgemm<64, 2, EPI, PRECLIP, TOUT>), grid, dim3(256), 0, st, IN, S, OUT, Dmap, aux, ws, K, R, C, channels, vj);
  else
    NHMC_LAUNCH((k_sgemm<32, 1, EPI, PRECLIP, TOUT>), grid, dim3(64), 0, st, IN, S, OUT, Dmap, aux, ws, K, R, C, channels, vj);
  return nhmc_launch_status();
}

template <int EPI, bool PRECLIP, bool TOUT = false>
int gemm(const float* IN, const float* S, float* OUT, const float* Dmap, const float* aux, double* ws, int n_img,
         int channels, int d, hipStream_t st, VjpArgs vj = VjpArgs{}) {
  return gemm_krc<EPI, PRECLIP, TOUT>(IN, S, OUT, Dmap, aux, ws, n_img, channels, d, d, d, st, vj);
}

int tile_of(int d) { return d % 128 == 0 ? 128 : (d % 64 == 0 ? 64 : 32); }

// A/B switch for tools/pair_bench.hip and the parity tests: NHMC_SPECTRAL_PAIRS=0 keeps the one-product-per-launch chain.
bool pairs_enabled() {
  const char* v = getenv("NHMC_SPECTRAL_PAIRS");
  return !(v && v[0] == '0');
}

bool bad(int n_chains, int channels, int dim) {
  return n_chains <= 0 || channels <= 0 || dim <= 0 || (dim % 32) || (int64_t)n_chains * channels > 65535;
}

}  // namespace

extern "C" int nhmc_spectral_tiles(int channels, int dim) {
  const int t = dim / tile_of(dim);
  return channels * t * t;
}

// out_c = Lo (D_c o (L^T X_c R)) Ro^T.  Arguments: L and R as stored ([d][d] row-major), LoT = Lo^T and
// RoT = Ro^T as stored.
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
  if (dim == 256 && pairs_enabled()) {
    if ((rc = pair256<EPI_MULD, false>(x, L, R, tmp, Dmap, nullptr, nullptr, n, channels, st))) return rc;      // (L^T X R) o D
    return pair256<EPI_NONE, false>(tmp, LoT, RoT, out, nullptr, nullptr, nullptr, n, channels, st);              // Lo . Ro^T
  }
  if ((rc = gemm<EPI_NONE, false>(x, L, tmp, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc;     // X^T L
  if ((rc = gemm<EPI_MULD, false>(tmp, R, out, Dmap, nullptr, nullptr, n, channels, dim, st))) return rc;       // (L^T X R) o D
  if ((rc = gemm<EPI_NONE, false>(out, LoT, tmp, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc;  // (Lo .)^T
  return gemm<EPI_NONE, false>(tmp, RoT, out, nullptr, nullptr, nullptr, n, channels, dim, st);                 // Lo . Ro^T
}

// Data term for the spectral operator, in the reference's rounding sequence INCLUDING autograd's.  `factors` is the
// packed resident block [8][d][d] (row-major): U1, U2, V1, V2, U1^T, U2^T, V1^T, V2^T.
//   forward  (Hfuncs.py:65-71, Vt :493-499, U :501-509):  A = D o ((V1^T x) V2),  H x = (U1 A) U2^T   -- left factor first
//   backward (autograd of those matmuls):  dA4 = r~ U2,  dA3 = U1^T dA4,  dA2 = D o dA3,  dA1 = dA2 V2^T,  g = V1 dA1
//                                                                                    -- RIGHT factor first, r~ = -2 r
// Every product is an exact k-ascending FMA chain from zero on the MFMA (v_mfma_f32_32x32x2_f32), which is also what
// torch's CPU sgemm computes (tools/mfma_bits.py), so with the same stage order the data term is the reference's bits.
// The kernel form OUT = IN^T S multiplies from the left; the backward's right-first order is run on TRANSPOSED operands:
// the forward's last product stores r^T (TOUT; `yT` = the observation transposed per channel plane, constant over a run)
// and  (r^T)^T U2 = r U2, ... the adjoint's last product yields g^T, transposed back in its epilogue (TOUT) where the
// clip mask / the last DDIM step's VJP are applied in natural coordinates.  DmapT = D transposed per channel.
namespace {
template <bool PRECLIP>
int spectral_chain(const float* x, const float* yT, const float* factors, const float* Dmap, const float* DmapT,
                   const float* mask_or_xt, const VjpArgs* vjp, float* g_xt, double* loss_ws, float* tmp, int n,
                   int channels, int dim, hipStream_t st) {
  const int64_t dd = (int64_t)dim * dim;
  const float *U1 = factors, *U2 = factors + dd, *V1 = factors + 2 * dd, *V2 = factors + 3 * dd;
  const float *U1T = U1 + 4 * dd, *U2T = U1 + 5 * dd, *V1T = U1 + 6 * dd, *V2T = U1 + 7 * dd;
  float* A = tmp;
  float* B = tmp + (int64_t)n * dd;
  int rc;
  if (dim == 256 && pairs_enabled()) {                      // two products per launch, intermediates stay in LDS
    if ((rc = pair256<EPI_MULD, PRECLIP>(x, V1, V2, A, Dmap, nullptr, nullptr, n, channels, st))) return rc;          // D o (V1^T x V2)
    if ((rc = pair256<EPI_RESID, false, true>(A, U1T, U2T, B, nullptr, yT, loss_ws, n, channels, st))) return rc;      // r^T
    if ((rc = pair256<EPI_MULD, false>(B, U2, U1, A, DmapT, nullptr, nullptr, n, channels, st))) return rc;            // (D o (U1^T (r U2)))^T
    if (vjp) return pair256<EPI_VJP, false, true>(A, V2T, V1T, g_xt, nullptr, mask_or_xt, nullptr, n, channels, st, *vjp);
    return pair256<EPI_GRAD, false, true>(A, V2T, V1T, g_xt, nullptr, mask_or_xt, nullptr, n, channels, st);         // (V1 ((.) V2^T))
  }
  if ((rc = gemm<EPI_NONE, PRECLIP>(x, V1, A, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc;            // (V1^T x)^T
  if ((rc = gemm<EPI_MULD, false>(A, V2, B, Dmap, nullptr, nullptr, n, channels, dim, st))) return rc;                 // D o (V1^T x V2)
  if ((rc = gemm<EPI_NONE, false>(B, U1T, A, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc;             // (U1 A)^T
  if ((rc = gemm<EPI_RESID, false, true>(A, U2T, B, nullptr, yT, loss_ws, n, channels, dim, st))) return rc;           // r^T = (y - U1 A U2^T)^T
  if ((rc = gemm<EPI_NONE, false>(B, U2, A, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc;              // r U2
  if ((rc = gemm<EPI_MULD, false>(A, U1, B, DmapT, nullptr, nullptr, n, channels, dim, st))) return rc;                // (D o (U1^T r U2))^T
  if ((rc = gemm<EPI_NONE, false>(B, V2T, A, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc;             // (.) V2^T
  if (vjp) return gemm<EPI_VJP, false, true>(A, V1T, g_xt, nullptr, mask_or_xt, nullptr, n, channels, dim, st, *vjp);
  return gemm<EPI_GRAD, false, true>(A, V1T, g_xt, nullptr, mask_or_xt, nullptr, n, channels, dim, st);               // V1 (.)
}
}  // namespace

extern "C" int nhmc_data_spectral(const float* xt, const float* yT, const float* factors, const float* Dmap,
                                  const float* DmapT, int apply_clip, float* g_xt, double* loss_ws, float* tmp,
                                  int n_chains, int channels, int dim, nhmc_stream_t stream) {
  if (!xt || !yT || !factors || !Dmap || !DmapT || !g_xt || !loss_ws || !tmp) return NHMC_ERR_ARG;
  if (bad(n_chains, channels, dim)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(xt) || !nhmc_aligned16(yT) || !nhmc_aligned16(factors) || !nhmc_aligned16(g_xt) ||
      !nhmc_aligned16(tmp) || !nhmc_aligned16(Dmap) || !nhmc_aligned16(DmapT))
    return NHMC_ERR_ALIGN;
  const int n = n_chains * channels;
  if (apply_clip)
    return spectral_chain<true>(xt, yT, factors, Dmap, DmapT, xt, nullptr, g_xt, loss_ws, tmp, n, channels, dim, nhmc_s(stream));
  return spectral_chain<false>(xt, yT, factors, Dmap, DmapT, nullptr, nullptr, g_xt, loss_ws, tmp, n, channels, dim, nhmc_s(stream));
}

// Data term + VJP of the last DDIM step for the spectral operator: xt_next = the clipped decode (what k_mix_fwd with
// final_clip wrote), (xt, e) the step's inputs.  The eighth product's epilogue writes g_xt and g_e directly.
extern "C" int nhmc_data_spectral_vjp(const float* xt_next, const float* yT, const float* factors, const float* Dmap,
                                      const float* DmapT, const float* xt, const float* e, int e_channels,
                                      const float* at, const float* at_next, float* g_xt, float* g_e, double* loss_ws,
                                      float* tmp, int n_chains, int channels, int dim, nhmc_stream_t stream) {
  if (!xt_next || !yT || !factors || !Dmap || !DmapT || !xt || !e || !at || !at_next || !g_xt || !g_e || !loss_ws || !tmp)
    return NHMC_ERR_ARG;
  if (bad(n_chains, channels, dim) || (e_channels != channels && e_channels != 2 * channels)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(xt_next) || !nhmc_aligned16(yT) || !nhmc_aligned16(factors) || !nhmc_aligned16(g_xt) ||
      !nhmc_aligned16(tmp) || !nhmc_aligned16(xt) || !nhmc_aligned16(e) || !nhmc_aligned16(g_e) || !nhmc_aligned16(Dmap) ||
      !nhmc_aligned16(DmapT))
    return NHMC_ERR_ALIGN;
  const VjpArgs vjp{e, g_e, at, at_next, e_channels};
  return spectral_chain<false>(xt_next, yT, factors, Dmap, DmapT, xt, &vjp, g_xt, loss_ws, tmp, n_chains * channels, channels,
                               dim, nhmc_s(stream));
}

// ---- the data term with the residual taken in the left singular basis ------------------------------------------------
// U1, U2 are orthogonal (full SVDs, Hfuncs.py:473-474), so with y^ = U1^T y U2 (constant over a run: one sandwich when
// the observation is set)
//     y - H x = U1 (y^ - D o S) U2^T,   S = V1^T x V2      =>   |y - H x|^2 = |y^ - D o S|^2,
//     H^T (y - H x) = V1 (D o (y^ - D o S)) V2^T,
// the two products by U^T and the two by U of the chain above cancel: FOUR products per evaluation instead of eight,
// one intermediate instead of three.  Not the reference's rounding sequence: against the reference's operator evaluated
// in fp64 the gradient moves by 3-7e-6 relative (the factors' 1e-6 departure from orthogonality times |H x| / |r|; the
// reference's own fp32 evaluation sits 1-2e-6 from the same fp64 value), the loss by 1e-7 -- inside the 1e-4 contract
// (tests/test_spectral_proj_gpu.py measures both against the eight-product form).
extern "C" int nhmc_spectral_project(const float* y, const float* L, const float* R, float* out, float* tmp, int n_chains,
                                     int channels, int dim, nhmc_stream_t stream) {
  if (!y || !L || !R || !out || !tmp) return NHMC_ERR_ARG;
  if (bad(n_chains, channels, dim)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(y) || !nhmc_aligned16(L) || !nhmc_aligned16(R) || !nhmc_aligned16(out) || !nhmc_aligned16(tmp))
    return NHMC_ERR_ALIGN;
  hipStream_t st = nhmc_s(stream);
  const int n = n_chains * channels;
  if (dim == 256 && pairs_enabled()) return pair256<EPI_NONE, false>(y, L, R, out, nullptr, nullptr, nullptr, n, channels, st);
  int rc;
  if ((rc = gemm<EPI_NONE, false>(y, L, tmp, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc;       // Y^T L
  return gemm<EPI_NONE, false>(tmp, R, out, nullptr, nullptr, nullptr, n, channels, dim, st);                    // L^T Y R
}

namespace {
// shared body: `vjp` == nullptr -> EPI_GRAD (mask from `clip_src` when given), else the last DDIM step's VJP epilogue
int data_spectral_proj(const float* x_in, bool preclip, const float* y_proj, const float* factors, const float* Dmap,
                       const float* mask_or_xt, const VjpArgs* vjp, float* g_xt, double* loss_ws, float* tmp, int n,
                       int channels, int dim, hipStream_t st) {
  const int64_t dd = (int64_t)dim * dim;
  const float *V1 = factors + 2 * dd, *V2 = factors + 3 * dd, *V1T = factors + 6 * dd, *V2T = factors + 7 * dd;
  float* A = tmp;
  float* B = g_xt;                                          // free until the last product writes it
  int rc;
  if (dim == 256 && pairs_enabled()) {
    if (preclip) { if ((rc = pair256<EPI_SRES, true>(x_in, V1, V2, A, Dmap, y_proj, loss_ws, n, channels, st))) return rc; }
    else         { if ((rc = pair256<EPI_SRES, false>(x_in, V1, V2, A, Dmap, y_proj, loss_ws, n, channels, st))) return rc; }
    if (vjp) return pair256<EPI_VJP, false>(A, V1T, V2T, g_xt, nullptr, mask_or_xt, nullptr, n, channels, st, *vjp);
    return pair256<EPI_GRAD, false>(A, V1T, V2T, g_xt, nullptr, mask_or_xt, nullptr, n, channels, st);
  }
  if (preclip) { if ((rc = gemm<EPI_NONE, true>(x_in, V1, A, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc; }
  else         { if ((rc = gemm<EPI_NONE, false>(x_in, V1, A, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc; }
  if ((rc = gemm<EPI_SRES, false>(A, V2, B, Dmap, y_proj, loss_ws, n, channels, dim, st))) return rc;
  if ((rc = gemm<EPI_NONE, false>(B, V1T, A, nullptr, nullptr, nullptr, n, channels, dim, st))) return rc;
  if (vjp) return gemm<EPI_VJP, false>(A, V2T, g_xt, nullptr, mask_or_xt, nullptr, n, channels, dim, st, *vjp);
  return gemm<EPI_GRAD, false>(A, V2T, g_xt, nullptr, mask_or_xt, nullptr, n, channels, dim, st);
}
}  // namespace

// nhmc_data_spectral with y_proj = U1^T y U2 (nhmc_spectral_project(y, U1, U2)) in place of y.  tmp: float[n*C*d*d]
// (the one-product-per-launch form parks its second intermediate in g_xt).
extern "C" int nhmc_data_spectral_proj(const float* xt, const float* y_proj, const float* factors, const float* Dmap,
                                       int apply_clip, float* g_xt, double* loss_ws, float* tmp, int n_chains,
                                       int channels, int dim, nhmc_stream_t stream) {
  if (!xt || !y_proj || !factors || !Dmap || !g_xt || !loss_ws || !tmp) return NHMC_ERR_ARG;
  if (bad(n_chains, channels, dim)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(xt) || !nhmc_aligned16(y_proj) || !nhmc_aligned16(factors) || !nhmc_aligned16(g_xt) ||
      !nhmc_aligned16(tmp) || !nhmc_aligned16(Dmap))
    return NHMC_ERR_ALIGN;
  return data_spectral_proj(xt, apply_clip != 0, y_proj, factors, Dmap, apply_clip ? xt : nullptr, nullptr, g_xt, loss_ws,
                            tmp, n_chains * channels, channels, dim, nhmc_s(stream));
}

// nhmc_data_spectral_vjp with y_proj in place of y.
extern "C" int nhmc_data_spectral_proj_vjp(const float* xt_next, const float* y_proj, const float* factors,
                                           const float* Dmap, const float* xt, const float* e, int e_channels,
                                           const float* at, const float* at_next, float* g_xt, float* g_e,
                                           double* loss_ws, float* tmp, int n_chains, int channels, int dim,
                                           nhmc_stream_t stream) {
  if (!xt_next || !y_proj || !factors || !Dmap || !xt || !e || !at || !at_next || !g_xt || !g_e || !loss_ws || !tmp)
    return NHMC_ERR_ARG;
  if (bad(n_chains, channels, dim) || (e_channels != channels && e_channels != 2 * channels)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(xt_next) || !nhmc_aligned16(y_proj) || !nhmc_aligned16(factors) || !nhmc_aligned16(g_xt) ||
      !nhmc_aligned16(tmp) || !nhmc_aligned16(xt) || !nhmc_aligned16(e) || !nhmc_aligned16(g_e) || !nhmc_aligned16(Dmap))
    return NHMC_ERR_ALIGN;
  const VjpArgs vjp{e, g_e, at, at_next, e_channels};
  return data_spectral_proj(xt_next, false, y_proj, factors, Dmap, xt, &vjp, g_xt, loss_ws, tmp, n_chains * channels,
                            channels, dim, nhmc_s(stream));
}

// ---- separable strided convolution (SRConv, obs_functions/Hfuncs.py:527-607) ----------------------------------
// The reference applies its SVD factors one after the other, each a rounded fp32 stage (H at Hfuncs.py:65-71 with
// Vt :576-583, singulars :598-599, U :585-592):
//     H(X)   = U  ( S o (V1^T X V1) ) U^T        V1 = V_small[:, :sd] ([d][sd]), U = U_small ([sd][sd]),
//     H^T(Y) = V1 ( S o (U^T  Y U ) ) V1^T       S[i][j] = fl(s_i * s_j) (:560; thresholded singular values),
//     H^+(Y) = V1 ( S+ o (U^T Y U ) ) V1^T       S+ = 1 / S where S != 0 (:82-88)
// -- the Deblurring2D chain with rectangular factors and one multiplier map for all channels.  Round 2 collapsed the
// factors into A = U diag(s) V1^T (two products); the stages are kept now because whole-run agreement with the
// reference needs its rounding sequence, not only its mathematics.
// All products are the one kernel form OUT = IN^T S:
//   nhmc_sandwich_rect: t = in^T S1 ([R1][C1]),  out = (t^T S2) o mul ([C1][C2]; mul nullable) = (S1^T in S2) o mul.
extern "C" int nhmc_sandwich_rect(const float* in, const float* S1, const float* S2, const float* mul, float* out,
                                  float* tmp, int n_img, int K1, int R1, int C1, int C2, nhmc_stream_t stream) {
  if (!in || !S1 || !S2 || !out || !tmp) return NHMC_ERR_ARG;
  if (n_img <= 0 || n_img > 65535 || (K1 % 32) || (R1 % 32) || (C1 % 32) || (C2 % 32) || K1 <= 0 || R1 <= 0 || C1 <= 0 || C2 <= 0)
    return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(in) || !nhmc_aligned16(S1) || !nhmc_aligned16(S2) || !nhmc_aligned16(out) || !nhmc_aligned16(tmp) ||
      !nhmc_aligned16(mul))
    return NHMC_ERR_ALIGN;
  hipStream_t st = nhmc_s(stream);
  int rc;
  // t[R1][C1] = in[K1][R1]^T S1[K1][C1];   out[C1][C2] = t[R1][C1]^T S2[R1][C2]
  if ((rc = gemm_krc<EPI_NONE, false>(in, S1, tmp, nullptr, nullptr, nullptr, n_img, 1, K1, R1, C1, st))) return rc;
  if (mul) return gemm_krc<EPI_MULD, false>(tmp, S2, out, mul, nullptr, nullptr, n_img, 1, R1, C1, C2, st);
  return gemm_krc<EPI_NONE, false>(tmp, S2, out, nullptr, nullptr, nullptr, n_img, 1, R1, C1, C2, st);
}

extern "C" int nhmc_srconv_tiles(int channels, int small_dim) {
  const int t = small_dim / tile_of2(small_dim, small_dim);
  return channels * t * t;
}

namespace {
struct SrconvFactors { const float *V1, *V1T, *U, *UT, *S; };   // [d][sd], [sd][d], [sd][sd], [sd][sd], [sd][sd]

// r = y - U (S o (V1^T x V1)) U^T with loss partials, stored transposed (TOUT; yT = the observation transposed per
// plane); then the adjoint in autograd's order -- r U, U^T (.), S o, (.) V1^T, V1 (.) : right factor first, run on the
// transposed operands as in spectral_chain -- the last product transposing back under the gradient epilogue (`vjp` ==
// nullptr: g = -2 H^T r (x) clip mask of `mask_or_xt` when given) or the last DDIM step's VJP.  S is symmetric.
// tmp: float[n * (d*sd + 3*sd*sd)].
template <bool PRECLIP>
int srconv_chain(const float* x, const float* yT, const SrconvFactors& f, const float* mask_or_xt, const VjpArgs* vjp,
                 float* g_xt, double* loss_ws, float* tmp, int n, int channels, int d, int sd, hipStream_t st) {
  float* T1 = tmp;                                    // [d][sd], later [sd][d]
  float* Z = T1 + (int64_t)n * d * sd;                // [sd][sd]
  float* T2 = Z + (int64_t)n * sd * sd;               // [sd][sd]
  float* Rr = T2 + (int64_t)n * sd * sd;              // [sd][sd]
  int rc;
  if ((rc = gemm_krc<EPI_NONE, PRECLIP>(x, f.V1, T1, nullptr, nullptr, nullptr, n, channels, d, d, sd, st))) return rc;   // (V1^T x)^T
  if ((rc = gemm_krc<EPI_MULD, false>(T1, f.V1, Z, f.S, nullptr, nullptr, n, 1, d, sd, sd, st))) return rc;              // S o (V1^T x V1)
  if ((rc = gemm_krc<EPI_NONE, false>(Z, f.UT, T2, nullptr, nullptr, nullptr, n, channels, sd, sd, sd, st))) return rc;  // (U Z)^T
  if ((rc = gemm_krc<EPI_RESID, false, true>(T2, f.UT, Rr, nullptr, yT, loss_ws, n, channels, sd, sd, sd, st))) return rc;  // r^T
  if ((rc = gemm_krc<EPI_NONE, false>(Rr, f.U, T2, nullptr, nullptr, nullptr, n, channels, sd, sd, sd, st))) return rc;  // r U
  if ((rc = gemm_krc<EPI_MULD, false>(T2, f.U, Z, f.S, nullptr, nullptr, n, 1, sd, sd, sd, st))) return rc;              // (S o (U^T r U))^T
  if ((rc = gemm_krc<EPI_NONE, false>(Z, f.V1T, T1, nullptr, nullptr, nullptr, n, channels, sd, sd, d, st))) return rc;  // (.) V1^T
  if (vjp) return gemm_krc<EPI_VJP, false, true>(T1, f.V1T, g_xt, nullptr, mask_or_xt, nullptr, n, channels, sd, d, d, st, *vjp);
  return gemm_krc<EPI_GRAD, false, true>(T1, f.V1T, g_xt, nullptr, mask_or_xt, nullptr, n, channels, sd, d, d, st);     // V1 (.)
}

bool srconv_bad_shape(int n_chains, int channels, int dim, int small_dim) {
  return n_chains <= 0 || channels <= 0 || (int64_t)n_chains * channels > 65535 || (dim % 32) || (small_dim % 32) ||
         dim <= 0 || small_dim <= 0 || small_dim > dim;
}
}  // namespace

// Data term: r = y - H(clip(xt)), loss partials (nhmc_srconv_tiles per chain), g = -2 H^T r (x) clip mask.
// y = the observation TRANSPOSED per channel plane ([n_chains][C][sd][sd], each plane transposed).
extern "C" int nhmc_data_srconv(const float* xt, const float* y, const float* V1, const float* V1T, const float* U,
                                const float* UT, const float* S, int apply_clip, float* g_xt, double* loss_ws, float* tmp,
                                int n_chains, int channels, int dim, int small_dim, nhmc_stream_t stream) {
  if (!xt || !y || !V1 || !V1T || !U || !UT || !S || !g_xt || !loss_ws || !tmp) return NHMC_ERR_ARG;
  if (srconv_bad_shape(n_chains, channels, dim, small_dim)) return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(xt) || !nhmc_aligned16(y) || !nhmc_aligned16(V1) || !nhmc_aligned16(V1T) || !nhmc_aligned16(U) ||
      !nhmc_aligned16(UT) || !nhmc_aligned16(S) || !nhmc_aligned16(g_xt) || !nhmc_aligned16(tmp))
    return NHMC_ERR_ALIGN;
  const SrconvFactors f{V1, V1T, U, UT, S};
  const int n = n_chains * channels;
  if (apply_clip)
    return srconv_chain<true>(xt, y, f, xt, nullptr, g_xt, loss_ws, tmp, n, channels, dim, small_dim, nhmc_s(stream));
  return srconv_chain<false>(xt, y, f, nullptr, nullptr, g_xt, loss_ws, tmp, n, channels, dim, small_dim, nhmc_s(stream));
}

// Same with the VJP of the last DDIM step applied in the last product's epilogue (xt_next = the clipped decode).
extern "C" int nhmc_data_srconv_vjp(const float* xt_next, const float* y, const float* V1, const float* V1T,
                                    const float* U, const float* UT, const float* S, const float* xt, const float* e,
                                    int e_channels, const float* at, const float* at_next, float* g_xt, float* g_e,
                                    double* loss_ws, float* tmp, int n_chains, int channels, int dim, int small_dim,
                                    nhmc_stream_t stream) {
  if (!xt_next || !y || !V1 || !V1T || !U || !UT || !S || !xt || !e || !at || !at_next || !g_xt || !g_e || !loss_ws || !tmp)
    return NHMC_ERR_ARG;
  if (srconv_bad_shape(n_chains, channels, dim, small_dim) || (e_channels != channels && e_channels != 2 * channels))
    return NHMC_ERR_SHAPE;
  if (!nhmc_aligned16(xt_next) || !nhmc_aligned16(y) || !nhmc_aligned16(V1) || !nhmc_aligned16(V1T) || !nhmc_aligned16(U) ||
      !nhmc_aligned16(UT) || !nhmc_aligned16(S) || !nhmc_aligned16(xt) || !nhmc_aligned16(e) || !nhmc_aligned16(g_xt) ||
      !nhmc_aligned16(g_e) || !nhmc_aligned16(tmp))
    return NHMC_ERR_ALIGN;
  const SrconvFactors f{V1, V1T, U, UT, S};
  const VjpArgs vj{e, g_e, at, at_next, e_channels};
  return srconv_chain<false>(xt_next, y, f, xt, &vj, g_xt, loss_ws, tmp, n_chains * channels, channels, dim, small_dim,
                             nhmc_s(stream));
}
