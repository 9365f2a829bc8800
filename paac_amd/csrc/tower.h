// Conv tower of the Nature network (networks.py:154-169: conv1 8x8 s4 -> conv2 4x4 s2 -> conv3 3x3 s1, each + bias + ReLU,
// input = uint8 * (1/255), networks.py:115) as ONE launch for gfx950: a workgroup owns one sample (or one spatial region of
// one sample) end to end and keeps every intermediate activation in LDS, so the three layers cost one kernel boundary
// instead of three and no activation makes a round trip through L2 / Infinity Cache between layers.
//
//   * Arithmetic = the fp32-exact bf16-MFMA schemes of dmm.h (XB = 1 for conv1, XB = 2 for conv2 / conv3), on
//     v_mfma_f32_16x16x32_bf16: u8 pixels are exact in bf16; an fp32 weight or activation is split EXACTLY into three bf16
//     terms (truncate, subtract, repeat); conv1 accumulates the 3 exact products per multiply, conv2 / conv3 the 6 products
//     down to 2^-16 of the leading one (the 3 dropped ones are below one fp32 rounding); accumulation in fp32.
//   * The splitting is done ONCE per value instead of once per use: weights are pre-split into (hi, mid, lo) bf16 planes in
//     the MFMA operand order by pack_tower_kernel (re-run whenever the weights change: fused behind the RMSProp step),
//     activations are split by the PRODUCING layer's epilogue and live in LDS as three bf16 planes -- the K loops are pure
//     ds_read_b128 + global 16-byte loads + MFMA, no VALU.
//   * "Transposed" contraction: out^T[Cout, pixels] = W^T[Cout, K] * patches^T[K, pixels].  The MFMA's A operand is a
//     pre-packed weight fragment (row = output channel), the B operand a patch fragment (column = output pixel): both are
//     8 consecutive K values = 16 contiguous bytes per lane, and a D tile leaves 4 consecutive channels of one pixel in a
//     lane -- one 8-byte LDS store per plane (one 16-byte global store when the fp32 activation is kept for backward).
//   * 8 waves per workgroup.  A wave owns one 16-channel tile (weights are streamed from L2 exactly once per workgroup)
//     and either all pixel tiles over half of K (the two halves summed through LDS) or half of the pixel tiles over all
//     of K, whichever divides evenly.
//   * Regions: with 32 acting rows, one workgroup per sample would light 32 of 256 CUs.  A sample is cut into 2x2
//     overlapping regions of 4x4 conv3 outputs (origins 0 / 3; conv2 6x6, conv1 14x14, input 60x60 each): 128 workgroups,
//     about 2x redundant conv1 / conv2 arithmetic, no exchange between workgroups (up to 16 rows: 2x4 regions of 4x2, the
//     last column shifted back to end at 7).  Training batches (>= 160 rows) use one workgroup per sample (and also write
//     the fp32 activations the backward pass reads).
//   * LDS image strides are padded (conv1 plane 80 B per pixel, conv2 plane 160 B) so that the 16 lanes a ds_read_b128
//     services together fall on 64 distinct banks for the stride-2 / stride-1 walks of conv2 / conv3.
#pragma once
#include "dmm.h"

namespace paac {

struct TowerArgs {
  const uint8_t* states;                   // [B,84,84,4]
  const bf16x8* w1p;                        // packed planes, conv1: [8 k-steps][2 channel tiles][3 planes][64 lanes]
  const bf16x8* w2p;                        // conv2: [16][4][3][64]
  const bf16x8* w3p;                        // conv3: [18][4][3][64]
  const float* b1;
  const float* b2;
  const float* b3;
  float* act1;                             // [B,20,20,32] fp32 (WRITE_ALL only)
  float* act2;                             // [B,9,9,64]   fp32 (WRITE_ALL only)
  float* act3;                             // [B,7,7,64]   fp32 = the fc layer's input rows (flatten keeps HWC, networks.py:6-9)
  int act3_packed;                         // 1: act3 is written in fc_heads_kernel's A-fragment order instead:
                                           //    [row tile b/16][K group k/16][(k%16)/4 * 16 + b%16][k%4], k = (y*7 + x)*64 + c
  float* act3_rows;                        // nullable: a second, plain-row copy of conv3's output (acting rows kept for the
                                           // update: the fc weight gradient reads rows, the fc kernel fragments)
  int batch;
#ifdef PAAC_DMM_STAMPS
  unsigned long long* stamps;               // diagnostic build only: 12 x u64 per wave
#endif
};

#ifdef PAAC_DMM_STAMPS
#define TOWER_STAMP(i)                                                                                      \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (p.stamps && lane == 0)                                                                              \
      p.stamps[((long)blockIdx.x * 8 + wave) * 12 + (i)] =                                                  \
          ((i) == 0 || (i) == 11) ? (unsigned long long)wall_clock64() : (unsigned long long)clock64();    \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  } while (0)
#else
#define TOWER_STAMP(i)
#endif

#ifndef PAAC_T_KS2
#define PAAC_T_KS2 2
#endif
#ifndef PAAC_T_KS3
#define PAAC_T_KS3 2
#endif
constexpr int kTowerW1Vecs = 8 * 2 * 3 * 64, kTowerW2Vecs = 16 * 4 * 3 * 64, kTowerW3Vecs = 18 * 4 * 3 * 64;   // bf16x8 each
constexpr int kTowerPackVecs = kTowerW1Vecs + kTowerW2Vecs + kTowerW3Vecs;
// data-gradient weights of the backward tower (dgrad_tower.h), stored behind the forward planes in the same buffer
constexpr int kDgradW3Vecs = 18 * 4 * 3 * 64, kDgradW2Vecs = 4 * 8 * 2 * 3 * 64;
constexpr int kTowerPackAllVecs = kTowerPackVecs + kDgradW3Vecs + kDgradW2Vecs;

// ---------------------------------------------------------------------------------------------
// Weight packing: fp32 HWIO conv weights ([K, Cout] row-major, K = (kh, kw, cin)) -> per (k-step of 32, 16-channel tile):
// three planes of 64 lanes x 8 bf16; lane (ch = l & 15, kq = l >> 4) holds W[32 s + 8 kq + j][16 ct + ch], j = 0..7.
__device__ __forceinline__ void split3_store(const float (&x)[8], bf16x8* dst /* plane 0 of this lane */) {
  bf16x8 h, m, l;
  split3_bf16(x, h, m, l);
  dst[0] = h;
  dst[64] = m;
  dst[128] = l;
}

static __global__ __launch_bounds__(256) void pack_tower_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                         const float* __restrict__ w3, bf16x8* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;   // one thread per (k-step, channel tile, lane)
  const float* w;
  int cout, ctiles;
  bf16x8* dst;
  if (i < 8 * 2 * 64) {
    w = w1; cout = 32; ctiles = 2; dst = out;
  } else if (i < 8 * 2 * 64 + 16 * 4 * 64) {
    i -= 8 * 2 * 64;
    w = w2; cout = 64; ctiles = 4; dst = out + kTowerW1Vecs;
  } else if (i < 8 * 2 * 64 + 16 * 4 * 64 + 18 * 4 * 64) {
    i -= 8 * 2 * 64 + 16 * 4 * 64;
    w = w3; cout = 64; ctiles = 4; dst = out + kTowerW1Vecs + kTowerW2Vecs;
  } else {
    return;
  }
  const int lane = i & 63, unit = i >> 6;   // unit = s * ctiles + ct
  const int ct = unit % ctiles, s = unit / ctiles;
  const int ch = lane & 15, kq = lane >> 4;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = w[(long)(32 * s + 8 * kq + j) * cout + 16 * ct + ch];
  split3_store(x, dst + (long)unit * 192 + lane);
}

// ---------------------------------------------------------------------------------------------
template <int R3H_, int R3W_>
struct TowerGeom {
  static constexpr int R3H = R3H_, R3W = R3W_;                         // conv3 outputs per region
  static constexpr int R2H = R3H + 2, R2W = R3W + 2;                   // conv2 outputs it needs (3x3, stride 1)
  static constexpr int R1H = 2 * R2H + 2, R1W = 2 * R2W + 2;           // conv1 outputs (4x4, stride 2)
  static constexpr int RIH = 4 * R1H + 4, RIW = 4 * R1W + 4;           // input pixels (8x8, stride 4)
  static constexpr int NRY = (7 + R3H - 1) / R3H, NRX = (7 + R3W - 1) / R3W, NR = NRY * NRX;   // regions per axis: the last one is shifted back to end at 7
  static constexpr int P1 = R1H * R1W, P2 = R2H * R2W, P3 = R3H * R3W;
  static constexpr int PT1 = (P1 + 15) / 16, PT2 = (P2 + 15) / 16, PT3 = (P3 + 15) / 16;   // 16-pixel tiles
  static constexpr int S1 = 80, S2 = 160;                              // bytes per pixel in the conv1 / conv2 LDS planes
  static constexpr int PL1 = P1 * S1, PL2 = P2 * S2;                   // plane strides
  // K split over the two waves of a channel tile (else the pixel tiles are): forced when the tiles do not halve -- and for
  // the small regions, where a weight fragment fetched by BOTH waves for one or two pixel tiles each makes the layer wait
  // for the CU's L1 (stamped at 32 rows: conv2 6.4 k cycles for 3.1 k of MFMA issue, 392 KB of fragments through the L1)
  static constexpr bool KSPLIT2 = (PT2 % 2) != 0 || PT2 <= PAAC_T_KS2, KSPLIT3 = (PT3 % 2) != 0 || PT3 <= PAAC_T_KS3;
  static constexpr int NT1 = (PT1 + 3) / 4;                            // conv1: pixel tiles per wave (4 pixel groups x 2 channel tiles)
  static constexpr int NT2 = KSPLIT2 ? PT2 : PT2 / 2, NT3 = KSPLIT3 ? PT3 : PT3 / 2;
  // weight prefetch depth per layer (k-steps ahead); a depth >= the k-step count loads every fragment up front
  static constexpr int PF1 = 8, PF2 = KSPLIT2 ? 8 : 6, PF3 = KSPLIT3 ? 9 : 6;
  static constexpr int IN_BYTES = RIH * RIW * 8;
  static constexpr int SCR2 = KSPLIT2 ? 4 * NT2 * 1024 : 0;            // K-split partials of 4 waves, f32x4 per lane and tile
  static constexpr int FRONT = (IN_BYTES > 3 * PL2 + SCR2) ? IN_BYTES : 3 * PL2 + SCR2;   // input image, later conv2 planes (+ partials)
  static constexpr int FRONT_AL = (FRONT + 15) / 16 * 16;
  static constexpr int W1_BYTES = kTowerW1Vecs * 16;                    // conv1's packed weights: 48 KB
  // conv1's pixel tiles are split over waves that share a channel tile, so each weight fragment would be fetched by four
  // waves: where LDS has room the workgroup stages conv1's weights once (and then has the registers to request ALL of
  // conv2's weight fragments at kernel start)
  static constexpr bool W1_LDS = FRONT_AL + 3 * PL1 + W1_BYTES <= 160 * 1024;
  static constexpr int LDS_BYTES = FRONT_AL + 3 * PL1 + (W1_LDS ? W1_BYTES : 0);
  static_assert(RIW % 4 == 0, "input rows are staged 4 pixels (16 bytes) at a time");
  static_assert(!KSPLIT3 || 4 * NT3 * 1024 <= 3 * PL1, "conv3 K-split partials reuse the conv1 planes");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ f32x4 mfma_bf16(const bf16x8 a, const bf16x8 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// One wave: acc[t] += sum over NS k-steps of  A(weights, global k-step sg_mul * i + sg_add) x B(patch fragments of tile t).
//   wl     : packed weights of the layer + lane; unit (k-step sg, channel tile ct) = 192 vectors (3 planes x 64 lanes)
//   KOFF   : functor, KOFF::at(i) = byte offset of k-step i inside the LDS image (compile-time after unrolling)
//   BP     : B planes: 1 (exact bf16 operand) or 3 (hi, mid, lo at BPL bytes apart)
//   PF     : weight fragments are requested PF k-steps ahead of their MFMAs (PF >= NS: all of them up front)
// The weight prefetch is split off (prologue) so a phase can request its first fragments BEFORE the barrier that ends the
// previous phase; patch fragments of step i + 1 are read from LDS before the MFMAs of step i.  sched_barrier pins that
// order: left alone, the machine scheduler sinks every load to just before its first use (s_waitcnt vmcnt(0) per step).
struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};

template <int NS, int NT, int CT, int BP, int BPL, int PF_, class KOFF, bool DB = true>
struct WaveGemm {
  static constexpr int PF = PF_ < NS ? PF_ : NS;
  static constexpr int RING = PF < NS ? PF + 1 : NS;
  bf16x8 a[RING][3];
  const bf16x8* wl;        // global (or LDS, generic address space: the compiler sees which from the caller's pointer)
  int ct, sg_mul, sg_add;

  __device__ __forceinline__ void set(const bf16x8* wl_, const int ct_, const int sg_mul_, const int sg_add_) {
    wl = wl_; ct = ct_; sg_mul = sg_mul_; sg_add = sg_add_;
  }
  __device__ __forceinline__ void load_a(const int slot, const int i) {
    const bf16x8* src = wl + (long)((sg_mul * i + sg_add) * CT + ct) * 192;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) a[slot][pl] = src[pl * 64];
  }
  // request the weight fragments of prefetch step j (0 <= j < PF); out-of-range j: nothing
  __device__ __forceinline__ void prologue_step(const int j) {
    if (j >= 0 && j < PF) load_a(j, j);
  }
  __device__ __forceinline__ void prologue(const bf16x8* wl_, const int ct_, const int sg_mul_, const int sg_add_) {
    set(wl_, ct_, sg_mul_, sg_add_);
#pragma unroll
    for (int i = 0; i < PF; ++i) load_a(i, i);
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void read_b(bf16x8 (&b)[NT][BP], const char* lds_b, const unsigned (&bb)[NT], const int i) {
    const int ko = KOFF::at(i);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int pl = 0; pl < BP; ++pl) b[t][pl] = *reinterpret_cast<const bf16x8*>(lds_b + bb[t] + ko + pl * BPL);
  }
  // hook(i): extra requests issued with step i's own (the NEXT layer's weight fragments, into the registers this layer's
  // consumed fragments free)
  template <class Hook = NoHook>
  __device__ __forceinline__ void run(const char* __restrict__ lds_b, const unsigned (&bb)[NT], f32x4 (&acc)[NT],
                                      Hook hook = Hook()) {
    // DB: the patch fragments of step i + 1 are read while step i computes (two register sets); !DB (many tiles per
    // wave): one set, read at the top of the step
    bf16x8 b[DB ? 2 : 1][NT][BP];
    if constexpr (DB) read_b(b[0], lds_b, bb, 0);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (i + PF < NS) load_a((i + PF) % RING, i + PF);
      hook(i);
      if constexpr (DB) {
        if (i + 1 < NS) read_b(b[(i + 1) & 1], lds_b, bb, i + 1);
      } else {
        read_b(b[0], lds_b, bb, i);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int slot = i % RING;
      constexpr int NB = DB ? 2 : 1;
      if constexpr (BP == 1) {
        // smallest terms first; plane-major, so that consecutive MFMAs are independent (a lone wave on its SIMD has no
        // partner to fill the slots between dependent ones)
#pragma unroll
        for (int pl = 2; pl >= 0; --pl)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = mfma_bf16(a[slot][pl], b[i % NB][t][0], acc[t]);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x8(&bt)[BP] = b[i % NB][t];
        if constexpr (BP == 1) {
        } else {
          acc[t] = mfma_bf16(a[slot][2], bt[0], acc[t]);
          acc[t] = mfma_bf16(a[slot][1], bt[1], acc[t]);
          acc[t] = mfma_bf16(a[slot][0], bt[2], acc[t]);
          acc[t] = mfma_bf16(a[slot][1], bt[0], acc[t]);
          acc[t] = mfma_bf16(a[slot][0], bt[1], acc[t]);
          acc[t] = mfma_bf16(a[slot][0], bt[0], acc[t]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// 4 fp32 (consecutive channels of one pixel) -> (hi, mid, lo) bf16 x 4 -> one 8-byte LDS store per plane
template <int PLANE_STRIDE>
__device__ __forceinline__ void store_split4(char* dst, const f32x4 v) {
  unsigned xb[4], r1b[4], r2b[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x = v[e];   // a scalar copy: __builtin_bit_cast applied to the vector-element expression itself reads element 0
    xb[e] = __builtin_bit_cast(unsigned, x);
    const float r1 = x - __builtin_bit_cast(float, xb[e] & 0xFFFF0000u);
    r1b[e] = __builtin_bit_cast(unsigned, r1);
    const float r2 = r1 - __builtin_bit_cast(float, r1b[e] & 0xFFFF0000u);
    r2b[e] = __builtin_bit_cast(unsigned, r2);
  }
  *reinterpret_cast<u32x2*>(dst) = (u32x2){pack_hi16(xb[0], xb[1]), pack_hi16(xb[2], xb[3])};
  *reinterpret_cast<u32x2*>(dst + PLANE_STRIDE) = (u32x2){pack_hi16(r1b[0], r1b[1]), pack_hi16(r1b[2], r1b[3])};
  *reinterpret_cast<u32x2*>(dst + 2 * PLANE_STRIDE) = (u32x2){pack_hi16(r2b[0], r2b[1]), pack_hi16(r2b[2], r2b[3])};
}

template <class G>
struct Koff1 {   // conv1: k-step = kernel row kh; 32 K values = 8 pixels x 4 channels of that input row
  __device__ static constexpr int at(int i) { return i * G::RIW * 8; }
};
template <class G>
struct Koff2 {   // conv2: k-step = tap (kh, kw) of the 4x4 kernel, 32 channels
  __device__ static constexpr int at(int i) { return ((i / 4) * G::R1W + (i % 4)) * G::S1; }
};
template <class G>
struct Koff3Half {   // conv3, K split by channel half: k-step i = tap i of the 3x3 kernel (this wave's 32 channels)
  __device__ static constexpr int at(int i) { return ((i / 3) * G::R2W + (i % 3)) * G::S2; }
};
template <class G>
struct Koff3Full {   // conv3, all of K: k-step i = (tap i / 2, channel half i % 2)
  __device__ static constexpr int at(int i) { return (((i / 2) / 3) * G::R2W + ((i / 2) % 3)) * G::S2 + (i % 2) * 64; }
};

template <class G, bool WRITE_ALL>
__global__ __launch_bounds__(512) void tower_kernel(const TowerArgs p) {
  // WRITE_ALL with several regions per sample: pixels in the overlap are written by more than one workgroup, with identical values
  __shared__ __attribute__((aligned(16))) char lds[G::LDS_BYTES];
  char* const lds_in = lds;                       // bf16 input image [RIH][RIW][4]            (phase 0-1)
  char* const lds_a2 = lds;                       // conv2 planes [3][P2][S2]                  (phase 2-3), then
  char* const lds_s2 = lds + 3 * G::PL2;          // conv2 K-split partials                    (phase 2)
  char* const lds_a1 = lds + G::FRONT_AL;         // conv1 planes [3][P1][S1]                  (phase 1-2)
  char* const lds_s3 = lds_a1;                    // conv3 K-split partials                    (phase 3)
  char* const lds_w1 = lds_a1 + 3 * G::PL1;       // conv1's packed weights (W1_LDS)             (phase 0-1)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int b = blockIdx.x / G::NR, reg = blockIdx.x % G::NR;
  const int ry = reg / G::NRX, rx = reg % G::NRX;
  const int y3n = ry * G::R3H, x3n = rx * G::R3W;                   // first conv3 row / column this region owns
  const int y3a = y3n < 7 - G::R3H ? y3n : 7 - G::R3H, x3a = x3n < 7 - G::R3W ? x3n : 7 - G::R3W;   // region origin in conv3 / conv2 coordinates
  const int y1a = 2 * y3a, x1a = 2 * x3a;                          // ... in conv1 coordinates

  TOWER_STAMP(0);
  TOWER_STAMP(1);
  // conv1's weights: from LDS (staged below) with a short read-ahead, or straight from global, all 8 k-steps requested now
  WaveGemm<8, G::NT1, 2, 1, 0, G::W1_LDS ? 2 : G::PF1, Koff1<G>> g1;
  WaveGemm<G::KSPLIT2 ? 8 : 16, G::NT2, 4, 3, G::PL1, G::PF2, Koff2<G>> g2;

  // ---- phase 0: input region -> LDS as bf16 (a byte is exact in bf16) ---------------------------------------------
  {
    constexpr int VROW = G::RIW / 4, NV = G::RIH * VROW, ITERS = (NV + 511) / 512;
    const uint8_t* src = p.states + (size_t)b * 28224 + ((8 * y3a) * 84 + 8 * x3a) * 4;
    uint4 v[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int i = tid + it * 512;
      const int r = i / VROW, c4 = i - r * VROW;
      if (i < NV) v[it] = *reinterpret_cast<const uint4*>(src + (r * 84 + 4 * c4) * 4);
    }
    // behind the image (a wave's loads return in order, and the eight waves' requests share one queue: anything requested
    // ahead of the image delays the staging of every wave): conv1's packed weights, for LDS or as this wave's fragments
    constexpr int W1V = G::W1_LDS ? kTowerW1Vecs / 512 : 1;
    bf16x8 w1v[W1V];
    if constexpr (G::W1_LDS) {
#pragma unroll
      for (int it = 0; it < W1V; ++it) w1v[it] = p.w1p[tid + it * 512];
    } else {
      g1.prologue(p.w1p + lane, wave & 1, 1, 0);
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int i = tid + it * 512;
      if (i < NV) {
        const unsigned w[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
        u32x4 o[2];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          unsigned f[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) f[c] = __builtin_bit_cast(unsigned, (float)((w[px] >> (8 * c)) & 255u));
          o[px >> 1][2 * (px & 1)] = pack_hi16(f[0], f[1]);
          o[px >> 1][2 * (px & 1) + 1] = pack_hi16(f[2], f[3]);
        }
        u32x4* dst = reinterpret_cast<u32x4*>(lds_in + (size_t)i * 32);   // image order == vector order
        dst[0] = o[0];
        dst[1] = o[1];
      }
    }
    if constexpr (G::W1_LDS) {
#pragma unroll
      for (int it = 0; it < W1V; ++it) reinterpret_cast<bf16x8*>(lds_w1)[tid + it * 512] = w1v[it];
    }
  }
  TOWER_STAMP(2);
  __syncthreads();
  TOWER_STAMP(3);

  // ---- phase 1: conv1 (K = 256 = 8 kernel rows x 32) ------------------------------------------------------------------
  {
    const int ct = wave & 1, grp = wave >> 1;
    unsigned bb[G::NT1];
    int pix[G::NT1];                        // region pixel of this lane in tile j, -1 = none
#pragma unroll
    for (int j = 0; j < G::NT1; ++j) {
      const int tile = grp + 4 * j;
      const int pi = 16 * tile + li;
      const bool ok = (tile < G::PT1) && (pi < G::P1);
      pix[j] = ok ? pi : -1;
      const int pc = ok ? pi : 0;
      const int y1 = pc / G::R1W, x1 = pc - y1 * G::R1W;
      bb[j] = (unsigned)(((4 * y1) * G::RIW + 4 * x1 + 2 * kq) * 8);
    }
    f32x4 acc[G::NT1];
#pragma unroll
    for (int j = 0; j < G::NT1; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.b1 + 16 * ct + 4 * kq);
    if constexpr (G::W1_LDS) {
      // conv2's weight fragments stream in while conv1 computes (conv1's own weights come from LDS: the registers are
      // free), one k-step's per k-step of conv1: 8 waves x 24 fragment loads are 3 k cycles of the CU's load path and a wave
      // cannot issue past a full queue -- requested all at once in front of conv1's GEMM they held its first MFMAs back
      // (tower 10.9 -> 10.5 us at 32 rows).  Requested earlier still, behind the staging loads, they delay the image
      // instead (staging 1.6 k -> 4 k cycles, with or without a barrier between the two groups of requests); and four
      // waves of 4-5 pixel tiles instead of eight of 2-3 (less LDS traffic for the weight fragments) measured equal.
      g2.set(p.w2p + lane, wave & 3, 1, G::KSPLIT2 ? 8 * (wave >> 2) : 0);
      g1.prologue(reinterpret_cast<const bf16x8*>(lds_w1) + lane, ct, 1, 0);
      g1.run(lds_in, bb, acc, [&](const int i) { g2.prologue_step(i); });
#pragma unroll
      for (int j = 8; j < decltype(g2)::PF; ++j) g2.prologue_step(j);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      g1.run(lds_in, bb, acc);
    }
    TOWER_STAMP(4);
    if constexpr (!G::W1_LDS)   // conv2's first weights: ahead of the epilogue + barrier (W1_LDS: requested before conv1's GEMM)
      g2.prologue(p.w2p + lane, wave & 3, 1, G::KSPLIT2 ? 8 * (wave >> 2) : 0);
#pragma unroll
    for (int j = 0; j < G::NT1; ++j) {
      if (pix[j] < 0) continue;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[j][e] * kInputScale + bias[e], 0.f);   // networks.py:115 scale, bias, ReLU
      if constexpr (WRITE_ALL) {
        const int y1 = pix[j] / G::R1W, x1 = pix[j] - y1 * G::R1W;
        *reinterpret_cast<f32x4*>(p.act1 + ((size_t)(b * 20 + y1a + y1) * 20 + x1a + x1) * 32 + 16 * ct + 4 * kq) = v;
      }
      store_split4<G::PL1>(lds_a1 + pix[j] * G::S1 + (16 * ct + 4 * kq) * 2, v);
    }
  }
  TOWER_STAMP(5);
  __syncthreads();
  TOWER_STAMP(6);

  WaveGemm<9, G::NT3, 4, 3, G::PL2, G::PF3, Koff3Half<G>> g3h;     // only the one matching KSPLIT3 is used
  WaveGemm<18, G::NT3, 4, 3, G::PL2, G::PF3, Koff3Full<G>> g3f;
  // ---- phase 2: conv2 (K = 512 = 16 taps x 32 channels) ---------------------------------------------------------------
  {
    const int ct = wave & 3, half = wave >> 2;
    unsigned bb[G::NT2];
    int pix[G::NT2];
#pragma unroll
    for (int j = 0; j < G::NT2; ++j) {
      const int tile = G::KSPLIT2 ? j : half * G::NT2 + j;
      const int pi = 16 * tile + li;
      const bool ok = pi < G::P2;
      pix[j] = ok ? pi : -1;
      const int pc = ok ? pi : 0;
      const int y2 = pc / G::R2W, x2 = pc - y2 * G::R2W;
      bb[j] = (unsigned)(((2 * y2) * G::R1W + 2 * x2) * G::S1 + kq * 16 +
                         (G::KSPLIT2 ? half * (2 * G::R1W * G::S1) : 0));      // K-split: taps 8..15 = kernel rows 2, 3
    }
    f32x4 acc[G::NT2];
#pragma unroll
    for (int j = 0; j < G::NT2; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.b2 + 16 * ct + 4 * kq);
    // conv3's weight fragments are requested while conv2 computes: step j of its prefetch window rides with conv2's k-step
    // j + (NS2 - PF3), into the registers conv2's consumed fragments have freed (the window's head first when it is longer)
    constexpr int NS2 = G::KSPLIT2 ? 8 : 16;
    if constexpr (G::KSPLIT3) {
      constexpr int LEAD = decltype(g3h)::PF - NS2;
      g3h.set(p.w3p + lane, ct, 2, half);
#pragma unroll
      for (int j = 0; j < LEAD; ++j) g3h.prologue_step(j);
      g2.run(lds_a1, bb, acc, [&](const int i) { g3h.prologue_step(i + LEAD); });
    } else {
      constexpr int LEAD = decltype(g3f)::PF - NS2;
      g3f.set(p.w3p + lane, ct, 1, 0);
#pragma unroll
      for (int j = 0; j < LEAD; ++j) g3f.prologue_step(j);
      g2.run(lds_a1, bb, acc, [&](const int i) { g3f.prologue_step(i + LEAD); });
    }
    TOWER_STAMP(7);
    // K-split: the two waves of a channel tile hold partial sums of the same tiles.  Each finishes part of them (half 0 the
    // first OWN0 tiles, half 1 the rest): it parks the partials of the tiles it does NOT own in LDS, and after the barrier adds
    // its partner's to its own (a + b is the same float either way round).
    constexpr int OWN0 = G::KSPLIT2 ? (G::NT2 + 1) / 2 : G::NT2;
    const int j_lo = (G::KSPLIT2 && half == 1) ? OWN0 : 0, j_hi = (G::KSPLIT2 && half == 0) ? OWN0 : G::NT2;
    if constexpr (G::KSPLIT2) {
      f32x4* scr = reinterpret_cast<f32x4*>(lds_s2);
#pragma unroll
      for (int j = 0; j < G::NT2; ++j)
        if (j < j_lo || j >= j_hi) scr[(ct * G::NT2 + j) * 64 + lane] = acc[j];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < G::NT2; ++j)
        if (j >= j_lo && j < j_hi) acc[j] += scr[(ct * G::NT2 + j) * 64 + lane];
    }
#pragma unroll
    for (int j = 0; j < G::NT2; ++j) {
      if (j < j_lo || j >= j_hi || pix[j] < 0) continue;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[j][e] + bias[e], 0.f);
      if constexpr (WRITE_ALL) {
        const int y2 = pix[j] / G::R2W, x2 = pix[j] - y2 * G::R2W;
        *reinterpret_cast<f32x4*>(p.act2 + ((size_t)(b * 9 + y3a + y2) * 9 + x3a + x2) * 64 + 16 * ct + 4 * kq) = v;
      }
      store_split4<G::PL2>(lds_a2 + pix[j] * G::S2 + (16 * ct + 4 * kq) * 2, v);
    }
  }
  __syncthreads();
  TOWER_STAMP(8);

  // ---- phase 3: conv3 (K = 576 = 9 taps x 64 channels) ----------------------------------------------------------------
  {
    const int ct = wave & 3, half = wave >> 2;
    unsigned bb[G::NT3];
    int pix[G::NT3];
#pragma unroll
    for (int j = 0; j < G::NT3; ++j) {
      const int tile = G::KSPLIT3 ? j : half * G::NT3 + j;
      const int pi = 16 * tile + li;
      const bool ok = pi < G::P3;
      pix[j] = ok ? pi : -1;
      const int pc = ok ? pi : 0;
      const int y3 = pc / G::R3W, x3 = pc - y3 * G::R3W;
      bb[j] = (unsigned)((y3 * G::R2W + x3) * G::S2 + kq * 16 + (G::KSPLIT3 ? half * 64 : 0));   // K-split: channels 32..63
    }
    f32x4 acc[G::NT3];
#pragma unroll
    for (int j = 0; j < G::NT3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.b3 + 16 * ct + 4 * kq);
    if constexpr (G::KSPLIT3) g3h.run(lds_a2, bb, acc);
    else g3f.run(lds_a2, bb, acc);
    TOWER_STAMP(9);
    if constexpr (G::KSPLIT3) {
      f32x4* scr = reinterpret_cast<f32x4*>(lds_s3);      // the conv1 planes: dead since the barrier before this phase
      if (half == 1) {
#pragma unroll
        for (int j = 0; j < G::NT3; ++j) scr[(ct * G::NT3 + j) * 64 + lane] = acc[j];
      }
      __syncthreads();
      if (half == 0) {
#pragma unroll
        for (int j = 0; j < G::NT3; ++j) acc[j] += scr[(ct * G::NT3 + j) * 64 + lane];
      }
    }
    if (!G::KSPLIT3 || half == 0) {
      // overlapping regions: the last region along an axis is shifted back to end at 7 and leaves the shared rows / columns
      // to its neighbour
      const int dup_y = y3n - y3a, dup_x = x3n - x3a;
#pragma unroll
      for (int j = 0; j < G::NT3; ++j) {
        if (pix[j] < 0) continue;
        const int y3 = pix[j] / G::R3W, x3 = pix[j] - y3 * G::R3W;
        if (y3 < dup_y || x3 < dup_x) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[j][e] + bias[e], 0.f);
        const int k0 = ((y3a + y3) * 7 + x3a + x3) * 64 + 16 * ct + 4 * kq;      // feature index of v[0] in the flattened row
        if (p.act3_packed)
          *reinterpret_cast<f32x4*>(p.act3 + ((((size_t)(b >> 4) * (3136 / 16) + (k0 >> 4)) * 64 + ((k0 >> 2) & 3) * 16 + (b & 15)) << 2)) = v;
        else
          *reinterpret_cast<f32x4*>(p.act3 + (size_t)b * 3136 + k0) = v;
        if (p.act3_rows) *reinterpret_cast<f32x4*>(p.act3_rows + (size_t)b * 3136 + k0) = v;
      }
    }
  }
  TOWER_STAMP(10);
  TOWER_STAMP(11);
}

}  // namespace paac
