// Conv tower of the reference's DEFAULT architecture (networks.py:138-151, NIPS: conv 8x8 / 4, 16 filters -> conv 4x4 / 2,
// 32 filters), forward only: both layers in ONE launch, conv1's output kept in LDS -- the two-layer sibling of tower.h, same
// arithmetic (u8 pixels exact in bf16, fp32 weights / activations split exactly into three bf16 terms, conv1 the 3 exact
// products, conv2 the 6 leading ones of 9, fp32 accumulation on v_mfma_f32_16x16x32_bf16), same "transposed" contraction
// out^T[Cout, pixels] = W^T[Cout, K] . patches^T[K, pixels] and the same WaveGemm.
//   * a workgroup (8 waves) owns one sample or one of NR overlapping regions of it (R2H x R2W conv2 outputs <- (2 R2 + 2)^2
//     conv1 outputs <- (8 R2 + 12)^2 input pixels);
//   * conv1 has ONE 16-channel tile: its packed weights (24 KB) are staged in LDS once, the eight waves split the pixel
//     tiles; conv2 has two channel tiles x four pixel groups, each wave holding all 8 k-steps of its channel tile's weight
//     fragments in registers (requested before conv1 computes);
//   * conv1 plane in LDS: 16 channels = 32 bytes per pixel and plane, padded to 48: the 16 lanes a ds_read_b128 services
//     together (two adjacent taps x 8 channels per lane, stride-2 walk) then fall on 64 distinct banks;
//   * a k-step of conv2 (32 K values) = two horizontally adjacent taps x 16 channels: lane group kq reads tap kq >> 1,
//     channels 8 (kq & 1) ..+7.
#pragma once
#include "tower.h"

namespace paac {

struct Tower2Args {
  const uint8_t* states;                   // [B,84,84,4]
  const bf16x8* w1p;                        // packed planes, conv1: [8 k-steps][1 channel tile][3 planes][64 lanes]
  const bf16x8* w2p;                        // conv2: [8][2][3][64]
  const float* b1;
  const float* b2;
  float* act1;                             // [B,20,20,16] fp32 (WRITE_ALL only)
  float* act2;                             // [B,9,9,32] fp32 = the fc layer's input rows (HWC flatten, networks.py:6-9), or
  int act2_packed;                         // 1: in fc_heads_kernel's A-fragment order [row tile][K group][lane][4] instead
  float* act2_rows;                        // nullable: a second, plain-row copy (acting rows kept for the update)
  int batch;
};

constexpr int kT2W1Vecs = 8 * 1 * 3 * 64, kT2W2Vecs = 8 * 2 * 3 * 64;   // bf16x8 each
constexpr int kT2PackVecs = kT2W1Vecs + kT2W2Vecs;
constexpr int kT2Flat = 9 * 9 * 32;

static __global__ __launch_bounds__(256) void pack_tower2_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                          bf16x8* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;   // one thread per (k-step, channel tile, lane)
  const float* w;
  int cout, ctiles;
  bf16x8* dst;
  if (i < 8 * 1 * 64) {
    w = w1; cout = 16; ctiles = 1; dst = out;
  } else if (i < 8 * 1 * 64 + 8 * 2 * 64) {
    i -= 8 * 1 * 64;
    w = w2; cout = 32; ctiles = 2; dst = out + kT2W1Vecs;
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

template <int R2H_, int R2W_>
struct Tower2Geom {
  static constexpr int R2H = R2H_, R2W = R2W_;                         // conv2 outputs per region
  static constexpr int R1H = 2 * R2H + 2, R1W = 2 * R2W + 2;           // conv1 outputs it needs (4x4, stride 2)
  static constexpr int RIH = 4 * R1H + 4, RIW = 4 * R1W + 4;           // input pixels (8x8, stride 4)
  static constexpr int NRY = (9 + R2H - 1) / R2H, NRX = (9 + R2W - 1) / R2W, NR = NRY * NRX;   // the last region is shifted back to end at 9
  static constexpr int P1 = R1H * R1W, P2 = R2H * R2W;
  static constexpr int PT1 = (P1 + 15) / 16, PT2 = (P2 + 15) / 16;     // 16-pixel tiles
  static constexpr int S1 = 48;                                        // bytes per pixel in a conv1 LDS plane
  static constexpr int PL1 = P1 * S1;
  static constexpr int NT1 = (PT1 + 7) / 8;                            // conv1: pixel tiles per wave (8 waves, one channel tile)
  static constexpr int NT2 = (PT2 + 3) / 4;                            // conv2: pixel tiles per wave (4 pixel groups x 2 channel tiles)
  static constexpr int IN_BYTES = RIH * RIW * 8;
  static constexpr int IN_AL = (IN_BYTES + 15) / 16 * 16;
  static constexpr int W1_BYTES = kT2W1Vecs * 16;
  static constexpr int LDS_BYTES = IN_AL + 3 * PL1 + W1_BYTES;
  static_assert(RIW % 4 == 0, "input rows are staged 4 pixels (16 bytes) at a time");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class G>
struct Koff2N {   // conv2: k-step i = kernel row i / 2, taps 2 (i % 2) and 2 (i % 2) + 1, 16 channels each
  __device__ static constexpr int at(int i) { return ((i / 2) * G::R1W + 2 * (i % 2)) * G::S1; }
};

template <class G, bool WRITE_ALL>
__global__ __launch_bounds__(512) void tower2_kernel(const Tower2Args p) {
  __shared__ __attribute__((aligned(16))) char lds[G::LDS_BYTES];
  char* const lds_in = lds;                       // bf16 input image [RIH][RIW][4]
  char* const lds_a1 = lds + G::IN_AL;            // conv1 planes [3][P1][S1]
  char* const lds_w1 = lds_a1 + 3 * G::PL1;       // conv1's packed weights

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int b = blockIdx.x / G::NR, reg = blockIdx.x % G::NR;
  const int ry = reg / G::NRX, rx = reg % G::NRX;
  const int y2n = ry * G::R2H, x2n = rx * G::R2W;                   // first conv2 row / column this region owns
  const int y2a = y2n < 9 - G::R2H ? y2n : 9 - G::R2H, x2a = x2n < 9 - G::R2W ? x2n : 9 - G::R2W;   // region origin (conv2 coordinates)
  const int y1a = 2 * y2a, x1a = 2 * x2a;                          // ... in conv1 coordinates

  WaveGemm<8, G::NT1, 1, 1, 0, 2, Koff1<G>> g1;                    // weights from LDS, short read-ahead
  WaveGemm<8, G::NT2, 2, 3, G::PL1, 8, Koff2N<G>> g2;              // all 8 k-steps of the wave's channel tile up front

  // ---- phase 0: input region -> LDS as bf16 (a byte is exact in bf16); conv1's packed weights -> LDS ----------------
  {
    constexpr int VROW = G::RIW / 4, NV = G::RIH * VROW, ITERS = (NV + 511) / 512;
    const uint8_t* src = p.states + (size_t)b * 28224 + ((8 * y2a) * 84 + 8 * x2a) * 4;
    uint4 v[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int i = tid + it * 512;
      const int r = i / VROW, c4 = i - r * VROW;
      if (i < NV) v[it] = *reinterpret_cast<const uint4*>(src + (r * 84 + 4 * c4) * 4);
    }
    constexpr int W1V = kT2W1Vecs / 512;
    bf16x8 w1v[W1V];
#pragma unroll
    for (int it = 0; it < W1V; ++it) w1v[it] = p.w1p[tid + it * 512];
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
#pragma unroll
    for (int it = 0; it < W1V; ++it) reinterpret_cast<bf16x8*>(lds_w1)[tid + it * 512] = w1v[it];
  }
  __syncthreads();

  // ---- phase 1: conv1 (K = 256 = 8 kernel rows x 32) ------------------------------------------------------------------
  {
    unsigned bb[G::NT1];
    int pix[G::NT1];                        // region pixel of this lane in tile j, -1 = none
#pragma unroll
    for (int j = 0; j < G::NT1; ++j) {
      const int tile = wave + 8 * j;
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
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.b1 + 4 * kq);
    // conv2's weight fragments stream in while conv1 computes (conv1's own come from LDS: the registers are free)
    g2.prologue(p.w2p + lane, wave & 1, 1, 0);
    g1.prologue(reinterpret_cast<const bf16x8*>(lds_w1) + lane, 0, 1, 0);
    g1.run(lds_in, bb, acc);
#pragma unroll
    for (int j = 0; j < G::NT1; ++j) {
      if (pix[j] < 0) continue;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[j][e] * kInputScale + bias[e], 0.f);   // networks.py:115 scale, bias, ReLU
      if constexpr (WRITE_ALL) {
        const int y1 = pix[j] / G::R1W, x1 = pix[j] - y1 * G::R1W;
        *reinterpret_cast<f32x4*>(p.act1 + ((size_t)(b * 20 + y1a + y1) * 20 + x1a + x1) * 16 + 4 * kq) = v;
      }
      store_split4<G::PL1>(lds_a1 + pix[j] * G::S1 + (4 * kq) * 2, v);
    }
  }
  __syncthreads();

  // ---- phase 2: conv2 (K = 256 = 8 k-steps of two taps x 16 channels) -------------------------------------------------
  {
    const int ct = wave & 1, grp = wave >> 1;
    unsigned bb[G::NT2];
    int pix[G::NT2];
#pragma unroll
    for (int j = 0; j < G::NT2; ++j) {
      const int tile = grp + 4 * j;
      const int pi = 16 * tile + li;
      const bool ok = (tile < G::PT2) && (pi < G::P2);
      pix[j] = ok ? pi : -1;
      const int pc = ok ? pi : 0;
      const int y2 = pc / G::R2W, x2 = pc - y2 * G::R2W;
      bb[j] = (unsigned)(((2 * y2) * G::R1W + 2 * x2 + (kq >> 1)) * G::S1 + (kq & 1) * 16);
    }
    f32x4 acc[G::NT2];
#pragma unroll
    for (int j = 0; j < G::NT2; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.b2 + 16 * ct + 4 * kq);
    g2.run(lds_a1, bb, acc);
    // overlapping regions: the last region along an axis is shifted back to end at 9 and leaves the shared rows / columns to
    // its neighbour
    const int dup_y = y2n - y2a, dup_x = x2n - x2a;
#pragma unroll
    for (int j = 0; j < G::NT2; ++j) {
      if (pix[j] < 0) continue;
      const int y2 = pix[j] / G::R2W, x2 = pix[j] - y2 * G::R2W;
      if (y2 < dup_y || x2 < dup_x) continue;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[j][e] + bias[e], 0.f);
      const int k0 = ((y2a + y2) * 9 + x2a + x2) * 32 + 16 * ct + 4 * kq;      // feature index of v[0] in the flattened row
      if (p.act2_packed)
        *reinterpret_cast<f32x4*>(p.act2 + ((((size_t)(b >> 4) * (kT2Flat / 16) + (k0 >> 4)) * 64 + ((k0 >> 2) & 3) * 16 + (b & 15)) << 2)) = v;
      else
        *reinterpret_cast<f32x4*>(p.act2 + (size_t)b * kT2Flat + k0) = v;
      if (p.act2_rows) *reinterpret_cast<f32x4*>(p.act2_rows + (size_t)b * kT2Flat + k0) = v;
    }
  }
}

}  // namespace paac
