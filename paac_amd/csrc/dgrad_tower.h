// Backward counterpart of the conv tower (tower.h) for the Nature network: the data gradients of conv3 and conv2 in ONE
// launch, one workgroup per sample, the intermediate gradient kept in LDS as three bf16 planes.
//
//   da2 = corr_full(da3, W3)  * 1[a2 > 0]        (the transpose of conv3: 3x3, stride 1)
//   da1 = corr_T_s2(da2, W2)  * 1[a1 > 0]        (the transpose of conv2: 4x4, stride 2)
//
// Both are rewritten as stride-1 VALID convolutions over zero-padded gradient images, so they run through the same
// WaveGemm as the forward tower (out^T[C, pixels] = W'[C, K] * patches^T[K, pixels], exact 3-way bf16 splits, 6 products):
//   * conv3: da3 padded by 2 (11 x 11 x 64); output pixel (y2, x2) reads taps (a, b) of the padded image at
//     (y2 + a, x2 + b) against W3[2 - a, 2 - b, c2, c3]: K = 9 taps x 64 = 576, 81 output pixels, 64 output channels.
//   * conv2: by output parity class (py, px) = (y1 & 1, x1 & 1): y1 = 2 u + py only meets kernel rows kh = py + 2 (1 - a),
//     a in {0, 1}, reading da2 padded by 1 (11 x 11 x 64) at (u + a, v + b): a 2 x 2 stride-1 convolution per class,
//     K = 4 taps x 64 = 256, 100 output pixels per class, 32 output channels -- no zero-stuffed work.
//     A wave owns one (class, 16-channel tile): its weight slice is read by no other wave.
// The masked fp32 gradients also go to HBM: the weight-gradient kernels of conv2 / conv1 read them.
// Weights: W3d / W2d planes pre-split and pre-arranged like the forward ones (pack_dgrad_kernel; the optimizer step
// rewrites them with the forward planes).
#pragma once
#include "tower.h"

namespace paac {

struct DgradTowerArgs {
  const float* da3;       // [B,7,7,64]  gradient wrt conv3's output (already masked by the fc data-gradient kernel)
  const float* act2;      // [B,9,9,64]  forward activations (ReLU masks)
  const float* act1;      // [B,20,20,32]
  const bf16x8* w3d;      // [18 k-steps][4 tiles][3 planes][64 lanes]
  const bf16x8* w2d;      // [4 classes][8 k-steps][2 tiles][3 planes][64 lanes]
  float* da2;             // [B,9,9,64]
  float* da1;             // [B,20,20,32]
  int batch;
};

// W3d[(a, b, c3)][c2] = W3[2 - a, 2 - b, c2, c3];  W2d[cls = (py, px)][(a, b, c2)][c1] = W2[py + 2 (1 - a), px + 2 (1 - b), c1, c2]
// (forward weights are HWIO: W[kh, kw, cin, cout]).  One thread per (unit, lane) like pack_tower_kernel.
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const float* __restrict__ w2, const float* __restrict__ w3,
                                                         bf16x8* __restrict__ out3, bf16x8* __restrict__ out2) {
  int i = blockIdx.x * 256 + threadIdx.x;
  float x[8];
  if (i < 18 * 4 * 64) {
    const int lane = i & 63, unit = i >> 6;          // unit = s * 4 + ct
    const int ct = unit & 3, s = unit >> 2;
    const int tap = s >> 1, a = tap / 3, b = tap - 3 * a;
    const int c2 = 16 * ct + (lane & 15), c3_0 = 32 * (s & 1) + 8 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = w3[((long)((2 - a) * 3 + (2 - b)) * 64 + c2) * 64 + c3_0 + j];
    split3_store(x, out3 + (long)unit * 192 + lane);
    return;
  }
  i -= 18 * 4 * 64;
  if (i < 4 * 8 * 2 * 64) {
    const int lane = i & 63, unit = i >> 6;          // unit = (cls * 8 + s) * 2 + ct
    const int ct = unit & 1, s = (unit >> 1) & 7, cls = unit >> 4;
    const int py = cls >> 1, px = cls & 1, tap = s >> 1, a = tap >> 1, b = tap & 1;
    const int c1 = 16 * ct + (lane & 15), c2_0 = 32 * (s & 1) + 8 * (lane >> 4);
    const int kh = py + 2 * (1 - a), kw = px + 2 * (1 - b);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = w2[((long)(kh * 4 + kw) * 32 + c1) * 64 + c2_0 + j];
    split3_store(x, out2 + (long)unit * 192 + lane);
  }
}

struct DgGeom {
  static constexpr int PW = 11, PS = 160;                 // padded image width, bytes per pixel in a plane
  static constexpr int PLANE = PW * PW * PS;              // 19,360 B
  static constexpr int LDS_BYTES = 6 * PLANE;             // da3 planes + da2 planes
};
struct KoffD3 {   // conv3 data gradient: k-step i = (tap i / 2 = (a, b) of 3 x 3, channel half i % 2)
  __device__ static constexpr int at(int i) { return (((i / 2) / 3) * DgGeom::PW + ((i / 2) % 3)) * DgGeom::PS + (i % 2) * 64; }
};
struct KoffD2 {   // conv2 data gradient, one parity class: k-step i = (tap i / 2 = (a, b) of 2 x 2, channel half i % 2)
  __device__ static constexpr int at(int i) { return (((i / 2) / 2) * DgGeom::PW + ((i / 2) % 2)) * DgGeom::PS + (i % 2) * 64; }
};

// hg.blocks > 0: the FIRST hg.blocks workgroups are not samples but the row reductions of the head gradients
// (heads.h: heads_param_grads -- head weight / bias gradients and the loss scalars from the per-row gradients the fused
// training-heads launch left behind).  They wait on nothing this kernel produces and run on CUs the 160 sample
// workgroups leave idle: a place to be, not a dependency.  (First, not last: with more samples than CUs the last
// workgroups of the grid start when everything else is done, and these are long.)
__global__ __launch_bounds__(512) void dgrad_tower_kernel(const DgradTowerArgs p, const HeadsGradArgs hg) {
  using G = DgGeom;
  __shared__ __attribute__((aligned(16))) char lds[G::LDS_BYTES];
  if ((int)blockIdx.x < hg.blocks) {
    if (threadIdx.x >= 256) return;
    static_assert(G::LDS_BYTES >= (32 + 4) * 264 * 4 + 64 * 4, "head-gradient scratch");
    float* smem = reinterpret_cast<float*>(lds);
    const int role = (int)blockIdx.x;
    if (hg.A <= 4) heads_param_grads<512, 4>(role, hg.h, hg.dl_buf, hg.A, hg.B, hg.gWa, hg.gba, hg.gWc, hg.gbc, hg.loss_out, smem);
    else if (hg.A <= 8) heads_param_grads<512, 8>(role, hg.h, hg.dl_buf, hg.A, hg.B, hg.gWa, hg.gba, hg.gWc, hg.gbc, hg.loss_out, smem);
    else if (hg.A <= 20) heads_param_grads<512, 20>(role, hg.h, hg.dl_buf, hg.A, hg.B, hg.gWa, hg.gba, hg.gWc, hg.gbc, hg.loss_out, smem);
    else heads_param_grads<512, 32>(role, hg.h, hg.dl_buf, hg.A, hg.B, hg.gWa, hg.gba, hg.gWc, hg.gbc, hg.loss_out, smem);
    return;
  }
  char* const lds_d3 = lds;                 // da3, zero-padded by 2: [3 planes][11 x 11][160 B]
  char* const lds_d2 = lds + 3 * G::PLANE;  // da2, zero-padded by 1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int b = (int)blockIdx.x - hg.blocks;

  // ---- phase 0: da3 of this sample -> padded bf16 planes; the borders (and all of the da2 image) start as zeros ----------
  WaveGemm<18, 3, 4, 3, G::PLANE, 6, KoffD3> g3;
  WaveGemm<8, 7, 2, 3, G::PLANE, 8, KoffD2, false> g2;
  {
    constexpr int NV = 49 * 16;             // float4 of the 7 x 7 x 64 gradient
    f32x4 v[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = tid + it * 512;
      if (i < NV) v[it] = *reinterpret_cast<const f32x4*>(p.da3 + (size_t)b * 3136 + 4 * i);
    }
    g3.prologue(p.w3d + lane, wave & 3, 1, 0);
    const u32x4 z = (u32x4){0u, 0u, 0u, 0u};
    for (int i = tid; i < G::LDS_BYTES / 16; i += 512) reinterpret_cast<u32x4*>(lds)[i] = z;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = tid + it * 512;
      if (i < NV) {
        const int pix = i >> 4, c0 = 4 * (i & 15);
        const int y3 = pix / 7, x3 = pix - 7 * y3;
        store_split4<G::PLANE>(lds_d3 + ((y3 + 2) * G::PW + x3 + 2) * G::PS + c0 * 2, v[it]);
      }
    }
  }
  __syncthreads();

  // ---- phase 1: conv3 data gradient: 81 pixels (6 tiles, 3 per wave half) x 4 channel tiles x 18 k-steps ----------------
  {
    const int ct = wave & 3, half = wave >> 2;
    unsigned bb[3];
    int pix[3];
    f32x4 act[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int pi = 16 * (3 * half + j) + li;
      const bool ok = pi < 81;
      pix[j] = ok ? pi : -1;
      const int pc = ok ? pi : 0;
      const int y2 = pc / 9, x2 = pc - 9 * y2;
      bb[j] = (unsigned)((y2 * G::PW + x2) * G::PS + kq * 16);
      // the forward activations the ReLU mask is read from: requested now, consumed in the epilogue
      act[j] = *reinterpret_cast<const f32x4*>(p.act2 + ((size_t)b * 81 + pc) * 64 + 16 * ct + 4 * kq);
    }
    f32x4 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // conv2's weight fragments (one (class, tile) slice per wave) ride with conv3's last k-steps
    g2.set(p.w2d + (size_t)(wave >> 1) * (8 * 2 * 192) + lane, wave & 1, 1, 0);
    g3.run(lds_d3, bb, acc, [&](const int i) { g2.prologue_step(i - (18 - decltype(g2)::PF)); });
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (pix[j] < 0) continue;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act[j][e] > 0.f ? acc[j][e] : 0.f;
      *reinterpret_cast<f32x4*>(p.da2 + ((size_t)b * 81 + pix[j]) * 64 + 16 * ct + 4 * kq) = v;
      const int y2 = pix[j] / 9, x2 = pix[j] - 9 * y2;
      store_split4<G::PLANE>(lds_d2 + ((y2 + 1) * G::PW + x2 + 1) * G::PS + (16 * ct + 4 * kq) * 2, v);
    }
  }
  __syncthreads();

  // ---- phase 2: conv2 data gradient: wave = (parity class, channel tile); 100 pixels (7 tiles) x 8 k-steps ----------------
  {
    const int ct = wave & 1, cls = wave >> 1, py = cls >> 1, px = cls & 1;
    unsigned bb[7];
    int opix[7];                 // output pixel in the 20 x 20 image, -1 = none
    f32x4 a1[7];                 // forward activations the ReLU mask is read from: requested now, consumed in the epilogue
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int pi = 16 * j + li;
      const bool ok = pi < 100;
      const int pc = ok ? pi : 0;
      const int u = pc / 10, v = pc - 10 * u;
      bb[j] = (unsigned)((u * G::PW + v) * G::PS + kq * 16);
      const int op = (2 * u + py) * 20 + 2 * v + px;
      opix[j] = ok ? op : -1;
      a1[j] = *reinterpret_cast<const f32x4*>(p.act1 + ((size_t)b * 400 + op) * 32 + 16 * ct + 4 * kq);
    }
    f32x4 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    g2.run(lds_d2, bb, acc);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      if (opix[j] < 0) continue;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = a1[j][e] > 0.f ? acc[j][e] : 0.f;
      *reinterpret_cast<f32x4*>(p.da1 + ((size_t)b * 400 + opix[j]) * 32 + 16 * ct + 4 * kq) = v;
    }
  }
}

}  // namespace paac
