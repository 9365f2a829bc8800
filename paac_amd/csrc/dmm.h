// Direct-to-register fp32 MFMA contraction for CDNA4 (v_mfma_f32_16x16x4_f32), one template for every
// GEMM-shaped op of the actor-critic network (conv/fc forward, dgrad, wgrad).
//
//   C[M,N] = A[M,K] * B[K,N]
//
// Design (MI355X-first, sized for the tiny batches of PAAC: 32 rows when acting, N*T = 160 when training):
//   * No LDS in the main loop.  Every 64-lane wave owns TM x TN accumulator tiles of 16x16 and loads its own
//     operand fragments from global memory (L2 / Infinity-Cache resident) straight into the MFMA register
//     layout -- lane (i = l&15, kq = l>>4) supplies row/column i at k-slot kq.  The K order inside a 16-wide
//     group is permuted (slot kq, step s  <->  k = 4*kq + s) identically for A and B, which is free for a
//     sum and lets one 16-byte load feed four MFMA steps.
//   * Two fragment patterns per operand:
//       FRAG_K  (source contiguous along K):   one float4 (or uchar4) per tile  = 4 k-steps of that tile
//       FRAG_MN (source contiguous along M/N): one vector per k-step            = V tiles (row/col = V*i + c)
//     forward = (A FRAG_K patches, B FRAG_MN weights), dgrad = (FRAG_K, FRAG_K on W^T taps),
//     wgrad = (FRAG_MN patches^T, FRAG_MN dY).
//   * A is never materialised: patches are gathered through a compile-time geometry (u8 frames for conv1 with
//     the 1/255 scale fused, fp32 NHWC otherwise; zero padding and stride-2 parity classes for dgrad).
//   * Latency is hidden by a register ring PF groups deep and by splitting K over the WK waves of a workgroup
//     (partials summed through LDS once, in the epilogue) and over blockIdx.z (split-K slabs reduced by the
//     consumer) -- that is what fills 256 CUs when M is 32.
//   * Epilogues: bias+ReLU, raw split-K slab (+ bias-gradient row for wgrad), ReLU-mask (dgrad) with the
//     stride-2 parity scatter.
#pragma once
#include <stdlib.h>

#include "common.h"

namespace paac {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int IH_, int IW_, int C_, int OH_, int OW_, int S_, int PH_, int PW_, int KH_, int KW_>
struct Geom {
  static constexpr int IH = IH_, IW = IW_, C = C_, OH = OH_, OW = OW_, S = S_, PH = PH_, PW = PW_, KH = KH_, KW = KW_;
  static constexpr int KWC = KW_ * C_;
  static constexpr int FEATS = KH_ * KW_ * C_;
  static constexpr int OPIX = OH_ * OW_;
  static constexpr bool PADDED = (PH_ != 0) || (PW_ != 0);
};

enum { FRAG_K = 0, FRAG_MN = 1 };
enum { EPI_BIAS_RELU = 0, EPI_SLAB = 1, EPI_MASK = 2, EPI_MASK_PARITY = 3 };

struct GemmArgs {
  const void* A;
  const float* B;
  float* out;
  const float* aux;       // bias (EPI_BIAS_RELU) | activation the ReLU mask is derived from (EPI_MASK*)
  int M, N, K;
  int ldb;                // FRAG_MN B: row stride
  int ldo;
  int groups_per_part;    // 16-wide K groups per (blockIdx.z, wk) part
  int slab_rows;          // EPI_SLAB: rows per slab (M, or M+1 with the bias-gradient row)
  // FRAG_K B (dgrad): element offset of tap t = (th, tw) of parity class z is
  // tap_base[z] + th * tap_sh + tw * tap_sw  (th = t / G::KW, tw = t % G::KW) -- an affine walk over the flipped
  // forward taps, so the K loop needs no table lookup (a scalar load + wait per group).
  int tap_base[4], tap_sh, tap_sw;
  // XCD-aware block -> tile map.  Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8;
  // observed, used for speed only) and every XCD has its own L2, which a kernel boundary leaves cold: an operand
  // slice read by workgroups on k different XCDs is fetched from the Infinity Cache k times.  xcd_dim names the
  // grid dimension (0 = M tiles, 1 = N tiles, 2 = z) whose index is tied to the XCD, so that all workgroups that
  // share the big operand slice of one index run on one XCD; -1 = plain (x fastest) order.
  int MT, NT, Z, xcd_dim;
  unsigned a_bytes, b_bytes;   // extents of the A / B tensors (buffer-resource range check)
#ifdef PAAC_DMM_STAMPS
  unsigned long long* stamps;   // diagnostic build only: 8 x u64 per wave
#endif
};

#ifdef PAAC_DMM_STAMPS
#define DMM_STAMP(i)                                                                          \
  do {                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    if (p.stamps && lane == 0)                                                                \
      p.stamps[((long)bid * (NWM * NWN * WK) + wave) * 8 + (i)] = \
          ((i) == 0 || (i) == 7) ? (unsigned long long)wall_clock64() : (unsigned long long)clock64();                          \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  } while (0)
#else
#define DMM_STAMP(i)
#endif

constexpr float kInputScale = 0.003921568859368563f;  // networks.py:115, float32(1/255)

template <bool U8>
__device__ __forceinline__ f32x4 load4(const void* base, long off) {
  if constexpr (U8) {
    const uchar4 v = *reinterpret_cast<const uchar4*>(static_cast<const uint8_t*>(base) + off);
    return (f32x4){(float)v.x * kInputScale, (float)v.y * kInputScale, (float)v.z * kInputScale,
                   (float)v.w * kInputScale};
  } else {
    return *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + off);
  }
}

// Buffer-resource loads (SRSRC in SGPRs + 32-bit per-lane byte offset + scalar byte offset): no 64-bit address
// arithmetic and no per-load predication in the K loop -- a lane whose row/column is out of range carries the
// sentinel offset and the hardware range check returns zeros.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned kOob = 0x80000000u;   // >= num_records of any tensor here (< 2 GiB)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

template <bool U8>
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if constexpr (U8) {
    const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
    return (f32x4){(float)(v & 255u) * kInputScale, (float)((v >> 8) & 255u) * kInputScale,
                   (float)((v >> 16) & 255u) * kInputScale, (float)(v >> 24) * kInputScale};
  } else {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  }
}

// NOTE (ROCm 7.2 hipcc): __builtin_amdgcn_raw_buffer_load_b64 is lowered to a single buffer_load_dword (the
// second dword is garbage), so 8-byte fragments are two dword buffer loads (same bytes through the texture
// path; a plain global load + select for the range check would put an s_waitcnt right behind every load).
// The scalar offset of a buffer instruction is NOT range-checked, so an operand whose K tail must read as zero
// passes its whole offset in `voff`.
template <int V>
__device__ __forceinline__ f32x4 bloadv(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if constexpr (V == 4) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  } else if constexpr (V == 2) {
    return (f32x4){__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)),
                   __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff + 4u, soff, 0)), 0.f, 0.f};
  } else {
    return (f32x4){__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)), 0.f, 0.f, 0.f};
  }
}

// bf16 pair from the HIGH halves of two fp32 bit patterns (element 0 in the low half): one v_perm_b32.
__device__ __forceinline__ unsigned pack_hi16(unsigned x0, unsigned x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

// Exact three-way bf16 split of 8 fp32 values: x = hi + mid + lo with 8 mantissa bits each (truncate, subtract,
// repeat; every subtraction is exact).  4 VALU ops per value + 1.5 for the packing.
__device__ __forceinline__ void split3_bf16(const float (&x)[8], bf16x8& ph, bf16x8& pm, bf16x8& pl) {
  unsigned xb[8], r1b[8], r2b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    xb[e] = __builtin_bit_cast(unsigned, x[e]);
    const float r1 = x[e] - __builtin_bit_cast(float, xb[e] & 0xFFFF0000u);
    r1b[e] = __builtin_bit_cast(unsigned, r1);
    const float r2 = r1 - __builtin_bit_cast(float, r1b[e] & 0xFFFF0000u);
    r2b[e] = __builtin_bit_cast(unsigned, r2);
  }
  u32x4 hi, mid, lo;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    hi[q] = pack_hi16(xb[2 * q], xb[2 * q + 1]);
    mid[q] = pack_hi16(r1b[2 * q], r1b[2 * q + 1]);
    lo[q] = pack_hi16(r2b[2 * q], r2b[2 * q + 1]);
  }
  ph = __builtin_bit_cast(bf16x8, hi);
  pm = __builtin_bit_cast(bf16x8, mid);
  pl = __builtin_bit_cast(bf16x8, lo);
}

template <int V>
__device__ __forceinline__ f32x4 loadv(const float* p) {
  if constexpr (V == 4) {
    return *reinterpret_cast<const f32x4*>(p);
  } else if constexpr (V == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    return (f32x4){t.x, t.y, 0.f, 0.f};
  } else {
    return (f32x4){p[0], 0.f, 0.f, 0.f};
  }
}

template <int V>
__device__ __forceinline__ void storev(float* p, const f32x4 v) {
  if constexpr (V == 4) {
    *reinterpret_cast<f32x4*>(p) = v;
  } else if constexpr (V == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  } else {
    p[0] = v[0];
  }
}

// AP/BP: fragment pattern of A / B.  TM/TN: 16x16 tiles per wave (for a FRAG_MN side this is also the vector
// width V in {1,2,4}).  NWM x NWN x WK waves per workgroup.  BCO: channels per tap of a FRAG_K B operand.
// The contraction is a device-side body: one workgroup = one call of run(p, workgroup id, LDS).
// XB = 1 ("exact bf16"): for the u8-frame operand of conv1.  A byte is exact in bf16 and an fp32 weight / gradient splits
// EXACTLY into three bf16 terms (8 + 8 + 8 mantissa bits), so x * w = x*w_hi + x*w_mid + x*w_lo with every product
// exact and the sums in fp32 -- the arithmetic of the fp32 MFMA -- on v_mfma_f32_16x16x32_bf16, which retires a 32-deep
// K step in 16 cycles where the fp32 form needs 8 x 32.  The 1/255 input scale moves to the epilogue.  A K stage is
// then two 16-wide groups (the lane's 4 + 4 k values make the 8 the instruction wants; the K permutation is the
// same for A and B, so it is free).
// XB = 2: both operands fp32, each split into (hi, mid, lo) bf16 terms; the six products down to 2^-16 of the leading
// one are accumulated (hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi), the three dropped ones are below 2^-23 of it --
// the size of one fp32 rounding, of which the fp32 MFMA commits one per accumulated product anyway.  Six 16-cycle
// MFMAs replace eight 32-cycle ones per 32-deep K step; the splitting costs VALU work, so it pays for the
// MFMA-bound shapes only (the tuner decides per op).
template <class G_, bool U8, int AP, int BP, int TM, int TN, int NWM, int NWN, int WK, int BCO, int EPI_, bool BIASROW,
          int PF, int XB = 0>
struct Dmm {
  using G = G_;
  static constexpr int EPI = EPI_;
  static constexpr int GS = XB ? 2 : 1;          // 16-wide K groups per pipeline stage
  static constexpr int PRODUCTS = XB == 0 ? 1 : XB == 1 ? 3 : 6;   // MFMA products per fp32 multiply (prof_mix)
  static constexpr int KSTAGE = 16 * GS;
  static_assert(XB != 1 || (U8 && BP == FRAG_MN), "exact-bf16 path: u8 A operand, fp32 FRAG_MN B operand");
  // conv1 forward on the exact path: a K stage is exactly one patch row (8 pixels x 4 channels = 32 contiguous
  // bytes), so lane slot kq takes the instruction's natural k = 8 kq .. 8 kq + 7 = two adjacent pixels: ONE 8-byte
  // load per tile and stage.  Its weights are always in range (N is a multiple of 32, K of 32): plain 8-byte
  // loads for them too -- half the load instructions of the checked dword pairs.
  static constexpr bool NAT = (XB == 1) && (AP == FRAG_K) && !BIASROW && (TN == 2) && (G_::KW * G_::C == 32) &&
                              !((G_::PH != 0) || (G_::PW != 0));
  // A FRAG_MN B operand of width 2 without a bias row is always a weight matrix [K, N] with N a multiple of 32 and K of
  // 16 (conv1, NIPS conv2, the half-width forward tiles): no range check needed, plain 8-byte loads.
  static constexpr bool B_PLAIN = (BP == FRAG_MN) && !BIASROW && (TN == 2) && !NAT;
  static_assert(XB != 2 || !U8, "split path: fp32 operands");
  static constexpr int THREADS = 64 * NWM * NWN * WK;
  static constexpr int M_TILE = NWM * TM * 16, N_TILE = NWN * TN * 16, WAVES_K = WK;
  static constexpr int NA = (AP == FRAG_K) ? TM : 4;
  static constexpr int NB = (BP == FRAG_K) ? TN : 4;
  static constexpr int RING = PF + 1;
  static constexpr int T = TM * TN;
  static constexpr int LDS_F4 = (WK > 1) ? (NWM * NWN * WK * T * 64) : 1;
  static constexpr int RED_F4 = LDS_F4 + (BIASROW ? NWN * WK * 16 : 0);
  static constexpr int EPI_FLOATS = (BP == FRAG_K) ? NWM * NWN * WK * 16 * (TN * 16 + 4) : 4;
  static constexpr int SMEM_BYTES = RED_F4 * 16 + EPI_FLOATS * 4;
  static_assert(AP == FRAG_K || TM == 1 || TM == 2 || TM == 4, "FRAG_MN A: TM is the vector width");
  static_assert(BP == FRAG_K || TN == 1 || TN == 2 || TN == 4, "FRAG_MN B: TN is the vector width");
  static_assert(!BIASROW || (BP == FRAG_MN && NWM == 1), "bias row needs dY as a FRAG_MN B operand and one M-wave");

  __device__ __forceinline__ static void run(const GemmArgs& p, const int bid, char* smem) {
  f32x4* red = reinterpret_cast<f32x4*>(smem);
  float* epi = reinterpret_cast<float*>(smem + RED_F4 * 16);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler: scalar offsets below
  const int wk = wave % WK;
  const int wn = (wave / WK) % NWN;
  const int wm = wave / (WK * NWN);
  const int li = lane & 15;   // row / column inside a tile
  const int kq = lane >> 4;   // k-slot
  int bx, by, bz;
  {
    if (p.xcd_dim < 0) {
      bx = bid % p.MT;
      const int r = bid / p.MT;
      by = r % p.NT;
      bz = r / p.NT;
    } else {
      const int xcd = bid & 7, l = bid >> 3;
      if (p.xcd_dim == 2) {
        const int per = p.MT * p.NT;
        bz = xcd + 8 * (l / per);
        const int r = l % per;
        bx = r % p.MT;
        by = r / p.MT;
      } else if (p.xcd_dim == 1) {
        const int per = p.MT * p.Z;
        by = xcd + 8 * (l / per);
        const int r = l % per;
        bx = r % p.MT;
        bz = r / p.MT;
      } else {
        const int per = p.NT * p.Z;
        bx = xcd + 8 * (l / per);
        const int r = l % per;
        by = r % p.NT;
        bz = r / p.NT;
      }
      if (bx >= p.MT || by >= p.NT || bz >= p.Z) return;   // padding blocks of the 8-way split
    }
  }
  const int z = bz;
  const int m0 = (bx * NWM + wm) * (TM * 16);
  const int n0 = (by * NWN + wn) * (TN * 16);
  const int par = (EPI == EPI_MASK_PARITY) ? z : 0;
  DMM_STAMP(0);
  DMM_STAMP(1);

  // ---- per-lane invariants ------------------------------------------------------------------------
  constexpr unsigned ES = U8 ? 1u : 4u;   // bytes per A element
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B, p.b_bytes);
  // FRAG_K A: TM patch rows fixed for the whole loop -> one 32-bit byte offset per tile (sentinel when the row
  // is out of range); the K group only moves a wave-uniform scalar offset.
  unsigned a_voff[(AP == FRAG_K) ? TM : 1];
  int a_iy0[(AP == FRAG_K) ? TM : 1], a_ix0[(AP == FRAG_K) ? TM : 1];
  // FRAG_MN A (wgrad): the lane's TM consecutive features are fixed, rows move
  int f_kh = 0, f_kwc = 0, f_kw = 0;
  bool f_ok = false;
  if constexpr (AP == FRAG_K) {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int row = m0 + t * 16 + li;
      const bool ok = row < p.M;
      const int r = ok ? row : 0;
      const int b = r / G::OPIX;
      const int rem = r - b * G::OPIX;
      const int oy = rem / G::OW;
      const int ox = rem - oy * G::OW;
      a_iy0[t] = oy * G::S - G::PH;
      a_ix0[t] = ox * G::S - G::PW;
      const int base = ((b * G::IH + a_iy0[t]) * G::IW + a_ix0[t]) * G::C + (NAT ? 8 : 4) * kq;   // elements; may be < 0 when padded
      a_voff[t] = ok ? (unsigned)base * ES : (NAT ? 0u : kOob);   // NAT: plain loads -- an out-of-range row reads row 0, its result is never stored
      if constexpr (G::PADDED) a_iy0[t] = ok ? a_iy0[t] : -(1 << 20);   // padded path re-derives validity per group
    }
  } else {
    const int f = m0 + TM * li;
    f_ok = f < p.M;
    f_kh = f / G::KWC;
    f_kwc = f - f_kh * G::KWC;
    f_kw = f_kwc / G::C;
  }
  // FRAG_K B (dgrad): TN weight rows (input channels of the forward conv) fixed
  unsigned b_voff[(BP == FRAG_K) ? TN : 4];
  if constexpr (BP == FRAG_K) {
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int n = n0 + t * 16 + li;
      b_voff[t] = (n < p.N) ? (unsigned)(n * BCO + 4 * kq) * 4u : kOob;
    }
  } else {
    // FRAG_MN B: row 4*kq + s of the group, TN consecutive columns
    const bool ok = n0 + TN * li < p.N;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      b_voff[s] = ok ? (unsigned)(((NAT ? 8 : 4) * kq + s) * p.ldb + n0 + TN * li) * 4u : ((NAT || B_PLAIN) ? 0u : kOob);
  }

  // ---- K range of this wave -------------------------------------------------------------------------
  const int ngroups = (p.K + KSTAGE - 1) / KSTAGE;   // pipeline stages of KSTAGE k values
  const int part = ((EPI == EPI_SLAB) ? z : 0) * WK + wk;
  const int g_begin = min(part * p.groups_per_part, ngroups);
  const int g_end = min(g_begin + p.groups_per_part, ngroups);
  const int ng = g_end - g_begin;
  const int k_end = min(p.K, g_end * KSTAGE);   // FRAG_MN A rows at or past this read as zero
  const int tap_base = (BP == FRAG_K) ? p.tap_base[par] : 0;

  // ---- fragment loads for K group `g` (k16 = 16 g) --------------------------------------------------
  // Every call issues the same NA + NB buffer loads whatever `live` is -- a group past the end of this wave's
  // range is loaded with all lanes out of range (no memory access, zeros, never consumed) -- so the K loop has
  // no branch around its loads and the compiler's s_waitcnt placement keeps the full prefetch distance.
  f32x4 fa[RING][GS][NA], fb[RING][GS][NB];
  unsigned ua[RING][GS][NA];                    // XB: the raw bytes of the A operand (4 per lane and load)
  auto load_group = [&](int slot, int g, bool live) {
    const unsigned kill = live ? 0u : kOob;     // wave-uniform
#pragma unroll
    for (int gs = 0; gs < GS; ++gs) {
    const int k16 = (g * GS + gs) * 16;         // wave-uniform
    if constexpr (AP == FRAG_K) {
      const int kh = k16 / G::KWC;
      const int kwc = k16 - kh * G::KWC;
      const unsigned goff = live ? (unsigned)(kh * (G::IW * G::C) + kwc) * ES : 0u;
      if constexpr (!G::PADDED) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          if constexpr (NAT) {
            if (gs == 0) {   // both dwords of the stage at once (the stage is one patch row: goff of gs 0)
              const uint2 w = *reinterpret_cast<const uint2*>(static_cast<const char*>(p.A) + a_voff[t] + goff);
              ua[slot][0][t] = w.x;
              ua[slot][1][t] = w.y;
            }
          } else if constexpr (XB == 1) {
            ua[slot][gs][t] = __builtin_amdgcn_raw_buffer_load_b32(rsA, a_voff[t] | kill, goff, 0);
          } else {
            fa[slot][gs][t] = bload4<U8>(rsA, a_voff[t] | kill, goff);
          }
        }
      } else {
        const int kw = kwc / G::C;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int iy = a_iy0[t] + kh;
          const int ix = a_ix0[t] + kw;
          const bool ok = live && (iy >= 0) && (iy < G::IH) && (ix >= 0) && (ix < G::IW);
          fa[slot][gs][t] = bload4<U8>(rsA, ok ? a_voff[t] + goff : kOob, 0);
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int r = k16 + 4 * kq + s;
        unsigned off = kOob;
        if (f_ok && r < k_end) {
          const int b = r / G::OPIX;
          const int rem = r - b * G::OPIX;
          const int oy = rem / G::OW;
          const int ox = rem - oy * G::OW;
          const int iy = oy * G::S - G::PH + f_kh;
          const int ix0 = ox * G::S - G::PW;
          bool ok = true;
          if constexpr (G::PADDED) ok = (iy >= 0) && (iy < G::IH) && (ix0 + f_kw >= 0) && (ix0 + f_kw < G::IW);
          if (ok) off = (unsigned)(((b * G::IH + iy) * G::IW + ix0) * G::C + f_kwc) * ES;
        }
        if constexpr (U8) {
          static_assert(!U8 || TM == 4, "u8 FRAG_MN loads are uchar4");
          if constexpr (XB == 1) ua[slot][gs][s] = __builtin_amdgcn_raw_buffer_load_b32(rsA, off, 0, 0);
          else fa[slot][gs][s] = bload4<true>(rsA, off, 0);
        } else {
          fa[slot][gs][s] = bloadv<TM>(rsA, off, 0);
        }
      }
    }
    if constexpr (BP == FRAG_K) {
      const int tap = k16 / BCO;
      const int co = k16 - tap * BCO;
      const int th = tap / G::KW;
      const int tw = tap - th * G::KW;
      const unsigned goff = live ? (unsigned)(tap_base + th * p.tap_sh + tw * p.tap_sw + co) * 4u : 0u;
#pragma unroll
      for (int t = 0; t < TN; ++t) fb[slot][gs][t] = bload4<false>(rsB, b_voff[t] | kill, goff);
    } else {
      // the whole offset goes through the range-checked VGPR offset: rows >= K (the tail of a K that is not a
      // multiple of 16) and dead groups read as zero
      if constexpr (NAT) {
        // rows 32 g + 8 kq + 4 gs + s; a dead stage re-reads stage 0 (valid memory, never consumed)
        const unsigned goff = live ? (unsigned)((g * 32 + 4 * gs) * p.ldb) * 4u : 0u;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float2 t = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(p.B) + b_voff[s] + goff);
          fb[slot][gs][s] = (f32x4){t.x, t.y, 0.f, 0.f};
        }
      } else if constexpr (B_PLAIN) {
        const unsigned goff = live ? (unsigned)(k16 * p.ldb) * 4u : 0u;   // a dead group re-reads group 0: never consumed
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float2 t = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(p.B) + b_voff[s] + goff);
          fb[slot][gs][s] = (f32x4){t.x, t.y, 0.f, 0.f};
        }
      } else {
        const unsigned goff = live ? (unsigned)(k16 * p.ldb) * 4u : kOob;
#pragma unroll
        for (int s = 0; s < 4; ++s) fb[slot][gs][s] = bloadv<TN>(rsB, b_voff[s] + goff, 0);
      }
    }
    }   // gs
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};

  // bias of this lane's TN output columns: issued ahead of the K loop, consumed in the epilogue
  f32x4 biasv = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == EPI_BIAS_RELU) {
    if (n0 + TN * li < p.N) biasv = loadv<TN>(p.aux + n0 + TN * li);
  }

  DMM_STAMP(2);
#pragma unroll
  for (int s = 0; s < PF; ++s) load_group(s, g_begin + s, s < ng);
  DMM_STAMP(3);
  for (int gb = 0;; gb += RING) {
#pragma unroll
    for (int st = 0; st < RING; ++st) {
      const int gi = gb + st;
      if (gi >= ng) goto k_done;   // side exit: nothing joins the loop body, the wait counts stay exact
      load_group((st + PF) % RING, g_begin + gi + PF, gi + PF < ng);
      if constexpr (XB == 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const float av = (AP == FRAG_K) ? fa[st][0][i][s] : fa[st][0][s][i];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const float bv = (BP == FRAG_K) ? fb[st][0][j][s] : fb[st][0][s][j];
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i][j], 0, 0, 0);
            }
          }
          if constexpr (BIASROW) bsum += fb[st][0][s];
        }
      } else if constexpr (XB == 2) {
        // three bf16 planes per operand tile; element e = 4 gs + s of a lane's 8 is k = 4kq + s of group gs
        bf16x8 ah[TM], am[TM], al[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float x[8];
#pragma unroll
          for (int gs = 0; gs < 2; ++gs)
#pragma unroll
            for (int s = 0; s < 4; ++s) x[4 * gs + s] = (AP == FRAG_K) ? fa[st][gs][i][s] : fa[st][gs][s][i];
          split3_bf16(x, ah[i], am[i], al[i]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float x[8];
#pragma unroll
          for (int gs = 0; gs < 2; ++gs)
#pragma unroll
            for (int s = 0; s < 4; ++s) x[4 * gs + s] = (BP == FRAG_K) ? fb[st][gs][j][s] : fb[st][gs][s][j];
          bf16x8 bh, bm, bl;
          split3_bf16(x, bh, bm, bl);
#pragma unroll
          for (int i = 0; i < TM; ++i) {   // smallest terms first
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bm, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bm, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
          }
        }
        if constexpr (BIASROW) {
#pragma unroll
          for (int gs = 0; gs < 2; ++gs)
#pragma unroll
            for (int s = 0; s < 4; ++s) bsum += fb[st][gs][s];
        }
      } else {
        // lane (i, kq) holds k = 4kq + s of both groups of the stage: element e = 4 gs + s of the 8 the MFMA takes
        bf16x8 a8[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          u32x4 w;
#pragma unroll
          for (int gs = 0; gs < 2; ++gs) {
            unsigned f[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              // FRAG_K: the dword of tile i carries k = 4kq .. 4kq+3; FRAG_MN: dword s carries the 4 tiles of k = 4kq+s
              const unsigned byte = (AP == FRAG_K) ? (ua[st][gs][i] >> (8 * s)) & 255u : (ua[st][gs][s] >> (8 * i)) & 255u;
              f[s] = __builtin_bit_cast(unsigned, (float)byte);          // v_cvt_f32_ubyteN; an integer < 256 is exact in bf16
            }
            w[2 * gs] = pack_hi16(f[0], f[1]);
            w[2 * gs + 1] = pack_hi16(f[2], f[3]);
          }
          a8[i] = __builtin_bit_cast(bf16x8, w);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float x[8];
#pragma unroll
          for (int gs = 0; gs < 2; ++gs)
#pragma unroll
            for (int s = 0; s < 4; ++s) x[4 * gs + s] = fb[st][gs][s][j];
          bf16x8 bh, bm, bl;
          split3_bf16(x, bh, bm, bl);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[i], bl, acc[i][j], 0, 0, 0);   // small terms first
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[i], bm, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[i], bh, acc[i][j], 0, 0, 0);
          }
        }
        if constexpr (BIASROW) {
#pragma unroll
          for (int gs = 0; gs < 2; ++gs)
#pragma unroll
            for (int s = 0; s < 4; ++s) bsum += fb[st][gs][s];
        }
      }
    }
  }
k_done:

  DMM_STAMP(4);
  // ---- dgrad: fetch the activations the ReLU mask is derived from now (the ring registers are dead), so the
  // loads overlap the LDS reduce / transpose instead of sitting exposed between the transpose and the store
  constexpr int F4_PER_ROW = TN * 4;
  constexpr int ROWS_PER_PASS = 64 / F4_PER_ROW;
  constexpr int PASSES = (BP == FRAG_K) ? 16 / ROWS_PER_PASS : 1;
  f32x4 mk[(BP == FRAG_K) ? TM : 1][PASSES];
  int mk_off[(BP == FRAG_K) ? TM : 1][PASSES];   // element offset of the lane's float4 in out / aux, -1 = none
  if constexpr (BP == FRAG_K) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      if ((tm % WK) != wk) continue;
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        const int row = pass * ROWS_PER_PASS + lane / F4_PER_ROW;
        const int m = m0 + tm * 16 + row;
        const int n = n0 + (lane % F4_PER_ROW) * 4;
        int off = -1;
        if (m < p.M && n < p.N) {
          int orow = m;
          if constexpr (EPI == EPI_MASK_PARITY) {
            const int b = m / G::OPIX;
            const int rem = m - b * G::OPIX;
            const int a = rem / G::OW;
            const int cc = rem - a * G::OW;
            orow = (b * (2 * G::OH) + 2 * a + (z >> 1)) * (2 * G::OW) + 2 * cc + (z & 1);
          }
          off = orow * p.ldo + n;
          mk[tm][pass] = *reinterpret_cast<const f32x4*>(p.aux + off);
        }
        mk_off[tm][pass] = off;
      }
    }
  }

  // ---- sum the WK partials through LDS ----------------------------------------------------------------
  const int grp = wm * NWN + wn;
  if constexpr (WK > 1) {
#pragma unroll
    for (int t = 0; t < T; ++t) red[((grp * WK + wk) * T + t) * 64 + lane] = acc[t / TN][t % TN];
    __syncthreads();
  }

  DMM_STAMP(5);
  // ---- epilogue: unit = one row of tiles (tm), handled by wave wk == tm % WK ----------------------------
  const int q = kq;  // D layout: row = 4*q + r, col = li
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    if ((tm % WK) != wk) continue;
    f32x4 c[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      if constexpr (WK > 1) {
        f32x4 v = red[((grp * WK + 0) * T + tm * TN + tn) * 64 + lane];
#pragma unroll
        for (int w = 1; w < WK; ++w) v += red[((grp * WK + w) * T + tm * TN + tn) * 64 + lane];
        c[tn] = v;
      } else {
        c[tn] = acc[tm][tn];
      }
      if constexpr (XB == 1) c[tn] *= kInputScale;   // the products were taken on the raw bytes (networks.py:115)
    }
    if constexpr (BP == FRAG_MN) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // global row of D element (tile tm, row 4q+r)
        const int m = (AP == FRAG_K) ? (m0 + tm * 16 + 4 * q + r) : (m0 + TM * (4 * q + r) + tm);
        if (m >= p.M) continue;
        const int n = n0 + TN * li;
        if (n >= p.N) continue;
        f32x4 v;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) v[tn] = c[tn][r];
        if constexpr (EPI == EPI_BIAS_RELU) {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) v[tn] = fmaxf(v[tn] + biasv[tn], 0.f);
          storev<TN>(p.out + (long)m * p.ldo + n, v);
        } else {
          static_assert(BP != FRAG_MN || EPI == EPI_BIAS_RELU || EPI == EPI_SLAB, "FRAG_MN B epilogues");
          storev<TN>(p.out + ((long)z * p.slab_rows + m) * p.ldo + n, v);
        }
      }
    } else {
      // dgrad: the D layout has the output column on the lane; transpose the unit through a wave-private LDS tile
      // so that each lane owns 4 consecutive columns of a row: one float4 mask load + one float4 store per lane
      // and pass instead of 4*TN scalar pairs.
      static_assert(BP == FRAG_MN || AP == FRAG_K, "dgrad epilogue assumes FRAG_K A");
      constexpr int EW = TN * 16 + 4;
      float* et = epi + wave * (16 * EW);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) et[(4 * q + r) * EW + tn * 16 + li] = c[tn][r];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        const int row = pass * ROWS_PER_PASS + lane / F4_PER_ROW;
        const int c4 = (lane % F4_PER_ROW) * 4;
        const int off = mk_off[tm][pass];
        if (off >= 0) {
          const f32x4 act = mk[tm][pass];
          f32x4 v = *reinterpret_cast<const f32x4*>(et + row * EW + c4);
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) v[e4] = act[e4] > 0.f ? v[e4] : 0.f;
          *reinterpret_cast<f32x4*>(p.out + off) = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }

  DMM_STAMP(6);
  DMM_STAMP(7);
  // ---- bias-gradient row: column sums of dY over this part's K range ------------------------------------
  if constexpr (BIASROW) {
    if (bx == 0 && wm == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = bsum[c];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        bsum[c] = v;
      }
      f32x4* bred = red + LDS_F4;
      if constexpr (WK > 1) {
        if (kq == 0) bred[(wn * WK + wk) * 16 + li] = bsum;
        __syncthreads();
        if (wk == 0 && kq == 0) {
          f32x4 v = bred[(wn * WK) * 16 + li];
#pragma unroll
          for (int w = 1; w < WK; ++w) v += bred[(wn * WK + w) * 16 + li];
          bsum = v;
        }
      }
      if (wk == 0 && kq == 0 && n0 + TN * li < p.N)
        storev<TN>(p.out + ((long)z * p.slab_rows + p.M) * p.ldo + n0 + TN * li, bsum);
    }
  }
  }   // run
};

template <class D>
__global__ __launch_bounds__(D::THREADS) void dmm_kernel(const GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[D::SMEM_BYTES];
  D::run(p, blockIdx.x, smem);
}

// Fills the launch geometry of `a` for body D; returns the number of workgroups.
template <class D>
inline long prepare_dmm(GemmArgs& a, int zdim, int ksplit_z, int xcd_dim) {
  prof_mix(D::PRODUCTS);
  const int ngroups = (a.K + D::KSTAGE - 1) / D::KSTAGE;
  const int parts = D::WAVES_K * ((D::EPI == EPI_SLAB) ? ksplit_z : 1);
  a.groups_per_part = (ngroups + parts - 1) / parts;
  a.MT = (a.M + D::M_TILE - 1) / D::M_TILE;
  a.NT = (a.N + D::N_TILE - 1) / D::N_TILE;
  a.Z = zdim;
  static const int xcd_mask = []() { const char* v = getenv("PAAC_TUNE_XCD"); return (v && *v) ? atoi(v) : 7; }();
  a.xcd_dim = (xcd_dim >= 0 && ((xcd_mask >> xcd_dim) & 1)) ? xcd_dim : -1;   // tuning knob: bit d enables dim d
  xcd_dim = a.xcd_dim;
  long blocks = (long)a.MT * a.NT * a.Z;
  if (xcd_dim == 0) blocks = (long)((a.MT + 7) / 8) * 8 * a.NT * a.Z;
  if (xcd_dim == 1) blocks = (long)((a.NT + 7) / 8) * 8 * a.MT * a.Z;
  if (xcd_dim == 2) blocks = (long)((a.Z + 7) / 8) * 8 * a.MT * a.NT;
  return blocks;
}

// Two independent contractions in ONE launch: workgroups [0, first1) run body D0 on g0, the rest body D1 on g1 (first1 is
// a multiple of 8, so each body's XCD-tied tile map still sees its own index modulo 8).  Pays only when a workgroup of
// each kind fits a CU together (LDS and registers) and all of them are resident at once; otherwise the second body waits
// for the first and the launch takes the sum of the two.  (Bodies of different sizes: the launch has the larger size and
// the smaller body's surplus waves end at once.)
struct PairArgs {
  GemmArgs g0, g1;
  int first1, count0;
};
template <class D0, class D1>
__global__ __launch_bounds__((D0::THREADS > D1::THREADS ? D0::THREADS : D1::THREADS)) void dmm_pair_kernel(const PairArgs p) {
  constexpr int SMEM = D0::SMEM_BYTES > D1::SMEM_BYTES ? D0::SMEM_BYTES : D1::SMEM_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  const int bid = blockIdx.x;
  if (bid < p.first1) {
    if (bid < p.count0 && (int)threadIdx.x < D0::THREADS) D0::run(p.g0, bid, smem);
  } else {
    if ((int)threadIdx.x < D1::THREADS) D1::run(p.g1, bid - p.first1, smem);
  }
}

template <class D>
inline void launch_dmm(GemmArgs a, int zdim, int ksplit_z, int xcd_dim, hipStream_t s) {
  const long blocks = prepare_dmm<D>(a, zdim, ksplit_z, xcd_dim);
  launch_k(dmm_kernel<D>, dim3((unsigned)blocks), dim3(D::THREADS), s, PROF_WHOLE, a);
}

}  // namespace paac
