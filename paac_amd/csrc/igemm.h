// fp32 implicit-GEMM on CDNA4 matrix cores (v_mfma_f32_16x16x4_f32), one template for every
// contraction of the actor-critic network: conv forward, conv dgrad (gather form, stride-2 by output
// parity), conv/fc wgrad (split-K slabs + bias row), fc forward (split-K) and fc dgrad.
//
//   C[M,N] = A[M,K] * B[K,N]
//
// A is never materialised: it is gathered from an NHWC tensor (u8 frames for conv1, fp32 otherwise)
// through a compile-time patch geometry `G`.  64-wide wavefronts: a 256-thread workgroup = 4 waves
// arranged WM x WN x WK (WK = intra-block split of each K chunk, reduced through LDS in the epilogue).
// Operands are staged global -> registers -> LDS with the next chunk's global loads in flight during
// the MFMAs of the current one; LDS strides are chosen so the per-MFMA ds_read_b32 operand fetches
// are bank-conflict free (row-major tiles: stride % 32 == 18, k-major tiles: stride % 32 == 16).
#pragma once
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

enum { A_ROWS_M = 0, A_ROWS_K = 1 };           // patch rows are GEMM-M (fwd, dgrad) or GEMM-K (wgrad)
enum { B_KN = 0, B_NK_TAPS = 1 };              // B[k*ldb+n]  or  B[tapoff[k/CO] + n*CO + k%CO]
enum { EPI_BIAS_RELU = 0, EPI_SLAB = 1, EPI_MASK = 2, EPI_MASK_PARITY = 3 };

struct GemmArgs {
  const void* A;
  const float* B;
  float* out;
  const float* aux;       // bias (EPI_BIAS_RELU) | activation to derive the ReLU mask from (EPI_MASK*)
  int M, N, K;
  int a_rows;             // patch rows available (batch * OH * OW)
  int ldb;
  int ldo;
  int chunks_per_split;
  int slab_rows;          // EPI_SLAB: rows per slab (M, or M+1 when a bias row is appended)
  int tapoff[4][9];       // B_NK_TAPS: element offset of each (parity, tap)
};

constexpr float kInputScale = 0.003921568859368563f;  // networks.py:115, float32(1/255)

template <class G, bool U8>
__device__ __forceinline__ float4 load_patch4(const void* base, long off) {
  if constexpr (U8) {
    const uchar4 v = *reinterpret_cast<const uchar4*>(static_cast<const uint8_t*>(base) + off);
    return make_float4((float)v.x * kInputScale, (float)v.y * kInputScale, (float)v.z * kInputScale,
                       (float)v.w * kInputScale);
  } else {
    return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + off);
  }
}

template <class G, bool U8, int AMODE, int BMODE, int BCO, int EPI, bool BIASROW, int BM, int BN, int WM, int WN,
          int WK>
__global__ __launch_bounds__(256) void igemm_kernel(const GemmArgs p) {
  constexpr int BK = 32;
  constexpr int THREADS = 256;
  static_assert(WM * WN * WK == 4, "4 waves");
  constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN);
  static_assert(TM >= 1 && TN >= 1 && TM * 16 * WM == BM && TN * 16 * WN == BN, "tile shape");
  constexpr int KW_PER_WAVE = BK / WK;
  static_assert(KW_PER_WAVE % 4 == 0, "k per wave");
  // LDS strides
  constexpr int SROW = BK + 18;                                  // row-major tile [rows][BK]: 50
  constexpr int SA_KM = BM + ((BM % 32 == 16) ? 0 : 16);         // k-major A tile [BK][BM]
  constexpr int SB_KM = BN + ((BN % 32 == 16) ? 0 : 16);         // k-major B tile [BK][BN]
  constexpr int A_FLOATS = (AMODE == A_ROWS_M) ? BM * SROW : BK * SA_KM;
  constexpr int B_FLOATS = (BMODE == B_KN) ? BK * SB_KM : BN * SROW;
  constexpr int SC = BN + 4;
  constexpr int C_FLOATS = WK * BM * SC;
  constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
  constexpr int LDS_FLOATS = (STAGE_FLOATS > C_FLOATS) ? STAGE_FLOATS : C_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float* As = lds;
  float* Bs = lds + A_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wk = wave % WK;
  const int wn = (wave / WK) % WN;
  const int wm = wave / (WK * WN);
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int z = blockIdx.z;

  // ---- per-thread staging units -------------------------------------------------------------
  constexpr int A_UNITS = BM * BK / 4;
  constexpr int B_UNITS = BN * BK / 4;
  constexpr int NA = (A_UNITS + THREADS - 1) / THREADS;
  constexpr int NB = (B_UNITS + THREADS - 1) / THREADS;
  float4 ra[NA], rb[NB];

  // A_ROWS_M: rows fixed for the whole K loop -> decode once.
  long a_base[NA];
  int a_iy0[NA], a_ix0[NA];
  bool a_ok[NA];
  if constexpr (AMODE == A_ROWS_M) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int u = tid + i * THREADS;
      const int row = m0 + u / (BK / 4);
      a_ok[i] = (u < A_UNITS) && (row < p.M);
      const int r = a_ok[i] ? row : 0;
      const int b = r / G::OPIX;
      const int rem = r - b * G::OPIX;
      const int oy = rem / G::OW;
      const int ox = rem - oy * G::OW;
      a_iy0[i] = oy * G::S - G::PH;
      a_ix0[i] = ox * G::S - G::PW;
      a_base[i] = ((long)(b * G::IH + a_iy0[i]) * G::IW + a_ix0[i]) * G::C;
    }
  }

  auto load_a = [&](int chunk) {
    const int k0 = chunk * BK;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int u = tid + i * THREADS;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (AMODE == A_ROWS_M) {
        const int f = k0 + (u % (BK / 4)) * 4;
        if (a_ok[i] && f < p.K) {
          const int kh = f / G::KWC;
          const int kwc = f - kh * G::KWC;
          bool ok = true;
          if constexpr (G::PADDED) {
            const int iy = a_iy0[i] + kh;
            const int ix = a_ix0[i] + kwc / G::C;
            ok = (iy >= 0) && (iy < G::IH) && (ix >= 0) && (ix < G::IW);
          }
          if (ok) v = load_patch4<G, U8>(p.A, a_base[i] + (long)kh * (G::IW * G::C) + kwc);
        }
      } else {
        const int rl = u / (BM / 4);
        const int f = m0 + (u % (BM / 4)) * 4;
        const int r = k0 + rl;
        if (u < A_UNITS && r < p.K && f < p.M) {
          const int b = r / G::OPIX;
          const int rem = r - b * G::OPIX;
          const int oy = rem / G::OW;
          const int ox = rem - oy * G::OW;
          const int kh = f / G::KWC;
          const int kwc = f - kh * G::KWC;
          const int iy = oy * G::S - G::PH + kh;
          const int ix = ox * G::S - G::PW + kwc / G::C;
          bool ok = true;
          if constexpr (G::PADDED) ok = (iy >= 0) && (iy < G::IH) && (ix >= 0) && (ix < G::IW);
          if (ok) v = load_patch4<G, U8>(p.A, ((long)(b * G::IH + iy) * G::IW + (ox * G::S - G::PW)) * G::C + kwc);
        }
      }
      ra[i] = v;
    }
  };

  auto load_b = [&](int chunk) {
    const int k0 = chunk * BK;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int u = tid + i * THREADS;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (BMODE == B_KN) {
        const int k = k0 + u / (BN / 4);
        const int n = n0 + (u % (BN / 4)) * 4;
        if (u < B_UNITS && k < p.K && n < p.N) v = *reinterpret_cast<const float4*>(p.B + (long)k * p.ldb + n);
      } else {
        const int n = n0 + u / (BK / 4);
        const int k = k0 + (u % (BK / 4)) * 4;
        if (u < B_UNITS && k < p.K && n < p.N) {
          const int tap = k / BCO;
          const int co = k - tap * BCO;
          const int par = (EPI == EPI_MASK_PARITY) ? z : 0;
          v = *reinterpret_cast<const float4*>(p.B + (long)p.tapoff[par][tap] + (long)n * BCO + co);
        }
      }
      rb[i] = v;
    }
  };

  auto store_lds = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int u = tid + i * THREADS;
      if (u < A_UNITS) {
        if constexpr (AMODE == A_ROWS_M) {
          float* d = As + (u / (BK / 4)) * SROW + (u % (BK / 4)) * 4;
          d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
        } else {
          *reinterpret_cast<float4*>(As + (u / (BM / 4)) * SA_KM + (u % (BM / 4)) * 4) = ra[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int u = tid + i * THREADS;
      if (u < B_UNITS) {
        if constexpr (BMODE == B_KN) {
          *reinterpret_cast<float4*>(Bs + (u / (BN / 4)) * SB_KM + (u % (BN / 4)) * 4) = rb[i];
        } else {
          float* d = Bs + (u / (BK / 4)) * SROW + (u % (BK / 4)) * 4;
          d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w;
        }
      }
    }
  };

  // ---- main loop ----------------------------------------------------------------------------
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = (p.K + BK - 1) / BK;
  int c_begin = 0, c_end = nchunks;
  if constexpr (EPI == EPI_SLAB) {
    c_begin = z * p.chunks_per_split;
    c_end = min(c_begin + p.chunks_per_split, nchunks);
  }
  float bias_acc = 0.f;  // BIASROW: column sum of the B tile (dY), thread n < BN

  const int l15 = lane & 15;
  const int l4 = lane >> 4;
  if (c_begin < c_end) {
    load_a(c_begin);
    load_b(c_begin);
  }
  for (int c = c_begin; c < c_end; ++c) {
    store_lds();
    __syncthreads();
    if (c + 1 < c_end) {
      load_a(c + 1);
      load_b(c + 1);
    }
#pragma unroll
    for (int ks = 0; ks < KW_PER_WAVE; ks += 4) {
      const int kk = wk * KW_PER_WAVE + ks + l4;
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (TM * 16) + i * 16 + l15;
        a[i] = (AMODE == A_ROWS_M) ? As[row * SROW + kk] : As[kk * SA_KM + row];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = wn * (TN * 16) + j * 16 + l15;
        b[j] = (BMODE == B_KN) ? Bs[kk * SB_KM + col] : Bs[col * SROW + kk];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if constexpr (BIASROW) {
      if (blockIdx.x == 0 && tid < BN) {
#pragma unroll 8
        for (int k = 0; k < BK; ++k) bias_acc += Bs[k * SB_KM + tid];
      }
    }
    __syncthreads();
  }

  // ---- epilogue through LDS (sums the WK partials, coalesced float4 stores) -------------------
  float* Cs = lds;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * (TM * 16) + i * 16 + l4 * 4 + r;
        const int col = wn * (TN * 16) + j * 16 + l15;
        Cs[(wk * BM + row) * SC + col] = acc[i][j][r];
      }
  __syncthreads();
  constexpr int C_UNITS = BM * BN / 4;
  for (int u = tid; u < C_UNITS; u += THREADS) {
    const int row = u / (BN / 4);
    const int c4 = (u % (BN / 4)) * 4;
    float4 v = *reinterpret_cast<const float4*>(Cs + row * SC + c4);
#pragma unroll
    for (int w = 1; w < WK; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(Cs + (w * BM + row) * SC + c4);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const int m = m0 + row;
    const int n = n0 + c4;
    if (m >= p.M || n >= p.N) continue;
    if constexpr (EPI == EPI_BIAS_RELU) {
      const float4 bb = *reinterpret_cast<const float4*>(p.aux + n);
      v.x = fmaxf(v.x + bb.x, 0.f); v.y = fmaxf(v.y + bb.y, 0.f);
      v.z = fmaxf(v.z + bb.z, 0.f); v.w = fmaxf(v.w + bb.w, 0.f);
      *reinterpret_cast<float4*>(p.out + (long)m * p.ldo + n) = v;
    } else if constexpr (EPI == EPI_SLAB) {
      *reinterpret_cast<float4*>(p.out + ((long)z * p.slab_rows + m) * p.ldo + n) = v;
    } else {
      long orow = m;
      if constexpr (EPI == EPI_MASK_PARITY) {
        const int b = m / G::OPIX;
        const int rem = m - b * G::OPIX;
        const int a = rem / G::OW;
        const int cc = rem - a * G::OW;
        orow = ((long)b * (2 * G::OH) + 2 * a + (z >> 1)) * (2 * G::OW) + 2 * cc + (z & 1);
      }
      const float4 act = *reinterpret_cast<const float4*>(p.aux + orow * p.ldo + n);
      v.x = act.x > 0.f ? v.x : 0.f; v.y = act.y > 0.f ? v.y : 0.f;
      v.z = act.z > 0.f ? v.z : 0.f; v.w = act.w > 0.f ? v.w : 0.f;
      *reinterpret_cast<float4*>(p.out + orow * p.ldo + n) = v;
    }
  }
  if constexpr (BIASROW) {
    if (blockIdx.x == 0 && tid < BN && n0 + tid < p.N)
      p.out[((long)z * p.slab_rows + p.M) * p.ldo + n0 + tid] = bias_acc;
  }
}

template <class G, bool U8, int AMODE, int BMODE, int BCO, int EPI, bool BIASROW, int BM, int BN, int WM, int WN,
          int WK>
inline void launch_igemm(const GemmArgs& a, int zdim, hipStream_t s) {
  dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, zdim);
  hipLaunchKernelGGL((igemm_kernel<G, U8, AMODE, BMODE, BCO, EPI, BIASROW, BM, BN, WM, WN, WK>), grid, dim3(256), 0, s, a);
}

}  // namespace paac
