// LDS-tiled split-bf16 GEMM for the fc layer at LARGE training batches (more than 512 rows: the 256-environment
// configuration, the 8 x 128-environment shards at t_max 20): forward (networks.py:49-60), data gradient and weight gradient
// of the same layer.  C[M, N] = A[M, K] * B[K, N], fp32 operands and results, v_mfma_f32_16x16x32_bf16.
//
// Arithmetic = dmm.h's XB = 2 path (each fp32 value split EXACTLY into hi + mid + lo bf16 terms; the six partial products
// down to 2^-16 of the leading one, smallest first, fp32 accumulation).  dmm.h keeps operand fragments in registers: its
// reuse is the wave's 64 x 64 register tile, and every wave re-splits what it loads -- at thousands of rows it is bound by
// L2 traffic and VALU work (fc_fwd at 2,688 rows: 77 TFLOP/s).  Here a workgroup of 8 waves owns a 128 x 128 tile of C:
// per 32-deep K step each of its 512 threads loads 8 consecutive-k values of one A row and of one B column, splits them
// ONCE, and parks the three planes in LDS in MFMA fragment order ([16-row tile][plane][64 lanes][8 bf16]: a fragment is one
// conflict-free ds_read_b128); the waves (2 x 4, 64 x 32 of C each) then issue 48 MFMAs per step from 18 fragment reads.
// Global loads run two K steps ahead in registers, LDS is double buffered: one barrier per step.
//
//   A_KC: A is [M, lda] row-major (k contiguous);  else A is given transposed, [K, lda] with m contiguous (weight gradient:
//         A = activations^T).   B_KC: B is given as [N, ldb] row-major (k contiguous; data gradient: B = the weights as
//         stored);  else B is [K, ldb] row-major (n contiguous).
//   EPI_SLAB: C goes to split-K slab z (out + z * slab_stride), summed by the consumer;  EPI_MASK: out = mask > 0 ? C : 0.
// K range of a workgroup: stages [z * stages_per_split, ...) of 32; rows / columns beyond M / N read as zero.
#pragma once
#include "dmm.h"

namespace paac {

struct Gemm3Args {
  const float* A;
  const float* B;
  float* out;
  const float* mask;
  int lda, ldb, ldo;
  int M, N, K;
  int MB, NB, S;          // workgroups = MB * NB * S
  int stages_per_split;
  long slab_stride;
  float* colsum_out;      // nullable: [S][N] column sums of B over each split's K range (the bias gradient of a weight gradient)
};

// WGM x WGN waves (8 in all), each owning TMW x TNW tiles of 16 x 16: the workgroup tile is TM x TN = (16 WGM TMW) x (16 WGN TNW).
//   <2, 4, 4, 2>: 128 x 128, the default (48 MFMAs per 18 fragment reads and wave);
//   <4, 2, 2, 2>: 128 x 64 for outputs with few column blocks (the weight gradient: 25 x 8 tiles of [3136, 512], no K split).
template <bool A_KC, bool B_KC, int EPI, int WGM = 2, int WGN = 4, int TMW = 4, int TNW = 2>
struct Gemm3 {
  static constexpr int THREADS = 512, TM = 16 * WGM * TMW, TN = 16 * WGN * TNW;
  static_assert(WGM * WGN == 8 && TM <= 128 && TN <= 128, "8 waves; one loader item per thread and operand");
  static constexpr int A_VECS = (TM / 16) * 3 * 64, B_VECS = (TN / 16) * 3 * 64;   // bf16x8 vectors of one operand stage
  static constexpr int STAGE_VECS = A_VECS + B_VECS;
  static constexpr int SMEM_BYTES = 2 * STAGE_VECS * 16;             // 96 KB at 128 x 128
  static constexpr int NMFMA = 6 * TMW * TNW;

  // 8 consecutive k of row / column `r` of an operand tile, from a k-contiguous or an r-contiguous source.  No branch
  // around the loads (the compiler then keeps exact s_waitcnt counts and with them the prefetch distance): a row / column
  // past the end re-reads the last one -- it only feeds rows / columns of C that are never stored.
  template <bool KC>
  __device__ __forceinline__ static void fetch(const float* base, const int ld, const int r_glob, const int r_lim, const int k,
                                               float (&x)[8]) {
    const int r = r_glob < r_lim ? r_glob : r_lim - 1;
    if constexpr (KC) {
      const float4* src = reinterpret_cast<const float4*>(base + (long)r * ld + k);
      const float4 v0 = src[0], v1 = src[1];
      x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = base[(long)(k + e) * ld + r];     // lanes = consecutive r: coalesced rows
    }
  }

  __device__ __forceinline__ static void run(const Gemm3Args& p, const int bid, char* smem) {
    bf16x8* lds = reinterpret_cast<bf16x8*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;              // WGM x WGN waves: rows 16 TMW wm .., columns 16 TNW wn ..
    // block -> (z, n block, m block): the K splits of one C tile are neighbours (one per XCD when S = 8 ... any S works)
    const int z = bid % p.S;
    const int nb = (bid / p.S) % p.NB;
    const int mb = bid / (p.S * p.NB);
    const int m0 = mb * TM, n0 = nb * TN;
    const int s_begin = z * p.stages_per_split;
    const int nst = min(p.stages_per_split, p.K / 32 - s_begin);
    // loader item of this thread, for both operands: row/column r = 16 (tid >> 6) + (tid & 15) of the tile, k slot
    // (tid >> 4) & 3  ==  MFMA lane (tid & 63) of 16-row tile tid >> 6
    const int r_item = (tid >> 6) * 16 + (tid & 15), kq = (tid >> 4) & 3;
    bf16x8* const park_at = lds + (tid >> 6) * 3 * 64 + lane;
    const bool a_item = r_item < TM, b_item = r_item < TN;   // wave-uniform (a narrower operand has fewer 16-row tiles)

    float ra[2][8], rb[2][8];
    auto issue = [&](const int set, const int st_) {      // a stage past the end re-reads the last one (never consumed)
      const int st = st_ < nst ? st_ : nst - 1;
      const int k = (s_begin + st) * 32 + 8 * kq;
      fetch<A_KC>(p.A, p.lda, m0 + (a_item ? r_item : 0), p.M, k, ra[set]);     // (a surplus wave re-reads item 0: never parked)
      fetch<B_KC>(p.B, p.ldb, n0 + (b_item ? r_item : 0), p.N, k, rb[set]);
    };
    auto park = [&](const int set, const int buf) {
      bf16x8 h, m, l;
      split3_bf16(ra[set], h, m, l);
      bf16x8* d = park_at + buf * STAGE_VECS;
      if (TM == 128 || a_item) { d[0] = h; d[64] = m; d[128] = l; }
      split3_bf16(rb[set], h, m, l);
      d += A_VECS;
      if (TN == 128 || b_item) { d[0] = h; d[64] = m; d[128] = l; }
    };

    f32x4 acc[TMW][TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < TNW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias gradient of a weight-gradient contraction: column sums of B over K, taken by the row-block-0 workgroups from the
    // values they load anyway (this thread: 8 k of column r_item per stage)
    float bsum = 0.f;
    const bool want_bsum = p.colsum_out != nullptr && mb == 0;

    if (nst <= 0) return;          // (uniform; cannot happen with the launcher's split arithmetic)
    issue(0, 0);
    issue(1, 1);
    park(0, 0);
    issue(0, 2);
    __syncthreads();
    for (int base = 0; base < nst; base += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int st = base + u;
        if (st >= nst) break;
        const bf16x8* a_src = lds + u * STAGE_VECS + (wm * TMW) * 3 * 64 + lane;
        const bf16x8* b_src = lds + u * STAGE_VECS + A_VECS + (wn * TNW) * 3 * 64 + lane;
        // Fragment reads pipelined against the MFMAs inside the stage: the column fragments and row tile 0 first, then row
        // tile i + 1 is requested before row tile i's 6 TNW MFMAs issue, so its LDS latency (and the other waves' LDS
        // traffic) hides behind the matrix pipe; the split of stage st + 1 (VALU only) is spread over the same MFMAs.
        bf16x8 bh[TNW], bm[TNW], bl[TNW], ah[2], am[2], al[2];
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
          bh[j] = b_src[j * 192];
          bm[j] = b_src[j * 192 + 64];
          bl[j] = b_src[j * 192 + 128];
        }
        ah[0] = a_src[0];
        am[0] = a_src[64];
        al[0] = a_src[128];
        bf16x8 nah, nam, nal, nbh, nbm, nbl;
        split3_bf16(ra[u ^ 1], nah, nam, nal);
        split3_bf16(rb[u ^ 1], nbh, nbm, nbl);
        if (want_bsum && st + 1 < nst) {
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum += rb[u ^ 1][e];
        }
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
          const int cur = i & 1, nxt = cur ^ 1;
          if (i + 1 < TMW) {
            ah[nxt] = a_src[(i + 1) * 192];
            am[nxt] = a_src[(i + 1) * 192 + 64];
            al[nxt] = a_src[(i + 1) * 192 + 128];
          }
          // smallest terms first; the TNW accumulators of the row alternate between two products of the same one
#pragma unroll
          for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[cur], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[cur], bm[j], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cur], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[cur], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cur], bm[j], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cur], bh[j], acc[i][j], 0, 0, 0);
        }
        // the parked planes of stage st + 1 go to the OTHER LDS buffer: nothing reads it during this stage, so the stores are
        // issued in the middle of the MFMA stream (a ds_write_b128 takes ~13 cycles of the store path per wave: eight waves
        // storing together at the end of the stage, in front of the barrier, left the matrix pipe idle for the duration)
        {
          bf16x8* d = park_at + (u ^ 1) * STAGE_VECS;
          if (TM == 128 || a_item) { d[0] = nah; d[64] = nam; d[128] = nal; }
          d += A_VECS;
          if (TN == 128 || b_item) { d[0] = nbh; d[64] = nbm; d[128] = nbl; }
        }
        // scheduling: per row tile the three fragment reads of the next one, then its MFMAs with the split's VALU between;
        // the A planes are stored behind the second-to-last row tile's MFMAs ... the B planes behind the one before the last
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
          if (i + 1 < TMW) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
          for (int k = 0; k < 6 * TNW; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 96 / NMFMA, 0);
          }
          if (i == (TMW >= 4 ? 1 : 0)) __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
          if (i == (TMW >= 4 ? 2 : TMW - 1)) __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
        }
        issue(u ^ 1, st + 3);
        __syncthreads();
      }
    }

    // ---- epilogue: D layout -- lane (j = lane & 15, q = lane >> 4) holds rows 4 q .. 4 q + 3 of column j -----------------
    const int jc = lane & 15, q = lane >> 4;
    if (want_bsum && b_item) {
      // the loop added stages 1 .. nst - 1 as it split them; stage 0 went to LDS in the prologue and is re-read here (once
      // per workgroup)
      float x0[8];
      fetch<B_KC>(p.B, p.ldb, n0 + r_item, p.N, s_begin * 32 + 8 * kq, x0);
#pragma unroll
      for (int e = 0; e < 8; ++e) bsum += x0[e];
      bsum += __shfl_xor(bsum, 16, 64);          // the four k slots of a column sit 16 lanes apart
      bsum += __shfl_xor(bsum, 32, 64);
      if (kq == 0 && n0 + r_item < p.N) p.colsum_out[(long)z * p.N + n0 + r_item] = bsum;
    }
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
#pragma unroll
      for (int j = 0; j < TNW; ++j) {
        const int n = n0 + (wn * TNW + j) * 16 + jc;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + (wm * TMW + i) * 16 + 4 * q + r;
          if (m >= p.M) continue;
          if constexpr (EPI == EPI_SLAB) {
            p.out[(long)z * p.slab_stride + (long)m * p.ldo + n] = acc[i][j][r];
          } else {
            const long off = (long)m * p.ldo + n;
            p.out[off] = p.mask[off] > 0.f ? acc[i][j][r] : 0.f;
          }
        }
      }
    }
  }
};

template <class D>
__global__ __launch_bounds__(D::THREADS) void gemm3_kernel(const Gemm3Args p) {
  __shared__ __attribute__((aligned(16))) char smem[D::SMEM_BYTES];
  D::run(p, blockIdx.x, smem);
}

}  // namespace paac
