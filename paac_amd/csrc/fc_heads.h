// Acting-path tail of the network for acting batches (<= 256 rows): the fc layer (networks.py:49-60) with the head
// contractions (policy_v_network.py:24-26) folded into its epilogue, and the head finish (bias, softmax).
//
//   fc_heads_q_kernel    the same on quarter tiles, (8 rows, 8 fc columns) per workgroup, while those are at most one
//                        workgroup per CU (32 rows of the stock fc widths): half the bytes through each CU's L1, see below
//   fc_heads_kernel      one workgroup per (16 rows, 16 fc columns): full K inside the workgroup (7 or 9 waves split K evenly, summed
//                        through LDS), so bias + ReLU apply to complete sums and NO split-K slab leaves the workgroup; its
//                        epilogue multiplies the 16x16 tile of h with the matching 16 rows of the actor / critic weights
//                        and writes the tile's share of every logit and of the value: partial[tile][row][A + 1] -- a few
//                        hundred bytes per workgroup instead of the 64 KB of h the heads used to re-read per step.
//   heads_from_partials  device routine, one workgroup: sums the H / 16 (or H / 8) partials per (row, output) in fixed order,
//                        adds the head biases, softmax.  Called by the stand-alone heads_finish_kernel (paac_forward) and
//                        by workgroup 0 of the fused sampler + environment-step launch (csrc/misc.hip), which leaves the
//                        probabilities in LDS for the sampler: the acting step is three launches (conv tower, fc + head
//                        partials, heads finish + sampler + env step).  Both callers run the same code: same bits.
// fp32 MFMA (v_mfma_f32_16x16x4_f32): at 32 rows the layer is 0.1 GFLOP against 6.4 MB of weights; each workgroup
// streams its 16 columns (200 KB) once, direct to registers in the MFMA layout (K order permuted inside each 16-wide
// group, identically for both operands, like dmm.h).
#pragma once
#include "dmm.h"
#include "mt_ahead.h"

namespace paac {

// waves per workgroup of fc_heads_kernel for an fc layer of `flat` inputs: they split the flat / 16 K groups evenly and each
// needs more than the prefetch depth (8) of them; 0 = no such split (the layer then runs on the split-K GEMM at every batch)
constexpr int fc_heads_waves(const int flat) {
  const int g = flat / 16;
  for (const int w : {7, 9, 8, 6, 5, 10, 4, 12, 3})
    if (flat % 16 == 0 && g % w == 0 && g / w > 8) return w;
  return 0;
}

constexpr int kFcHeadsMaxRows = 64;      // one finishing workgroup (it can be workgroup 0 of the sampler + env-step launch)
constexpr int kFcHeadsMidRows = 256;     // acting batches up to here take the same fc kernel, finished by heads_finish_rows_kernel

// Packed fc weights for fc_heads_kernel: wfp[tile nt][group g][lane][4] = Wf[16 g + 4 kq + s][16 nt + li], s = 0..3 (lane =
// 16 kq + li): the B fragment of one 16-wide K group is ONE 16-byte load per lane, 1 KB contiguous per wave (the plain
// [K, H] layout gives four dword loads that use half of every 128-byte line they touch).
template <int K, int H>
__global__ __launch_bounds__(256) void pack_fc_kernel(const float* __restrict__ Wf, f32x4* __restrict__ out) {
  constexpr int G = K / 16;
  const int i = blockIdx.x * 256 + threadIdx.x;       // one thread per (tile, group, lane)
  if (i >= (H / 16) * G * 64) return;
  const int lane = i & 63, unit = i >> 6;
  const int g = unit % G, nt = unit / G;
  const int li = lane & 15, kq = lane >> 4;
  f32x4 v;
#pragma unroll
  for (int s = 0; s < 4; ++s) v[s] = Wf[(size_t)(16 * g + 4 * kq + s) * H + 16 * nt + li];
  out[i] = v;
}

// NW waves split K evenly (NW divides K / 16: 7 x 28 groups for Nature's 3136, 9 x 18 for NIPS' 2592), so the K loop of a
// wave has a compile-time trip count and is fully unrolled: straight-line loads PF groups ahead of their MFMAs, exact
// s_waitcnt vmcnt(N) counts (a loop with clamped tails made the compiler sink the prefetch loads behind vmcnt(0)).
// A_PACKED: `act` is in fragment order [row tile][K group][lane][4] (written that way by the conv tower's epilogue): the A
// fragment of a group is one coalesced 16-byte load per lane too; otherwise plain rows [B, K].
template <int K, int H, int NW, bool A_PACKED>
__global__ __launch_bounds__(64 * NW) void fc_heads_kernel(const float* __restrict__ act, const f32x4* __restrict__ wfp,
                                                           const float* __restrict__ bf, const float* __restrict__ Wa,
                                                           const float* __restrict__ Wc, const int A, const int B,
                                                           float* __restrict__ partial, float* __restrict__ h_out,
                                                           const MtAheadArgs ahead) {
  constexpr int NTILES = H / 16, G = K / 16, GPW = G / NW, PF = 8;
  static_assert(K % 16 == 0 && H % 16 == 0 && G % NW == 0 && GPW > PF, "fc geometry");
  if (ahead.out != nullptr && blockIdx.x == gridDim.x - 1) {
    // the spare workgroup (csrc/mt_ahead.h): the doubles of the sampling step that follows this forward
    __shared__ uint32_t ahead_blocks[MT_AHEAD_BLK * 624];
    mt_produce_ahead<64 * NW>(ahead, ahead_blocks);
    return;
  }
  __shared__ f32x4 red[NW * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int nt = blockIdx.x % NTILES, mt = blockIdx.x / NTILES;
  const int row_a = mt * 16 + li;                       // A operand: this lane's row
  const bool row_ok = row_a < B;
  const int col = nt * 16 + li;                         // D: this lane's column
  // per-group stride of the A fragment pointer: 16 floats along a row, or one 64-lane x 4-float block
  constexpr int A_STEP = A_PACKED ? 256 : 16;
  const float* ap = A_PACKED ? act + (((size_t)mt * G + (size_t)wave * GPW) * 64 + lane) * 4
                             : act + (size_t)(row_ok ? row_a : 0) * K + 4 * kq + (size_t)wave * (GPW * 16);
  const f32x4* bp = wfp + ((size_t)nt * G + (size_t)wave * GPW) * 64 + lane;

  constexpr int RING = PF + 1;
  f32x4 fa[RING], fb[RING];
  // head weights of this workgroup's 16 columns, for the outputs this wave folds in the epilogue (o = wave, wave + NW, ...)
  constexpr int NO = (33 + NW - 1) / NW;
  float wh[NO];
#pragma unroll
  for (int q = 0; q < NO; ++q) {
    const int o = wave + NW * q;
    wh[q] = (o < A) ? Wa[(size_t)col * A + o] : (o == A ? Wc[col] : 0.f);
  }
  const float bias = bf[col];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    fa[i] = *reinterpret_cast<const f32x4*>(ap + A_STEP * i);
    fb[i] = bp[(size_t)i * 64];
  }
  __builtin_amdgcn_sched_barrier(0);       // the scheduler otherwise sinks every load to just before its use
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    if (i + PF < GPW) {
      fa[(i + PF) % RING] = *reinterpret_cast<const f32x4*>(ap + A_STEP * (i + PF));
      fb[(i + PF) % RING] = bp[(size_t)(i + PF) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(row_ok ? fa[i % RING][e] : 0.f, fb[i % RING][e], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  red[wave * 64 + lane] = acc;
  __syncthreads();
  // every wave rebuilds the complete tile (fixed order), applies bias + ReLU, then folds its share of the head outputs
  f32x4 h = red[lane];
#pragma unroll
  for (int w = 1; w < NW; ++w) h += red[w * 64 + lane];
#pragma unroll
  for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r] + bias, 0.f);     // D layout: rows 4 kq + r, column li
  if (h_out != nullptr && wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = mt * 16 + 4 * kq + r;
      if (row < B) h_out[(size_t)row * H + col] = h[r];
    }
  }
#pragma unroll
  for (int q = 0; q < NO; ++q) {
    const int o = wave + NW * q;
    if (o > A) break;                                  // wave-uniform
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = h[r] * wh[q];
      t += __shfl_xor(t, 1, 64);                       // sum over the 16 columns of the tile (lanes li of one kq)
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      t += __shfl_xor(t, 8, 64);
      v[r] = t;
    }
    if (li == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mt * 16 + 4 * kq + r;
        if (row < B) partial[((size_t)nt * B + row) * (A + 1) + o] = v[r];
      }
    }
  }
}

// Quarter tiles for the smallest batches: a workgroup owns 8 rows x 8 fc columns, 256 workgroups at 32 rows of the stock
// width -- one per CU, 200 KB each through its L1 instead of 64 workgroups of 400 KB.  Every lane still loads 16 useful
// bytes per operand and K group: lanes li < 8 feed the MFMA rows / columns 0..7 with K group 2 i, lanes li >= 8 feed rows /
// columns 8..15 with THE SAME 8 rows / columns at K group 2 i + 1 -- the diagonal blocks D[0..7][0..7] and D[8..15][8..15]
// are then the tile's partial sums over the even and the odd K groups (the off-diagonal blocks mix the two and are dropped):
// half of the matrix pipe's work is thrown away, in a layer that keeps it 8 % busy.  (Masking the upper lanes instead --
// zeros, no loads -- was measured SLOWER than whole tiles, 9.45 against 8.1 us: a load instruction occupies the L1 for
// the same time whatever its lane mask.)
template <int K, int H, int NW, bool A_PACKED>
__global__ __launch_bounds__(64 * NW) void fc_heads_q_kernel(const float* __restrict__ act, const f32x4* __restrict__ wfp,
                                                             const float* __restrict__ bf, const float* __restrict__ Wa,
                                                             const float* __restrict__ Wc, const int A, const int B,
                                                             float* __restrict__ partial, float* __restrict__ h_out) {
  constexpr int NTILES = H / 8, G = K / 16, GPW = G / NW, NS = GPW / 2, PF = 6;
  static_assert(K % 16 == 0 && H % 16 == 0 && G % NW == 0 && GPW % 2 == 0 && NS > PF, "fc geometry");
  __shared__ f32x4 red[NW * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int l8 = li & 7, odd = li >> 3;                 // position inside the 8-wide tile; which K group of a pair this lane feeds
  const int nt = blockIdx.x % NTILES, mt = blockIdx.x / NTILES;
  const int row_a = mt * 8 + l8;                        // A operand: this lane's row
  const bool row_ok = row_a < B;
  const int col = nt * 8 + l8;                          // B operand / D: this lane's column
  // per-pair stride of the A fragment pointer: 32 floats along a row, or two 64-lane x 4-float blocks
  constexpr int A_STEP = A_PACKED ? 512 : 32;
  const int g0 = wave * GPW + odd;                      // this lane's first K group
  const float* ap = A_PACKED ? act + ((((size_t)(mt >> 1) * G + g0) * 64) + kq * 16 + (mt & 1) * 8 + l8) * 4
                             : act + (size_t)(row_ok ? row_a : 0) * K + 4 * kq + (size_t)g0 * 16;
  const f32x4* bp = wfp + ((size_t)(nt >> 1) * G + g0) * 64 + kq * 16 + (nt & 1) * 8 + l8;

  constexpr int RING = PF + 1;
  f32x4 fa[RING], fb[RING];
  constexpr int NO = (33 + NW - 1) / NW;
  float wh[NO];
#pragma unroll
  for (int q = 0; q < NO; ++q) {
    const int o = wave + NW * q;
    wh[q] = (o < A) ? Wa[(size_t)col * A + o] : (o == A ? Wc[col] : 0.f);
  }
  const float bias = bf[col];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    fa[i] = *reinterpret_cast<const f32x4*>(ap + A_STEP * i);
    fb[i] = bp[(size_t)i * 128];
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    if (i + PF < NS) {
      fa[(i + PF) % RING] = *reinterpret_cast<const f32x4*>(ap + A_STEP * (i + PF));
      fb[(i + PF) % RING] = bp[(size_t)(i + PF) * 128];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(row_ok ? fa[i % RING][e] : 0.f, fb[i % RING][e], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  red[wave * 64 + lane] = acc;
  __syncthreads();
  // D layout: lane (li, kq) holds rows 4 kq + r of column li.  The tile's sums: even groups at (li < 8, kq < 2), odd groups at
  // (li + 8, kq + 2) = 40 lanes further on.  Every wave rebuilds the tile in the same fixed order.
  const bool own = li < 8 && kq < 2;
  const int src = own ? lane : 0;
  f32x4 h = red[src] + red[src + 40];
#pragma unroll
  for (int w = 1; w < NW; ++w) h += red[w * 64 + src] + red[w * 64 + src + 40];
#pragma unroll
  for (int r = 0; r < 4; ++r) h[r] = own ? fmaxf(h[r] + bias, 0.f) : 0.f;     // rows 4 kq + r (kq < 2), column li
  if (h_out != nullptr && wave == 0 && own) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = mt * 8 + 4 * kq + r;
      if (row < B) h_out[(size_t)row * H + col] = h[r];
    }
  }
#pragma unroll
  for (int q = 0; q < NO; ++q) {
    const int o = wave + NW * q;
    if (o > A) break;                                  // wave-uniform
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = h[r] * wh[q];
      t += __shfl_xor(t, 1, 64);                       // sum over the 8 columns of the tile (lanes li < 8 of one kq)
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      v[r] = t;
    }
    if (li == 0 && kq < 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mt * 8 + 4 * kq + r;
        if (row < B) partial[((size_t)nt * B + row) * (A + 1) + o] = v[r];
      }
    }
  }
}
// does the quarter-tile kernel exist for this geometry (the waves' K groups must pair up)?
constexpr bool fc_heads_quarter_ok(const int flat) {
  const int w = fc_heads_waves(flat);
  return w > 0 && (flat / 16 / w) % 2 == 0 && (flat / 16 / w) / 2 > 6;
}

// Second half of the heads finish: logits of all B rows in lg_s -> softmax, stores.  Ends with a barrier.
__device__ __forceinline__ void heads_softmax_store(const int B, const int A, const float* lg_s, float* probs_lds,
                                                    float* __restrict__ logits_out, float* __restrict__ probs_out,
                                                    float* __restrict__ values_out, float* __restrict__ logits_out2,
                                                    float* __restrict__ probs_out2, float* __restrict__ values_out2) {
  const int tid = threadIdx.x;
  for (int idx = tid; idx < B * A; idx += 256) {          // one (row, action) per thread; max and sum recomputed per thread
    const int row = idx / A, a = idx - row * A;
    const float* lg = lg_s + row * (A + 1);
    float m = lg[0];
    for (int j = 1; j < A; ++j) m = fmaxf(m, lg[j]);
    float sum = 0.f;
    for (int j = 0; j < A; ++j) sum += expf(lg[j] - m);
    const float pa = expf(lg[a] - m) / sum;
    if (probs_lds) probs_lds[idx] = pa;
    if (probs_out) probs_out[idx] = pa;
    if (probs_out2) probs_out2[idx] = pa;
    if (logits_out) logits_out[idx] = lg[a];
    if (logits_out2) logits_out2[idx] = lg[a];
  }
  for (int row = tid; row < B; row += 256) {
    const float v = lg_s[row * (A + 1) + A];
    if (values_out) values_out[row] = v;
    if (values_out2) values_out2[row] = v;
  }
  __syncthreads();
}

// One workgroup (256 threads): partial[NTILES][B][A+1] -> logits, probabilities, values of all B rows.
// lg_s: LDS scratch of B * (A + 1) floats; probs_lds (nullable): [B][A] in LDS for a consumer in the same workgroup.
// Ends with a barrier.
__device__ __forceinline__ void heads_from_partials(const float* __restrict__ partial, const int ntiles, const int B,
                                                    const int A, const float* __restrict__ ba,
                                                    const float* __restrict__ bc, float* lg_s, float* probs_lds,
                                                    float* __restrict__ logits_out, float* __restrict__ probs_out,
                                                    float* __restrict__ values_out, float* __restrict__ logits_out2,
                                                    float* __restrict__ probs_out2, float* __restrict__ values_out2) {
  const int tid = threadIdx.x;
  const int n = B * (A + 1);
  for (int idx = tid; idx < n; idx += 256) {
    const int a = idx % (A + 1);
    float acc = (a < A) ? ba[a] : bc[0];
    for (int t0 = 0; t0 < ntiles; t0 += 32) {          // fc widths beyond 512 (user architectures): 32 tiles at a time
      float v[32];
#pragma unroll
      for (int t = 0; t < 32; ++t) v[t] = partial[(size_t)(t0 + t < ntiles ? t0 + t : 0) * n + idx];   // every load before the sum
#pragma unroll
      for (int t = 0; t < 32; ++t) acc += (t0 + t < ntiles) ? v[t] : 0.f;
    }
    lg_s[idx] = acc;
  }
  __syncthreads();
  heads_softmax_store(B, A, lg_s, probs_lds, logits_out, probs_out, values_out, logits_out2, probs_out2, values_out2);
}

static __global__ __launch_bounds__(256) void heads_finish_kernel(const float* __restrict__ partial, int ntiles, int B, int A,
                                                           const float* __restrict__ ba, const float* __restrict__ bc,
                                                           float* __restrict__ logits_ws, float* __restrict__ probs_ws,
                                                           float* __restrict__ values_ws, float* __restrict__ logits_out,
                                                           float* __restrict__ probs_out, float* __restrict__ values_out) {
  __shared__ float lg_s[kFcHeadsMaxRows * 33];
  heads_from_partials(partial, ntiles, B, A, ba, bc, lg_s, nullptr, logits_ws, probs_ws, values_ws, logits_out, probs_out,
                      values_out);
}

// The heads finish for acting batches beyond one workgroup's reach (64 < B <= 256 rows: the 128- and 256-environment
// shards): workgroup w finishes rows [w * rpw, ...) with rpw = 256 / (A + 1) -- one (row, output) per thread, the same sums
// in the same order as heads_from_partials -- instead of a per-row workgroup re-reading 64 KB of split-K slabs per step.
static __global__ __launch_bounds__(256) void heads_finish_rows_kernel(const float* __restrict__ partial, int ntiles, int B,
                                                                int A, int rpw, const float* __restrict__ ba,
                                                                const float* __restrict__ bc, float* __restrict__ logits_ws,
                                                                float* __restrict__ probs_ws, float* __restrict__ values_ws,
                                                                float* __restrict__ logits_out, float* __restrict__ probs_out,
                                                                float* __restrict__ values_out) {
  __shared__ float lg_s[256];
  const int row0 = blockIdx.x * rpw;
  const int rows = min(rpw, B - row0);
  const int n = rows * (A + 1), n_total = B * (A + 1);
  const int tid = threadIdx.x;
  if (tid < n) {
    const int a = tid % (A + 1);
    float acc = (a < A) ? ba[a] : bc[0];
    for (int t0 = 0; t0 < ntiles; t0 += 32) {
      float v[32];
#pragma unroll
      for (int t = 0; t < 32; ++t)
        v[t] = partial[(size_t)(t0 + t < ntiles ? t0 + t : 0) * n_total + (size_t)row0 * (A + 1) + tid];
#pragma unroll
      for (int t = 0; t < 32; ++t) acc += (t0 + t < ntiles) ? v[t] : 0.f;
    }
    lg_s[tid] = acc;
  }
  __syncthreads();
  auto at = [](float* p, long off) { return p ? p + off : p; };
  heads_softmax_store(rows, A, lg_s, nullptr, at(logits_ws, (long)row0 * A), at(probs_ws, (long)row0 * A), at(values_ws, row0),
                      at(logits_out, (long)row0 * A), at(probs_out, (long)row0 * A), at(values_out, row0));
}

}  // namespace paac
