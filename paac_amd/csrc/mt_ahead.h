// MT19937 work done one launch AHEAD of the numpy-parity sampler (paac.py:34-45, np.random.multinomial on the global
// legacy stream).  The sampler's 53-bit doubles depend on nothing but the stream position: a spare workgroup of the acting
// forward's fc launch (csrc/fc_heads.h) twists the state blocks the next sampling step can reach and leaves the doubles, in
// stream order, with the blocks themselves and a key in global memory; every sampler workgroup of the following launch
// (csrc/misc.hip: sample_mt_body) then loads its doubles together with the probabilities -- one round trip -- instead of
// rebuilding blocks and doubles itself (4.2 of its 21 us at 128 environments x 18 actions).  The key (stream position +
// three state words) is checked by the consumer: a record that does not belong to the current stream state is ignored and
// the sampler builds everything itself, as before.
#pragma once
#include "common.h"

namespace paac {

constexpr int MT_LDS_D = 1024;            // sampler LDS class 1: draws (N * (A - 1)) the LDS arrays hold
constexpr int MT_LDS_D2 = 2304;           // class 2
constexpr int MT_AHEAD_BLK = (624 + 2 * MT_LDS_D2) / 624 + 2;   // state blocks a class-2 step can reach (10)

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b) {
  const uint32_t yy = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (yy >> 1) ^ ((yy & 1u) ? 0x9908b0dfu : 0u);
}

struct MtAhead {
  uint32_t hdr[16];                        // [0] stream position the record was made at, [1] [2] [3] state words 0, 1, 623,
                                           // [4] doubles produced (0: none), [5] blocks produced
  double u[MT_LDS_D2];                     // the doubles numpy would draw from that position on
  uint32_t blocks[MT_AHEAD_BLK * 624];     // the state blocks they come from (block 0 = the state at production)
};
struct MtAheadArgs {                       // producer side: null `out` = nothing to do
  const uint32_t* state;                   // mt_state [625]
  MtAhead* out;
  int D;
};

// One workgroup of NT threads (every thread of it must call).  lds: MT_AHEAD_BLK * 624 words.
template <int NT>
__device__ __forceinline__ void mt_produce_ahead(const MtAheadArgs a, uint32_t* lds) {
  const int tid = threadIdx.x;
  const uint32_t pos = a.state[624];
  for (int i = tid; i < 624; i += NT) lds[i] = a.state[i];
  __syncthreads();
  const int D = a.D < MT_LDS_D2 ? a.D : MT_LDS_D2;
  if (pos > 624u) {                        // not a numpy-convention state: leave no record
    if (tid == 0) a.out->hdr[4] = 0u;
    return;
  }
  const int nblk = (int)((pos + 2u * (uint32_t)D) / 624u) + 1;
  for (int b = 1; b < nblk; ++b) {
    const uint32_t* o = lds + (b - 1) * 624;
    uint32_t* nw = lds + b * 624;
    for (int k = tid; k < 227; k += NT) nw[k] = o[k + 397] ^ mt_mix(o[k], o[k + 1]);
    __syncthreads();
    for (int k = 227 + tid; k < 454; k += NT) nw[k] = nw[k - 227] ^ mt_mix(o[k], o[k + 1]);
    __syncthreads();
    for (int k = 454 + tid; k < 623; k += NT) nw[k] = nw[k - 227] ^ mt_mix(o[k], o[k + 1]);
    if (tid == NT - 1) nw[623] = nw[396] ^ mt_mix(o[623], nw[0]);      // both inputs are older than this pass
    __syncthreads();
  }
  for (int d = tid; d < D; d += NT) {
    const uint32_t q = pos + 2u * (uint32_t)d;
    const uint32_t hi = mt_temper(lds[q]) >> 5;
    const uint32_t lo = mt_temper(lds[q + 1]) >> 6;
    a.out->u[d] = ((double)hi * 67108864.0 + (double)lo) * 1.1102230246251565404e-16;   // * 2^-53, exact
  }
  for (int i = tid; i < nblk * 624; i += NT) a.out->blocks[i] = lds[i];
  if (tid == 0) {
    a.out->hdr[0] = pos;
    a.out->hdr[1] = lds[0];
    a.out->hdr[2] = lds[1];
    a.out->hdr[3] = lds[623];
    a.out->hdr[4] = (uint32_t)D;
    a.out->hdr[5] = (uint32_t)nblk;
  }
}

}  // namespace paac
