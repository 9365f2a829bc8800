// Device helpers of the synthetic environments (spec: paac_amd/synthetic.py), shared by the env-step kernels
// (csrc/misc.hip) and the heads kernel that absorbs the step in the counter-based-sampler mode (csrc/heads.h).
#pragma once
#include "common.h"

namespace paac {

constexpr int PRE_BANDS = 7;           // 84 rows = 7 bands x 12 rows

__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t synth_key(uint64_t seed, uint32_t env, uint64_t id) {
  uint32_t k = lowbias32((uint32_t)seed ^ lowbias32(env + 0x9E3779B9u));
  k = lowbias32(k ^ (uint32_t)(seed >> 32) ^ lowbias32((uint32_t)id * 0x85EBCA6Bu + (uint32_t)(id >> 32) + 0x7F4A7C15u));
  return k;
}
__device__ __forceinline__ uint32_t synth_word(uint32_t key, uint32_t w) { return lowbias32(key + w * 0x9E3779B9u + 0x165667B1u); }

struct FinishedRing {
  int32_t count;
  int32_t pad;
  float reward[4096];
  int32_t len[4096];
};

// Per-env bookkeeping shared by both paths: emulator_runner.py:30-31 + paac.py:119-138.  ep_reward0 / ep_len0 = the
// running totals before this step (callers that have something to wait for load them early).
// hr5 = lowbias32(key ^ 0xA511E9B3) % 5 and term = lowbias32(key ^ 0x3C6EF372) < thresh do not depend on the action:
// a caller that waits for the action computes them first (synth_reward_slot / synth_terminal).
__device__ __forceinline__ uint32_t synth_reward_slot(uint32_t key) { return lowbias32(key ^ 0xA511E9B3u) % 5u; }
__device__ __forceinline__ bool synth_terminal(uint32_t key, uint32_t thresh) { return lowbias32(key ^ 0x3C6EF372u) < thresh; }
__device__ __forceinline__ bool synth_bookkeep_hashed(uint32_t hr5, bool term, int e, int act, float ep_reward0,
                                                      int32_t ep_len0, float* rewards_out, float* masks_out,
                                                      float* ep_reward, int32_t* ep_len, FinishedRing* fin) {
  const float table[5] = {-2.f, 0.f, 0.f, 1.f, 3.f};
  const float r = table[(hr5 + (uint32_t)act) % 5u];
  rewards_out[e] = fminf(fmaxf(r, -1.f), 1.f);   // actor_learner.py:95-101
  masks_out[e] = term ? 0.f : 1.f;               // paac.py:119
  const float tot = ep_reward0 + r;
  const int len = ep_len0 + 1;
  if (term) {
    if (fin) {
      const int slot = atomicAdd(&fin->count, 1) & 4095;
      fin->reward[slot] = tot;
      fin->len[slot] = len;
    }
    ep_reward[e] = 0.f;
    ep_len[e] = 0;
  } else {
    ep_reward[e] = tot;
    ep_len[e] = len;
  }
  return term;
}
__device__ __forceinline__ bool synth_bookkeep_with(uint32_t key, int e, int act, uint32_t thresh, float ep_reward0,
                                                    int32_t ep_len0, float* rewards_out, float* masks_out,
                                                    float* ep_reward, int32_t* ep_len, FinishedRing* fin) {
  return synth_bookkeep_hashed(synth_reward_slot(key), synth_terminal(key, thresh), e, act, ep_reward0, ep_len0, rewards_out,
                               masks_out, ep_reward, ep_len, fin);
}
__device__ __forceinline__ bool synth_bookkeep(uint32_t key, int e, const int32_t* actions, uint32_t thresh,
                                               float* rewards_out, float* masks_out, float* ep_reward, int32_t* ep_len,
                                               FinishedRing* fin) {
  return synth_bookkeep_with(key, e, actions ? actions[e] : 0, thresh, ep_reward[e], ep_len[e], rewards_out, masks_out,
                             ep_reward, ep_len, fin);
}

// Frame shift of one (env, band) unit of path A: push the step's new 84x84 plane into the 4-deep history
// (one dword = the 4 channels of a pixel).  unit = env * PRE_BANDS + band; 256 threads, 252 of them own one quad of four
// pixels each: ONE generator word is the quad's four new bytes, one 16-byte load and one 16-byte store per thread.
__device__ __forceinline__ void synth_shift_band(uint64_t seed, uint32_t env_offset, uint64_t id, uint32_t thresh, int unit,
                                                 const uint32_t* __restrict__ stack_in, uint32_t* __restrict__ stack_out,
                                                 uint32_t* __restrict__ stack_out2 = nullptr,
                                                 const bool force_reset = false) {
  const int e = unit / PRE_BANDS;
  const int band = unit % PRE_BANDS;
  constexpr int QUADS_PER_BAND = OBS_PIX / PRE_BANDS / 4;  // 252: 12 rows of 21 quads
  const int i = threadIdx.x;
  if (i >= QUADS_PER_BAND) return;
  const int q = band * QUADS_PER_BAND + i;                 // = y * 21 + (x >> 2): the generator's word index
  const long quad = (long)e * (OBS_PIX / 4) + q;
  // the old stack is requested whether or not this step resets the environment (1 step in 100 reads 16 bytes for nothing):
  // the reset flag hangs on the step id, which the caller has just LOADED -- asked for only when needed, the two round
  // trips to memory ran one after the other in every workgroup of the launch
  uint4 old = force_reset ? make_uint4(0u, 0u, 0u, 0u) : reinterpret_cast<const uint4*>(stack_in)[quad];
  const uint32_t key = synth_key(seed, env_offset + (uint32_t)e, id);
  const bool reset = force_reset || lowbias32(key ^ 0x3C6EF372u) < thresh;
  if (reset) old = make_uint4(0u, 0u, 0u, 0u);
  const uint32_t w = synth_word(key, (uint32_t)q);
  const uint4 outv = make_uint4((old.x >> 8) | ((w & 255u) << 24), (old.y >> 8) | (((w >> 8) & 255u) << 24),
                                (old.z >> 8) | (((w >> 16) & 255u) << 24), (old.w >> 8) | ((w >> 24) << 24));
  reinterpret_cast<uint4*>(stack_out)[quad] = outv;
  if (stack_out2) reinterpret_cast<uint4*>(stack_out2)[quad] = outv;   // second copy (the observation ring's wrap-around slot)
}

// Path B: one of PRE_BANDS parts of environment e's raw 2 x 210 x 160 gray screen pair for step `id` (16,800 generator
// words = 4,200 16-byte stores per environment, 600 per unit).  unit = env * PRE_BANDS + part; 256 threads.
__device__ __forceinline__ void synth_raw_unit(uint64_t seed, uint32_t env_offset, uint64_t id, int unit,
                                               uint32_t* __restrict__ raw) {
  const int e = unit / PRE_BANDS;
  const int part = unit % PRE_BANDS;
  const uint32_t rkey = synth_key(seed, env_offset + (uint32_t)e, id) ^ 0x5bd1e995u;
  constexpr int WORDS = 2 * PAAC_RAW_H * PAAC_RAW_W / 4;   // 16800
  constexpr int PER = WORDS / 4 / PRE_BANDS;               // 600
  uint4* out = reinterpret_cast<uint4*>(raw + (long)e * WORDS);
  for (int q = part * PER + (int)threadIdx.x; q < (part + 1) * PER; q += 256)
    out[q] = make_uint4(synth_word(rkey, (uint32_t)(4 * q)), synth_word(rkey, (uint32_t)(4 * q + 1)),
                        synth_word(rkey, (uint32_t)(4 * q + 2)), synth_word(rkey, (uint32_t)(4 * q + 3)));
}

}  // namespace paac
