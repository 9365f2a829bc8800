// Actor / critic heads of the PAAC network on gfx950 (policy_v_network.py:24-57): forward (fc split-K slab
// reduction + bias + ReLU, two tiny GEMVs, softmax, optional counter-based categorical sampling) and the loss
// gradient wrt the heads.  HBM/latency-bound kernels: every independent global load is issued before anything
// is consumed, per-thread arrays are sized by a compile-time action-count bucket AP (4 / 8 / 20 / 32) so they stay
// in registers, and block-wide sums go through LDS in two short stages instead of long ds_bpermute chains.
#pragma once
#include "common.h"
#include "synth_dev.h"

namespace paac {

constexpr int FC_SPLITS_MAX = 8;   // fc forward split-K slabs (summed here)
constexpr int MAXA = 32;

#ifdef PAAC_DMM_STAMPS
__device__ unsigned long long* g_stamps_dev = nullptr;
#define HEADS_STAMP_INIT() unsigned long long* stamp_ptr = g_stamps_dev
#define HEADS_STAMP(i)                                                                                         \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (stamp_ptr && threadIdx.x == 0) stamp_ptr[(long)blockIdx.x * 8 + (i)] = (unsigned long long)clock64(); \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#else
#define HEADS_STAMP_INIT()
#define HEADS_STAMP(i)
#endif

// Block-wide sums (256 threads) of NV per-thread values: out_lds[a] = sum over threads of vals[a].
// scratch: NV*256 + NV*8 floats of LDS.  Ends with a barrier.
template <int NV>
__device__ __forceinline__ void block_sums_256(const float (&vals)[NV], float* scratch, float* out_lds) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int a = 0; a < NV; ++a) scratch[a * 256 + tid] = vals[a];
  __syncthreads();
  float* stage = scratch + NV * 256;
  for (int u = tid; u < NV * 8; u += 256) {
    const int a = u >> 3, c = u & 7;
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = scratch[a * 256 + c * 32 + ((i + c) & 31)];   // rotated start: spread the banks
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += v[i];
    stage[u] = t;
  }
  __syncthreads();
  if (tid < NV) {
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) t += stage[tid * 8 + c];
    out_lds[tid] = t;
  }
  __syncthreads();
}

struct PhiloxArgs {
  int enabled;
  uint64_t seed;
  const uint64_t* step_base;
  uint64_t step_off;
  uint32_t env_offset;
  int32_t* actions;
};

// Counter-based-sampler mode only: the step of the device-resident synthetic environments inside the heads launch.
// Row i's workgroup samples its own action, so it also does environment i's bookkeeping (reward clip, mask, episode
// totals) -- nothing crosses workgroups -- and workgroups [batch, batch + batch*PRE_BANDS) shift the observation
// stacks (the new frame does not depend on the action).
struct SynthStepArgs {
  int enabled;
  uint64_t seed;
  uint32_t thresh;
  const uint32_t* stack_in;
  uint32_t* stack_out;
  float *rewards, *masks, *ep_reward;
  int32_t* ep_len;
  FinishedRing* fin;
};

__device__ __forceinline__ uint32_t philox_word0(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c0;
}

// One 256-thread workgroup per batch row.
template <int H, int AP>
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ slab, int splits, long slab_stride,
                                                        const float* __restrict__ fc_b, const float* __restrict__ Wa,
                                                        const float* __restrict__ ba, const float* __restrict__ Wc,
                                                        const float* __restrict__ bc, int A, float* __restrict__ h_out,
                                                        float* __restrict__ logits_ws, float* __restrict__ probs_ws,
                                                        float* __restrict__ values_ws, float* __restrict__ logits_out,
                                                        float* __restrict__ probs_out, float* __restrict__ values_out,
                                                        const PhiloxArgs ph, const int batch, const SynthStepArgs st) {
  constexpr int JPT = H / 256;
  constexpr int NV = AP + 1;                 // A logits (padded) + value
  const int i = blockIdx.x;
  const int tid = threadIdx.x;
  float ep_reward0 = 0.f;
  int32_t ep_len0 = 0;
  if (st.enabled) {
    if (i >= batch) {
      const uint64_t id = (ph.step_base ? *ph.step_base : 0ull) + ph.step_off + 1ull;
      synth_shift_band(st.seed, ph.env_offset, id, st.thresh, i - batch, st.stack_in, st.stack_out);
      return;
    }
    if (tid == 0) {                          // the running totals do not depend on the action: request them now
      ep_reward0 = st.ep_reward[i];
      ep_len0 = st.ep_len[i];
    }
  }
  HEADS_STAMP_INIT();
  HEADS_STAMP(0);
  // ---- every independent load first ---------------------------------------------------------------
  const float head_bias = (tid < A) ? ba[tid] : ((tid == AP) ? bc[0] : 0.f);
  const uint64_t step0 = (ph.enabled && ph.step_base && tid == 0) ? *ph.step_base : 0ull;
  float sv[JPT][FC_SPLITS_MAX], bj[JPT], wcj[JPT], waj[JPT][AP];
#pragma unroll
  for (int jj = 0; jj < JPT; ++jj) {
    const int j = tid + jj * 256;
#pragma unroll
    for (int sp = 0; sp < FC_SPLITS_MAX; ++sp) sv[jj][sp] = slab[(sp < splits ? sp : 0) * slab_stride + (long)i * H + j];
    bj[jj] = fc_b[j];
    wcj[jj] = Wc[j];
#pragma unroll
    for (int a = 0; a < AP; ++a) waj[jj][a] = Wa[j * A + (a < A ? a : 0)];
  }
  float part[NV];
#pragma unroll
  for (int a = 0; a < NV; ++a) part[a] = 0.f;
#pragma unroll
  for (int jj = 0; jj < JPT; ++jj) {
    const int j = tid + jj * 256;
    float s = 0.f;
#pragma unroll
    for (int sp = 0; sp < FC_SPLITS_MAX; ++sp) s += (sp < splits) ? sv[jj][sp] : 0.f;
    s = fmaxf(s + bj[jj], 0.f);
    h_out[(long)i * H + j] = s;
#pragma unroll
    for (int a = 0; a < AP; ++a) part[a] = (a < A) ? fmaf(s, waj[jj][a], part[a]) : part[a];
    part[AP] = fmaf(s, wcj[jj], part[AP]);
  }
  HEADS_STAMP(1);
  __shared__ float scratch[NV * 256 + NV * 8];
  __shared__ float lg[NV];
  block_sums_256<NV>(part, scratch, lg);
  HEADS_STAMP(2);
  if (tid < NV) lg[tid] += head_bias;
  __syncthreads();
  HEADS_STAMP(3);
  if (tid == 0) {
    float m = lg[0];
#pragma unroll
    for (int a = 1; a < AP; ++a) m = (a < A) ? fmaxf(m, lg[a]) : m;
    float e[AP];
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < AP; ++a) {
      e[a] = (a < A) ? expf(lg[a] - m) : 0.f;
      sum += e[a];
    }
    float u = 0.f;
    if (ph.enabled) {
      const uint64_t step = step0 + ph.step_off;
      const uint32_t w = philox_word0(ph.env_offset + (uint32_t)i, (uint32_t)step, (uint32_t)(step >> 32), 0u,
                                      (uint32_t)ph.seed, (uint32_t)(ph.seed >> 32));
      u = (float)(w >> 8) * (1.0f / 16777216.0f);
    }
    int act = A - 1;
    bool found = false;
    float cum = 0.f;
#pragma unroll
    for (int a = 0; a < AP; ++a)
      if (a < A) {
        const float pa = e[a] / sum;
        probs_ws[(long)i * A + a] = pa;
        logits_ws[(long)i * A + a] = lg[a];
        if (probs_out) probs_out[(long)i * A + a] = pa;
        if (logits_out) logits_out[(long)i * A + a] = lg[a];
        if (a < A - 1) {
          cum += pa;
          if (!found && u < cum) {
            act = a;
            found = true;
          }
        }
      }
    values_ws[i] = lg[AP];
    if (values_out) values_out[i] = lg[AP];
    if (ph.enabled) ph.actions[i] = act;
    if (st.enabled) {
      const uint64_t id = step0 + ph.step_off + 1ull;
      const uint32_t key = synth_key(st.seed, ph.env_offset + (uint32_t)i, id);
      synth_bookkeep_with(key, i, act, st.thresh, ep_reward0, ep_len0, st.rewards, st.masks, st.ep_reward, st.ep_len, st.fin);
    }
  }
  HEADS_STAMP(4);
}

// Loss gradient wrt the heads of one row (policy_v_network.py:29-57; analytic form: DESIGN.md / SURVEY A.4):
//   s = 5/B; dv = s*0.5*(v - y); g_a = -(adv*1[a=act]/(pi_a+eps) - beta*(log(pi_a+eps) + pi_a/(pi_a+eps)))
//   dlogit_a = s*pi_a*(g_a - sum_j g_j pi_j)
// out[0..AP) = dlogits (0 beyond A), out[AP] = dv.  stats (optional): actor term, critic term, entropy.
template <int AP>
__device__ __forceinline__ void head_grad_row(const float (&pi)[AP], float v, int act, float y, float adv, float beta,
                                              float s, int A, float (&out)[AP + 1], float* stats) {
#pragma clang fp contract(off)   // inlined into several kernels that must produce the same bits: no context-dependent fma
  const float eps = 1e-30f;
  float g[AP];
  float dot = 0.f, ent = 0.f, logp = 0.f;
#pragma unroll
  for (int a = 0; a < AP; ++a) {
    const bool on = a < A;
    const float p = on ? pi[a] : 1.f;
    const float lp = logf(p + eps);
    const float inv = 1.0f / (p + eps);
    const float oh = (a == act) ? 1.f : 0.f;
    g[a] = on ? -(adv * oh * inv - beta * (lp + p * inv)) : 0.f;
    dot += on ? g[a] * p : 0.f;
    ent -= on ? p * lp : 0.f;
    logp += on ? oh * lp : 0.f;
  }
#pragma unroll
  for (int a = 0; a < AP; ++a) out[a] = (a < A) ? s * pi[a] * (g[a] - dot) : 0.f;
  out[AP] = s * 0.5f * (v - y);
  if (stats) {
    stats[0] = -(logp * adv + beta * ent);   // actor objective term
    stats[1] = 0.25f * (y - v) * (y - v);    // critic term
    stats[2] = ent;
  }
}

// The n-step returns of the rollout (paac.py:140-149) computed where they are consumed: with v_boot set, the heads
// gradient launch derives y / adv of a row itself from the rollout records (every consumer recomputes the short scan:
// cheaper than a launch of its own), its last block writes the y / adv arrays for the learner's records and does the
// per-cycle bookkeeping of paac_nstep_returns_tick (global_step, lr, frame counter).  Arithmetic = nstep_returns_kernel's
// (csrc/misc.hip), operation for operation.
struct ReturnsArgs {
  const float* v_boot;      // nullptr: y / adv are read from the arrays passed to the kernel (unless boot_in_fwd)
  int boot_in_fwd;          // the training forward covered T*N + N rows: bootstrap value of environment e = value head of
                            // forward row T*N + e (heads_train_kernel computes it itself; heads_bwd_kernel gets v_boot set)
  const float* rewards;     // [T,N] clipped
  const float* masks;       // [T,N]
  const float* values_act;  // [T,N] values of the acting forwards
  int T, N;
  double gamma;
  float* y_out;             // [T*N] t-major, written by the last block
  float* adv_out;
  int64_t* global_step;     // += step_inc, then lr = f32(lr0 - step*lr0/anneal)   (actor_learner.py:119-123); nullable
  int64_t step_inc;
  double lr0;
  int64_t anneal;
  float* lr_out;
  uint64_t* tick;           // += tick_inc; nullable
  uint64_t tick_inc;
};

__device__ __forceinline__ void nstep_row_from(const ReturnsArgs& r, const int i, const float vb, float& y, float& adv) {
  const int t = i / r.N, e = i - t * r.N;
  double R = 0.0;
  for (int tt = r.T - 1; tt >= t; --tt) {
    const long k = (long)tt * r.N + e;
    const double prod = (tt == r.T - 1) ? (double)__fmul_rn((float)r.gamma, vb) : __dmul_rn(r.gamma, R);
    R = __dadd_rn((double)r.rewards[k], __dmul_rn(prod, (double)r.masks[k]));
  }
  y = (float)R;
  adv = (float)__dsub_rn(R, (double)r.values_act[(long)t * r.N + e]);
}
// The same scan with the row's rewards / masks / acting value already in registers (kNstepPre steps from the end of the
// rollout; older steps, if the rollout is longer, are loaded here): the loads inside the loop are a chain of T round trips
// to memory on the one thread that runs it.
constexpr int kNstepPre = 8;
struct NstepPre {
  float rw[kNstepPre], mk[kNstepPre], vact;
};
__device__ __forceinline__ NstepPre nstep_preload(const ReturnsArgs& r, const int i) {
  NstepPre p;
  const int t = i / r.N, e = i - t * r.N;
#pragma unroll
  for (int k = 0; k < kNstepPre; ++k) {
    const int tt = r.T - 1 - k;
    const long at = (long)(tt >= t ? tt : t) * r.N + e;
    p.rw[k] = r.rewards[at];
    p.mk[k] = r.masks[at];
  }
  p.vact = r.values_act[(long)t * r.N + e];
  return p;
}
__device__ __forceinline__ void nstep_row_from(const ReturnsArgs& r, const int i, const float vb, const NstepPre& p, float& y,
                                               float& adv) {
  const int t = i / r.N, e = i - t * r.N;
  double R = 0.0;
#pragma unroll
  for (int k = 0; k < kNstepPre; ++k) {
    const int tt = r.T - 1 - k;
    if (tt >= t) {
      const double prod = (k == 0) ? (double)__fmul_rn((float)r.gamma, vb) : __dmul_rn(r.gamma, R);
      R = __dadd_rn((double)p.rw[k], __dmul_rn(prod, (double)p.mk[k]));
    }
  }
  for (int tt = r.T - 1 - kNstepPre; tt >= t; --tt) {
    const long k = (long)tt * r.N + e;
    R = __dadd_rn((double)r.rewards[k], __dmul_rn(__dmul_rn(r.gamma, R), (double)r.masks[k]));
  }
  y = (float)R;
  adv = (float)__dsub_rn(R, (double)p.vact);
}
__device__ __forceinline__ void nstep_row(const ReturnsArgs& r, const int i, float& y, float& adv) {
  nstep_row_from(r, i, r.v_boot[i % r.N], y, adv);
}

template <int AP>
__device__ __forceinline__ void load_row_and_grad(const float* __restrict__ probs, const float* __restrict__ values,
                                                  const int32_t* __restrict__ actions, const float* __restrict__ y,
                                                  const float* __restrict__ adv, int i, int A, float beta, float s,
                                                  float (&out)[AP + 1], float* stats, const ReturnsArgs& rt,
                                                  float* yv_out = nullptr, float* av_out = nullptr) {
  float pi[AP];
#pragma unroll
  for (int a = 0; a < AP; ++a) pi[a] = probs[(long)i * A + (a < A ? a : 0)];
  float yv, av;
  if (rt.v_boot) {
    nstep_row(rt, i, yv, av);
  } else {
    yv = y[i];
    av = adv[i];
  }
  if (yv_out) {
    *yv_out = yv;
    *av_out = av;
  }
  head_grad_row<AP>(pi, values[i], actions[i], yv, av, beta, s, A, out, stats);
}

// One launch, three roles by blockIdx:
//   [0, B)          row i -> dH[i,:] = (dlogits Wa^T + dv Wc^T) * 1[h > 0]
//   [B, B + H/32)   head weight gradients for 32 columns j: 8 row-groups x 32 columns per workgroup, dlogits
//                   recomputed into LDS in chunks of 256 rows, h rows loaded in batches
//   B + H/32        head bias gradients + loss scalars
constexpr int HB_CHUNK = 256;
template <int H, int AP>
__global__ __launch_bounds__(256) void heads_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ values,
                                                        const int32_t* __restrict__ actions, const float* __restrict__ y,
                                                        const float* __restrict__ adv, const float* __restrict__ h,
                                                        const float* __restrict__ Wa, const float* __restrict__ Wc,
                                                        int A, int B, float beta, float* __restrict__ dH,
                                                        float* __restrict__ gWa, float* __restrict__ gba,
                                                        float* __restrict__ gWc, float* __restrict__ gbc,
                                                        float* __restrict__ loss_out, const ReturnsArgs rt) {
  constexpr int NV = AP + 1;
  constexpr int NS = NV + 3;                       // + 3 loss statistics (role 3)
  const int tid = threadIdx.x;
  const float s = 5.0f / (float)B;
  HEADS_STAMP_INIT();
  HEADS_STAMP(0);
  __shared__ float smem[NS * 256 + NS * 8 > HB_CHUNK * NV ? NS * 256 + NS * 8 : HB_CHUNK * NV];
  __shared__ float red[NS];
  if ((int)blockIdx.x < B) {
    // ---- role 1 ----
    const int i = blockIdx.x;
    constexpr int JPT = H / 256;
    float hv[JPT], wcj[JPT], waj[JPT][AP];
#pragma unroll
    for (int jj = 0; jj < JPT; ++jj) {
      const int j = tid + jj * 256;
      hv[jj] = h[(long)i * H + j];
      wcj[jj] = Wc[j];
#pragma unroll
      for (int a = 0; a < AP; ++a) waj[jj][a] = Wa[j * A + (a < A ? a : 0)];
    }
    float dl[NV];
    load_row_and_grad<AP>(probs, values, actions, y, adv, i, A, beta, s, dl, nullptr, rt);   // every thread: same row
#pragma unroll
    for (int jj = 0; jj < JPT; ++jj) {
      const int j = tid + jj * 256;
      float acc = dl[AP] * wcj[jj];
#pragma unroll
      for (int a = 0; a < AP; ++a) acc = fmaf(dl[a], waj[jj][a], acc);   // explicit: the two kernels with this line must round alike
      dH[(long)i * H + j] = hv[jj] > 0.f ? acc : 0.f;
    }
    HEADS_STAMP(1);
    return;
  }
  const int role = blockIdx.x - B;
  if (role < H / 32) {
    // ---- role 2 ----
    const int jj = tid & 31, ig = tid >> 5;
    const int j = role * 32 + jj;
    float acc[NV];
#pragma unroll
    for (int a = 0; a < NV; ++a) acc[a] = 0.f;
    for (int i0 = 0; i0 < B; i0 += HB_CHUNK) {
      const int cnt = min(HB_CHUNK, B - i0);
      float hv[HB_CHUNK / 8];
#pragma unroll
      for (int q = 0; q < HB_CHUNK / 8; ++q) {       // issue the whole batch of h loads first
        const int r = ig + 8 * q;
        hv[q] = h[(long)(i0 + (r < cnt ? r : 0)) * H + j];
      }
      __syncthreads();
      if (tid < cnt) {
        float dl[NV];
        load_row_and_grad<AP>(probs, values, actions, y, adv, i0 + tid, A, beta, s, dl, nullptr, rt);
#pragma unroll
        for (int a = 0; a < NV; ++a) smem[tid * NV + a] = dl[a];
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < HB_CHUNK / 8; ++q) {
        const int r = ig + 8 * q;
        if (r < cnt) {
#pragma unroll
          for (int a = 0; a < NV; ++a) acc[a] = fmaf(hv[q], smem[r * NV + a], acc[a]);
        }
      }
    }
    HEADS_STAMP(2);
    __syncthreads();
    // reduce the 8 row-groups through LDS: smem as [8][32][NV]
#pragma unroll
    for (int a = 0; a < NV; ++a) smem[(ig * 32 + jj) * NV + a] = acc[a];
    __syncthreads();
    for (int u = tid; u < 32 * NV; u += 256) {
      const int c = u / NV, a = u - c * NV;
      float v = 0.f;
#pragma unroll
      for (int g8 = 0; g8 < 8; ++g8) v += smem[(g8 * 32 + c) * NV + a];
      const int jo = role * 32 + c;
      if (a == AP) gWc[jo] = v;
      else if (a < A) gWa[jo * A + a] = v;
    }
    HEADS_STAMP(3);
    return;
  }
  // ---- role 3: bias gradients + loss scalars ----
  float accv[NS];
#pragma unroll
  for (int a = 0; a < NS; ++a) accv[a] = 0.f;
  for (int i = tid; i < B; i += 256) {
    float dl[NV], stats[3];
    float yv, av;
    load_row_and_grad<AP>(probs, values, actions, y, adv, i, A, beta, s, dl, stats, rt, &yv, &av);
    if (rt.v_boot) {          // the learner's records of the returns (paac.py:151-154 feed layout)
      rt.y_out[i] = yv;
      rt.adv_out[i] = av;
    }
#pragma unroll
    for (int a = 0; a < NV; ++a) accv[a] += dl[a];
    accv[NV] += stats[0]; accv[NV + 1] += stats[1]; accv[NV + 2] += stats[2];
  }
  if (rt.v_boot && tid == 0) {      // per-cycle bookkeeping (paac.py:127, actor_learner.py:119-123)
    if (rt.global_step) {
      const int64_t step = *rt.global_step + rt.step_inc;
      *rt.global_step = step;
      double lr = 0.0;
      if (step <= rt.anneal) lr = rt.lr0 - ((double)step * rt.lr0 / (double)rt.anneal);
      *rt.lr_out = (float)lr;
    }
    if (rt.tick) *rt.tick += rt.tick_inc;
  }
  block_sums_256<NS>(accv, smem, red);
  if (tid < A) gba[tid] = red[tid];
  if (tid == AP) gbc[0] = red[AP];
  if (tid == 0 && loss_out) {
    const float actor = red[NV] / (float)B;
    const float critic = red[NV + 1] / (float)B;
    loss_out[0] = 5.0f * (actor + critic);
    loss_out[1] = actor;
    loss_out[2] = critic;
    loss_out[3] = red[NV + 2] / (float)B;
  }
  HEADS_STAMP(4);
}

// ---------------------------------------------------------------------------------------------
// Training update with the heads FORWARD folded into the heads gradient launch (the training forward stopped after the fc
// layer's split-K slabs: paac_train_forward_trunk).  Row i's workgroup finishes the forward of its own row (slab sums,
// bias, ReLU, the two head contractions, softmax) with heads_fwd_kernel's arithmetic, and -- when the returns are computed
// here -- also the value head of its environment's bootstrap row T*N + e (one more 512-long contraction: cheaper than a
// launch in between), then the n-step scan, the head gradient of the row and dH, like role 1 of heads_bwd_kernel.  It leaves
// dl[row][AP + 1] and the three loss terms per row behind; the reductions over rows (head weight / bias gradients, loss
// scalars: roles 2 and 3) need every row and ride as extra workgroups of a later launch (heads_param_grads, called from
// dgrad_tower_kernel).  One launch less per update; every value bit-identical to the separate launches.
static_assert(kDlStride == MAXA + 1 + 3, "floats per row in the dl buffer: AP + 1 gradients (padded to 33) + 3 loss terms");

template <int H, int AP>
__global__ __launch_bounds__(256) void heads_train_kernel(const float* __restrict__ slab, int splits, long slab_stride,
                                                          const float* __restrict__ fc_b, const float* __restrict__ Wa,
                                                          const float* __restrict__ ba, const float* __restrict__ Wc,
                                                          const float* __restrict__ bc, int A, int B,
                                                          float* __restrict__ h_out, float* __restrict__ logits_ws,
                                                          float* __restrict__ probs_ws, float* __restrict__ values_ws,
                                                          const int32_t* __restrict__ actions, const float* __restrict__ y,
                                                          const float* __restrict__ adv, float beta, float* __restrict__ dH,
                                                          float* __restrict__ dl_buf, const ReturnsArgs rt) {
  constexpr int JPT = H / 256;
  constexpr int NV = AP + 1;                 // A logits (padded) + value
  constexpr int NB = NV + 1;                 // + the bootstrap row's value
  const int i = blockIdx.x;
  const int tid = threadIdx.x;
  const bool boot = rt.boot_in_fwd != 0;
  const int ib = boot ? B + i % rt.N : i;    // forward row holding this environment's bootstrap observation
  // ---- every independent load first ---------------------------------------------------------------
  const float head_bias = (tid < A) ? ba[tid] : ((tid == AP || tid == NV) ? bc[0] : 0.f);
  float sv[JPT][FC_SPLITS_MAX], sb[JPT][FC_SPLITS_MAX], bj[JPT], wcj[JPT], waj[JPT][AP];
#pragma unroll
  for (int jj = 0; jj < JPT; ++jj) {
    const int j = tid + jj * 256;
    if (splits == 1) {         // one slab (rows kept by the acting forwards, or an unsplit fc layer): one load, not eight
      sv[jj][0] = slab[(long)i * H + j];
      sb[jj][0] = slab[(long)ib * H + j];
#pragma unroll
      for (int sp = 1; sp < FC_SPLITS_MAX; ++sp) sv[jj][sp] = sb[jj][sp] = 0.f;
    } else {
#pragma unroll
      for (int sp = 0; sp < FC_SPLITS_MAX; ++sp) {
        sv[jj][sp] = slab[(sp < splits ? sp : 0) * slab_stride + (long)i * H + j];
        sb[jj][sp] = slab[(sp < splits ? sp : 0) * slab_stride + (long)ib * H + j];
      }
    }
    bj[jj] = fc_b[j];
    wcj[jj] = Wc[j];
#pragma unroll
    for (int a = 0; a < AP; ++a) waj[jj][a] = Wa[j * A + (a < A ? a : 0)];
  }
  const int act = actions[i];
  // (the row's n-step scan runs on thread 0 further down: its inputs are wave-uniform addresses, requested here)
  NstepPre npre;
  if (rt.rewards) npre = nstep_preload(rt, i);
  float part[NB], hv[JPT];
#pragma unroll
  for (int a = 0; a < NB; ++a) part[a] = 0.f;
#pragma unroll
  for (int jj = 0; jj < JPT; ++jj) {
    const int j = tid + jj * 256;
    float s = 0.f, sbt = 0.f;
#pragma unroll
    for (int sp = 0; sp < FC_SPLITS_MAX; ++sp) {
      s += (sp < splits) ? sv[jj][sp] : 0.f;
      sbt += (sp < splits) ? sb[jj][sp] : 0.f;
    }
    s = fmaxf(s + bj[jj], 0.f);
    sbt = fmaxf(sbt + bj[jj], 0.f);
    hv[jj] = s;
    h_out[(long)i * H + j] = s;
#pragma unroll
    for (int a = 0; a < AP; ++a) part[a] = (a < A) ? fmaf(s, waj[jj][a], part[a]) : part[a];
    part[AP] = fmaf(s, wcj[jj], part[AP]);
    part[NV] = fmaf(sbt, wcj[jj], part[NV]);
  }
  __shared__ float scratch[NB * 256 + NB * 8];
  __shared__ float lg[NB];
  __shared__ float row_s[AP + 3];            // probabilities, then y, adv
  block_sums_256<NB>(part, scratch, lg);
  if (tid < NB) lg[tid] += head_bias;
  __syncthreads();
  if (tid == 0) {
    float m = lg[0];
#pragma unroll
    for (int a = 1; a < AP; ++a) m = (a < A) ? fmaxf(m, lg[a]) : m;
    float e[AP];
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < AP; ++a) {
      e[a] = (a < A) ? expf(lg[a] - m) : 0.f;
      sum += e[a];
    }
#pragma unroll
    for (int a = 0; a < AP; ++a) {
      const float pa = (a < A) ? e[a] / sum : 0.f;
      row_s[a] = pa;
      if (a < A) {
        probs_ws[(long)i * A + a] = pa;
        logits_ws[(long)i * A + a] = lg[a];
      }
    }
    values_ws[i] = lg[AP];
    float yv, av;
    if (rt.rewards) {
      nstep_row_from(rt, i, boot ? lg[NV] : rt.v_boot[i % rt.N], npre, yv, av);
      rt.y_out[i] = yv;                      // the learner's records of the returns (paac.py:151-154 feed layout)
      rt.adv_out[i] = av;
      if (boot && i < rt.N) values_ws[B + i] = lg[NV];
    } else {
      yv = y[i];
      av = adv[i];
    }
    row_s[AP] = yv;
    row_s[AP + 1] = av;
    if (i == 0 && rt.rewards) {              // per-cycle bookkeeping (paac.py:127, actor_learner.py:119-123)
      if (rt.global_step) {
        const int64_t step = *rt.global_step + rt.step_inc;
        *rt.global_step = step;
        double lr = 0.0;
        if (step <= rt.anneal) lr = rt.lr0 - ((double)step * rt.lr0 / (double)rt.anneal);
        *rt.lr_out = (float)lr;
      }
      if (rt.tick) *rt.tick += rt.tick_inc;
    }
  }
  __syncthreads();
  float pi[AP], dl[NV], stats[3];
#pragma unroll
  for (int a = 0; a < AP; ++a) pi[a] = row_s[a];
  head_grad_row<AP>(pi, lg[AP], act, row_s[AP], row_s[AP + 1], beta, 5.0f / (float)B, A, dl, stats);   // every thread: same row
  if (tid == 0) {
#pragma unroll
    for (int a = 0; a < NV; ++a) dl_buf[(long)i * kDlStride + a] = dl[a];
#pragma unroll
    for (int k = 0; k < 3; ++k) dl_buf[(long)i * kDlStride + MAXA + 1 + k] = stats[k];
  }
#pragma unroll
  for (int jj = 0; jj < JPT; ++jj) {
    const int j = tid + jj * 256;
    float acc = dl[AP] * wcj[jj];
#pragma unroll
    for (int a = 0; a < AP; ++a) acc = fmaf(dl[a], waj[jj][a], acc);   // explicit: the two kernels with this line must round alike
    dH[(long)i * H + j] = hv[jj] > 0.f ? acc : 0.f;
  }
}

// Roles 2 and 3 of heads_bwd_kernel with the per-row head gradients read from dl_buf (written by heads_train_kernel in an
// earlier launch): role in [0, H/32) = head weight gradients of 32 fc columns; role == H/32 = head bias gradients + loss
// scalars.  256 threads; smem: at least max((AP + 4) * 264, HB_CHUNK * (AP + 1)) floats.  Same summation orders as there.
template <int H, int AP>
__device__ __forceinline__ void heads_param_grads(const int role, const float* __restrict__ h,
                                                  const float* __restrict__ dl_buf, const int A, const int B,
                                                  float* __restrict__ gWa, float* __restrict__ gba, float* __restrict__ gWc,
                                                  float* __restrict__ gbc, float* __restrict__ loss_out, float* smem) {
  constexpr int NV = AP + 1;
  constexpr int NS = NV + 3;
  const int tid = threadIdx.x;
  if (role < H / 32) {
    const int jj = tid & 31, ig = tid >> 5;
    const int j = role * 32 + jj;
    float acc[NV];
#pragma unroll
    for (int a = 0; a < NV; ++a) acc[a] = 0.f;
    for (int i0 = 0; i0 < B; i0 += HB_CHUNK) {
      const int cnt = min(HB_CHUNK, B - i0);
      float hv[HB_CHUNK / 8];
#pragma unroll
      for (int q = 0; q < HB_CHUNK / 8; ++q) {       // issue the whole batch of h loads first
        const int r = ig + 8 * q;
        hv[q] = h[(long)(i0 + (r < cnt ? r : 0)) * H + j];
      }
      __syncthreads();
      if (tid < cnt) {
#pragma unroll
        for (int a = 0; a < NV; ++a) smem[tid * NV + a] = dl_buf[(long)(i0 + tid) * kDlStride + a];
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < HB_CHUNK / 8; ++q) {
        const int r = ig + 8 * q;
        if (r < cnt) {
#pragma unroll
          for (int a = 0; a < NV; ++a) acc[a] = fmaf(hv[q], smem[r * NV + a], acc[a]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NV; ++a) smem[(ig * 32 + jj) * NV + a] = acc[a];
    __syncthreads();
    for (int u = tid; u < 32 * NV; u += 256) {
      const int c = u / NV, a = u - c * NV;
      float v = 0.f;
#pragma unroll
      for (int g8 = 0; g8 < 8; ++g8) v += smem[(g8 * 32 + c) * NV + a];
      const int jo = role * 32 + c;
      if (a == AP) gWc[jo] = v;
      else if (a < A) gWa[jo * A + a] = v;
    }
    return;
  }
  float accv[NS];
#pragma unroll
  for (int a = 0; a < NS; ++a) accv[a] = 0.f;
  for (int i = tid; i < B; i += 256) {
#pragma unroll
    for (int a = 0; a < NV; ++a) accv[a] += dl_buf[(long)i * kDlStride + a];
#pragma unroll
    for (int k = 0; k < 3; ++k) accv[NV + k] += dl_buf[(long)i * kDlStride + MAXA + 1 + k];
  }
  float* red = smem + NS * 256 + NS * 8;
  block_sums_256<NS>(accv, smem, red);
  if (tid < A) gba[tid] = red[tid];
  if (tid == AP) gbc[0] = red[AP];
  if (tid == 0 && loss_out) {
    const float actor = red[NV] / (float)B;
    const float critic = red[NV + 1] / (float)B;
    loss_out[0] = 5.0f * (actor + critic);
    loss_out[1] = actor;
    loss_out[2] = critic;
    loss_out[3] = red[NV + 2] / (float)B;
  }
}
struct HeadsGradArgs {        // extra workgroups of dgrad_tower_kernel; blocks == 0: none
  const float* h;
  const float* dl_buf;
  float *gWa, *gba, *gWc, *gbc, *loss_out;
  int A, B, blocks;
};

// Dispatch on the action-count bucket.
template <int H, class... Args>
inline void launch_heads_fwd(int A, dim3 grid, hipStream_t s, Args... args) {
  if (A <= 4) launch_k(heads_fwd_kernel<H, 4>, grid, dim3(256), s, PROF_WHOLE, args...);
  else if (A <= 8) launch_k(heads_fwd_kernel<H, 8>, grid, dim3(256), s, PROF_WHOLE, args...);
  else if (A <= 20) launch_k(heads_fwd_kernel<H, 20>, grid, dim3(256), s, PROF_WHOLE, args...);
  else launch_k(heads_fwd_kernel<H, 32>, grid, dim3(256), s, PROF_WHOLE, args...);
}
template <int H, class... Args>
inline void launch_heads_train(int A, dim3 grid, hipStream_t s, Args... args) {
  if (A <= 4) launch_k(heads_train_kernel<H, 4>, grid, dim3(256), s, PROF_WHOLE, args...);
  else if (A <= 8) launch_k(heads_train_kernel<H, 8>, grid, dim3(256), s, PROF_WHOLE, args...);
  else if (A <= 20) launch_k(heads_train_kernel<H, 20>, grid, dim3(256), s, PROF_WHOLE, args...);
  else launch_k(heads_train_kernel<H, 32>, grid, dim3(256), s, PROF_WHOLE, args...);
}
template <int H, class... Args>
inline void launch_heads_bwd(int A, dim3 grid, hipStream_t s, Args... args) {
  if (A <= 4) launch_k(heads_bwd_kernel<H, 4>, grid, dim3(256), s, PROF_WHOLE, args...);
  else if (A <= 8) launch_k(heads_bwd_kernel<H, 8>, grid, dim3(256), s, PROF_WHOLE, args...);
  else if (A <= 20) launch_k(heads_bwd_kernel<H, 20>, grid, dim3(256), s, PROF_WHOLE, args...);
  else launch_k(heads_bwd_kernel<H, 32>, grid, dim3(256), s, PROF_WHOLE, args...);
}

}  // namespace paac
