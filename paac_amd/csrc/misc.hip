// HBM-bound kernels of the PAAC hot path on gfx950: n-step returns, lr schedule, samplers,
// frame preprocessing / stacking, device-resident synthetic environments, clip + RMSProp.
#include "common.h"
#include "synth_dev.h"
#include "fc_heads.h"
#include "tower.h"

namespace paac {

#ifdef PAAC_DMM_STAMPS
__device__ unsigned long long* g_misc_stamps = nullptr;   // diagnostic build: phase stamps of the sampler workgroup
#define MISC_STAMP(i)                                                                       \
  do {                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    if (g_misc_stamps && threadIdx.x == 0) g_misc_stamps[(i)] = (unsigned long long)wall_clock64(); \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)
#else
#define MISC_STAMP(i)
#endif

// =============================================================================================
// n-step returns (paac.py:140-149), fp64 scan like the reference's numpy buffers.
struct CycleTick {          // optional bookkeeping folded into the returns kernel (one launch instead of three)
  int64_t* global_step;     // += step_inc, then lr = f32(lr0 - step*lr0/anneal)   (actor_learner.py:119-123)
  int64_t step_inc;
  double lr0;
  int64_t anneal;
  float* lr_out;
  uint64_t* tick;           // += tick_inc (sampler / synthetic-env frame counter)
  uint64_t tick_inc;
};

__global__ void nstep_returns_kernel(const float* __restrict__ v_boot, const float* __restrict__ rewards,
                                     const float* __restrict__ masks, const float* __restrict__ values, int T, int N,
                                     double gamma, float* __restrict__ y, float* __restrict__ adv, const CycleTick ct) {
  // 128 threads: wave 0 scans 64 environments, wave 1 of workgroup 0 does the cycle bookkeeping -- its
  // read-modify-write round trip runs beside the scan's instead of in front of it
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  if (threadIdx.x >= 64) {
    if (blockIdx.x == 0 && threadIdx.x == 64) {
      if (ct.global_step) {
        const int64_t step = *ct.global_step + ct.step_inc;
        *ct.global_step = step;
        double lr = 0.0;
        if (step <= ct.anneal) lr = ct.lr0 - ((double)step * ct.lr0 / (double)ct.anneal);
        *ct.lr_out = (float)lr;
      }
      if (ct.tick) *ct.tick += ct.tick_inc;
    }
    return;
  }
  if (e >= N) return;
  // paac.py:146-147 as numpy evaluates it: estimated_return starts as the FLOAT32 network output, so the first
  // `gamma * estimated_return` is a float32 product (python float x float32 array -> float32, in numpy 1.x
  // and 2.x alike); it is promoted to float64 by `* masks[t]` and stays float64 afterwards.  Explicit
  // round-to-nearest mul/add (no FMA contraction) keep the scan bit-identical to numpy's.
  // The scan is serial in t but its inputs are not: each chunk of CH steps is requested at once (one memory round
  // trip per chunk instead of one per step -- at t_max = 5 the whole kernel is a single round trip).
  constexpr int CH = 8;
  const float vb = v_boot[e];
  double R = 0.0;
  for (int t0 = T - 1; t0 >= 0; t0 -= CH) {
    float r[CH], m[CH], v[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int t = t0 - u;
      const long i = (long)(t >= 0 ? t : 0) * N + e;
      r[u] = rewards[i];
      m[u] = masks[i];
      v[u] = values[i];
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int t = t0 - u;
      if (t >= 0) {
        const long i = (long)t * N + e;
        const double prod = (t == T - 1) ? (double)__fmul_rn((float)gamma, vb) : __dmul_rn(gamma, R);
        R = __dadd_rn((double)r[u], __dmul_rn(prod, (double)m[u]));
        y[i] = (float)R;
        adv[i] = (float)__dsub_rn(R, (double)v[u]);
      }
    }
  }
}

// actor_learner.py:119-123 evaluated after the cycle's increments (paac.py:127,156).
__global__ void lr_step_kernel(int64_t* global_step, int64_t inc, double lr0, int64_t anneal, float* lr_out) {
  const int64_t step = *global_step + inc;
  *global_step = step;
  double lr = 0.0;
  if (step <= anneal) lr = lr0 - ((double)step * lr0 / (double)anneal);
  *lr_out = (float)lr;
}

__global__ void counter_add_kernel(uint64_t* c, uint64_t inc) { *c += inc; }

// Diagnostic: {shader-clock ticks (s_memtime), constant 100 MHz ticks (s_memrealtime)} -- the quotient of two
// samples' differences is the shader clock the chip actually held in between.
__global__ void clock_probe_kernel(uint64_t* out) {
  out[0] = (uint64_t)clock64();
  out[1] = (uint64_t)wall_clock64();
}

// =============================================================================================
// Philox4x32-10 throughput sampler (oracle/sampler.py:sample_philox).
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__global__ void sample_philox_kernel(const float* __restrict__ probs, int N, int A, uint64_t seed,
                                     const uint64_t* __restrict__ step_base, uint64_t step_off, uint32_t env_offset,
                                     int32_t* __restrict__ actions) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const uint64_t step = (step_base ? *step_base : 0ull) + step_off;
  uint32_t c[4] = {env_offset + (uint32_t)e, (uint32_t)step, (uint32_t)(step >> 32), 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float u = (float)(c[0] >> 8) * (1.0f / 16777216.0f);
  int act = A - 1;
  float cum = 0.f;
  for (int j = 0; j < A - 1; ++j) {
    cum += probs[(long)e * A + j];
    if (u < cum) {
      act = j;
      break;
    }
  }
  actions[e] = act;
}

// =============================================================================================
// numpy-parity sampler: legacy MT19937 multinomial(1, p - epsneg) per env, ONE serial stream
// (paac.py:34-45; restated in oracle/sampler.py:sample_mt_restated).
// (mt_temper / mt_mix: csrc/mt_ahead.h)

// scratch layout (bytes): pj f64[N*(A-1)] | U f64[N*(A-1)] | blocks u32[nblk*624]
// LDSC > 0 (N*(A-1) <= mt_lds_d(LDSC)): the three work arrays live in LDS instead of the global scratch.
// LDS size classes of the sampler body: 0 = work arrays in the global scratch, 1 = small (the 32..64-environment shards),
// 2 = large (256 environments x 4 actions, 128 x 18: one workgroup with most of the CU's LDS)
// (MT_LDS_D = 1024 draws for class 1, MT_LDS_D2 = 2304 for class 2: csrc/mt_ahead.h)
constexpr int MT_TAB_MAX = 16384;         // class 1: entries of the first-hit table
constexpr int MT_TAB_MAX2 = 66560;        // class 2: 256 environments x 3 draws -> 65,536 + 256 entries
__host__ __device__ constexpr int mt_lds_d(int cls) { return cls == 2 ? MT_LDS_D2 : (cls == 1 ? MT_LDS_D : 1); }
__host__ __device__ constexpr int mt_lds_blk(int cls) { return cls == 0 ? 1 : (624 + 2 * mt_lds_d(cls)) / 624 + 2; }
__host__ __device__ constexpr int mt_tab_max(int cls) { return cls == 2 ? MT_TAB_MAX2 : (cls == 1 ? MT_TAB_MAX : 1); }
__host__ __device__ constexpr int mt_skip_max(int cls) { return cls == 2 ? 4352 : (cls == 1 ? 1280 : 1); }
// Phase 4a of sample_mt_body: jh(e, o) = first category hit when environment e starts drawing at stream offset o, for
// every reachable (e, o).  Environment e has e (J-1) + 1 reachable offsets (every earlier environment drew between 1
// and J doubles), so e is paired with N-1-e -- every pair has (N-1)(J-1) + 2 entries -- and each pair is filled by one
// 16-thread group.  JC > 0: J as a compile-time constant, thresholds in registers, the J draws of an entry requested
// before the compares.  JC == 0: any J, thresholds re-read per entry.
template <int JC>
__device__ __forceinline__ void mt_fill_table(const double* pj_buf, const double* u_buf, unsigned char* jh_tab, int N, int D,
                                              int Jdyn = 0) {
  const int J = JC > 0 ? JC : Jdyn;
  constexpr int JR = JC > 0 ? JC : 1;
  const int tid = threadIdx.x;
  const int grp = tid >> 4, k = tid & 15;
  const int npairs = (N + 1) / 2;
  auto threshold = [&](int e, int j, double& thr, bool& inv) {   // hit(U) == ((U > thr) != inv)
    const double pj = pj_buf[e * J + j];
    inv = !(pj <= 0.5);
    thr = inv ? 1.0 - (1.0 - pj) : 1.0 - pj;
  };
  for (int pe = grp; pe < npairs; pe += 16) {
    const int ea = pe, eb = N - 1 - pe;
    const int cnt_a = ea * (J - 1) + 1, cnt_b = (eb != ea) ? eb * (J - 1) + 1 : 0;
    const int base_a = ea + (J - 1) * ea * (ea - 1) / 2, base_b = eb + (J - 1) * eb * (eb - 1) / 2;
    double thr_a[JR], thr_b[JR];
    bool inv_a[JR], inv_b[JR];
    if constexpr (JC > 0) {
#pragma unroll
      for (int j = 0; j < JC; ++j) {
        threshold(ea, j, thr_a[j], inv_a[j]);
        threshold(eb, j, thr_b[j], inv_b[j]);
      }
    }
    if constexpr (JC > 0) {
      // four entries per iteration, every draw of all four requested before the first compare (the loop is LDS-latency
      // bound: one entry at a time spends ~400 cycles per entry, 256 entries per thread at 256 environments)
      constexpr int UNR = 4;
      const int total = cnt_a + cnt_b;
      for (int c0 = k; c0 < total; c0 += 16 * UNR) {
        double U[UNR][JR];
        bool first[UNR], live[UNR];
        int idx[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int c = c0 + 16 * u;
          live[u] = c < total;
          first[u] = c < cnt_a;
          idx[u] = first[u] ? c : c - cnt_a;
          const int o = (first[u] ? ea : eb) + (live[u] ? idx[u] : 0);
#pragma unroll
          for (int j = 0; j < JC; ++j) U[u][j] = u_buf[o + j < D ? o + j : D - 1];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          int jh = J;
#pragma unroll
          for (int j = JC - 1; j >= 0; --j) {
            const double thr = first[u] ? thr_a[j] : thr_b[j];
            const bool inv = first[u] ? inv_a[j] : inv_b[j];
            if ((U[u][j] > thr) != inv) jh = j;
          }
          if (live[u]) jh_tab[(first[u] ? base_a : base_b) + idx[u]] = (unsigned char)jh;
        }
      }
      continue;
    }
    for (int c = k; c < cnt_a + cnt_b; c += 16) {
      const bool first = c < cnt_a;
      const int e = first ? ea : eb;
      const int idx = first ? c : c - cnt_a;   // table slot of the pair member
      const int o = e + idx;                   // stream offset it stands for
      int jh = J;
      {
        for (int j = J - 1; j >= 0; --j) {
          double thr;
          bool inv;
          threshold(e, j, thr, inv);
          if ((u_buf[o + j < D ? o + j : D - 1] > thr) != inv) jh = j;
        }
      }
      jh_tab[(first ? base_a : base_b) + idx] = (unsigned char)jh;
    }
  }
}

// Walk thresholds carry their sense in the sign bit: hit(U) == ((U > |t|) != signbit(t)) -- thresholds lie in [0, 1], so one
// 8-byte LDS read per (environment, category) instead of a threshold and a flag (the walks are bound by the LDS instructions
// they issue: 2 x J per hop instead of 3 x J).
__device__ __forceinline__ double mt_pack_threshold(const double cond) {
  const bool inv = !(cond <= 0.5);
  const double thr = inv ? 1.0 - (1.0 - cond) : 1.0 - cond;
  return inv ? __longlong_as_double(__double_as_longlong(thr) | (long long)0x8000000000000000ull) : thr;
}
__device__ __forceinline__ bool mt_hit(const double U, const double t) {
  return (U > fabs(t)) != (__double_as_longlong(t) < 0);
}

// Small shards (up to 32 environments x up to 7 actions) without a table: the environments are cut into groups of QG (8,
// or 4 where the walks still fit the workgroup); group k can be entered at k QG (J-1) + 1 stream offsets, and one THREAD
// per (group, entry offset) walks its group from there, deciding each environment's first hit itself (J draws against
// the environment's J thresholds, all requested before the first compare) and recording the categories it meets -- QG
// dependent LDS round trips, all walks side by side (groups 0..31/QG: at most 256 threads).  Then one lane follows the
// group exits and every environment picks the record of the walk that really happened.  thr_s / inv_s: the thresholds
// phase 1 left behind.  Returns the consumed-draw count through *used_s.
__host__ __device__ constexpr int mt_walks(const int qg, const int c1) {      // walks of all 32 / qg groups
  return 32 / qg + qg * c1 * ((32 / qg) * (32 / qg - 1) / 2);
}
template <int JC, int QG>
__device__ __forceinline__ void mt_group_walks(const double* thr_s, const double* u_buf, const unsigned char* inv_s, unsigned short* rec_s,
                                               int* entry_s, int* used_s, const int N, const int D,
                                               int32_t* __restrict__ actions, int16_t* act_lds) {
  constexpr int C1 = JC - 1, NGRP = 32 / QG;
  static_assert(mt_walks(QG, C1) <= 256, "one thread per walk");
  const int tid = threadIdx.x;
  MISC_STAMP(5);
  unsigned char* act_h = reinterpret_cast<unsigned char*>(rec_s);                // [walk <= 256][QG hops]
  unsigned short* end_h = rec_s + 1024;                                           // [walk] exit offset
  auto first_walk = [&](const int k) { return k + QG * C1 * (k * (k - 1) / 2); };   // walks of groups 0..k-1
  int g = 0;
#pragma unroll
  for (int k = 1; k < NGRP; ++k) g = (tid >= first_walk(k)) ? k : g;
  const int h = tid - first_walk(g);
  if (g * QG < N && h < g * QG * C1 + 1) {
    int e = g * QG, o = e + h;
    for (int hop = 0; hop < QG; ++hop) {
      if (e < N) {
        double U[JC], T[JC];
#pragma unroll
        for (int j = 0; j < JC; ++j) {
          U[j] = u_buf[o + j < D ? o + j : D - 1];
          T[j] = thr_s[e * JC + j];
        }
        int jh = JC;
#pragma unroll
        for (int j = JC - 1; j >= 0; --j)
          if (mt_hit(U[j], T[j])) jh = j;
        act_h[tid * QG + hop] = (unsigned char)jh;
        o += (jh + 1 < JC) ? jh + 1 : JC;
        ++e;
      }
    }
    end_h[tid] = (unsigned short)o;
  }
  __syncthreads();
  if (tid == 0) {
    int o = 0;
    for (int k = 0; k * QG < N; ++k) {
      const int w = first_walk(k) + o - k * QG;          // the walk of group k that really happens
      entry_s[k] = w;
      o = end_h[w];
    }
    *used_s = o;
  }
  __syncthreads();
  if (tid < N) {
    const int k = tid / QG;
    const int jh = act_h[entry_s[k] * QG + (tid - k * QG)];
    actions[tid] = jh;                                   // jh == J  <=>  no hit  <=>  action A-1 = J
    if (act_lds) act_lds[tid] = (int16_t)jh;
  }
}

// Large shards (256 environments x 4 actions, 128 x 18): the same group walks spread over SEVERAL workgroups of the launch.
// Every sampler workgroup rebuilds thresholds and doubles for itself (cheap next to a serial walk over all environments),
// takes 256 of the (group of 8 environments, entry offset) walks, and leaves their records in global memory; the workgroup
// that takes the last ticket follows the group exits, reads every environment's category out of the record of the walk
// that really happened, and goes on as the one sampler workgroup used to (stream position, bookkeeping).  Tickets are a
// counter that is never reset: every launch adds W to it (2^32 tickets = more than 10^8 launches for any W here; one
// scratch per (N, A)).
struct MultiWalk {
  unsigned char* rec;         // [walks][8] categories met
  unsigned short* exits;      // [walks] stream offset each walk leaves its group at
  unsigned int* counter;
  int W;                      // sampler workgroups of the launch; 0: single-workgroup sampler
};
constexpr int MW_G = 8;
__host__ __device__ inline long mw_first_walk(const int k, const int c1) { return k + (long)MW_G * c1 * ((long)k * (k - 1) / 2); }
__host__ __device__ inline long mw_walks(const int N, const int A) { return mw_first_walk((N + MW_G - 1) / MW_G, A - 2); }

// group of walk `wid`: the largest k with first_walk(k) <= wid  (first_walk(k) = a k^2 + (1 - a) k, a = 4 (J - 1))
__device__ __forceinline__ int mw_group_of(const long wid, const int c1, const int NG) {
  if (c1 == 0) return (int)wid;
  const float a = 4.0f * (float)c1;
  int k = (int)((sqrtf((1.f - a) * (1.f - a) + 4.f * a * (float)wid) - (1.f - a)) / (2.f * a));
  k = k < 0 ? 0 : (k > NG - 1 ? NG - 1 : k);
  while (k > 0 && mw_first_walk(k, c1) > wid) --k;
  while (k < NG - 1 && mw_first_walk(k + 1, c1) <= wid) ++k;
  return k;
}

// one hop of a walk: first category hit by environment e when it starts drawing at offset o (J = JC > 0, or Jdyn)
template <int JC>
__device__ __forceinline__ int mw_first_hit(const double* thr_s, const unsigned char* inv_s, const double* u_buf, const int e,
                                            const int o, const int D, const int Jdyn) {
  if constexpr (JC > 0) {
    double U[JC], T[JC];
#pragma unroll
    for (int j = 0; j < JC; ++j) {
      U[j] = u_buf[o + j < D ? o + j : D - 1];
      T[j] = thr_s[e * JC + j];
    }
    int jh = JC;
#pragma unroll
    for (int j = JC - 1; j >= 0; --j)
      if (mt_hit(U[j], T[j])) jh = j;
    return jh;
  } else {
    const int J = Jdyn;
    int jh = J;
    constexpr int CH = 6;                                // categories per round trip; a wave scans until its last walk has hit
    for (int j0 = 0; j0 < J && jh == J; j0 += CH) {
      double U[CH], T[CH];
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int j = j0 + q < J ? j0 + q : J - 1;
        U[q] = u_buf[o + j < D ? o + j : D - 1];
        T[q] = thr_s[e * J + j];
      }
#pragma unroll
      for (int q = CH - 1; q >= 0; --q)
        if (j0 + q < J && mt_hit(U[q], T[q])) jh = j0 + q;
    }
    return jh;
  }
}

// Returns 1 in the workgroup that took the last ticket (actions written, *used_s = draws consumed), 0 elsewhere.
// dist_zero: every workgroup has seen only ITS environments' probabilities (the heads were finished inside this launch) and
// reports an exactly-zero conditional probability among them with its ticket -- the count rides in the upper half of the
// ticket word, one atomic --; the last-ticket workgroup then returns 2 WITHOUT following the walks when any workgroup
// reported one (the caller takes the serial path over all environments).
template <int JC>
__device__ __forceinline__ int mt_multi_walks(const MultiWalk mw, const double* thr_s, const unsigned char* inv_s,
                                              const double* u_buf, unsigned short* exits_lds, int* entry_s, int* used_s,
                                              const int N, const int J, const int D, int32_t* __restrict__ actions,
                                              int16_t* act_lds, const bool dist_zero = false, const bool local_zero = false) {
  const int tid = threadIdx.x;
  const int c1 = J - 1, NG = (N + MW_G - 1) / MW_G;
  const long nwk = mw_first_walk(NG, c1);
  const long wid = (long)blockIdx.x * 256 + tid;
  if (wid < nwk) {
    const int k = mw_group_of(wid, c1, NG);
    int e = k * MW_G, o = e + (int)(wid - mw_first_walk(k, c1));
    unsigned int lo = 0, hi = 0;                         // the 8 categories met, one byte each
    for (int hop = 0; hop < MW_G; ++hop) {
      if (e < N) {
        const int jh = mw_first_hit<JC>(thr_s, inv_s, u_buf, e, o, D, J);
        if (hop < 4) lo |= (unsigned int)jh << (8 * hop);
        else hi |= (unsigned int)jh << (8 * (hop - 4));
        o += (jh + 1 < J) ? jh + 1 : J;
        ++e;
      }
    }
    reinterpret_cast<uint2*>(mw.rec)[wid] = make_uint2(lo, hi);
    mw.exits[wid] = (unsigned short)o;
  }
  MISC_STAMP(9);
  // hand-off: every storing thread first drains its own stores to the L2 (workgroup-scope release: s_waitcnt vmcnt(0), no
  // cache maintenance) -- the barrier alone does not wait for another wave's stores in flight --, then thread 0's
  // device-scope release (L2 write-back, then the ticket) publishes the whole workgroup's records: one write-back per
  // workgroup, not one per thread; the same on the acquiring side.  The workgroup that takes the last ticket puts the
  // counter back to 0 (every ticket of this launch has been taken; the next launch is stream-ordered behind this one), so
  // the counter never leaves [0, W] and cannot wrap.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __shared__ int last_s;
  if (tid == 0) {
    __threadfence();
    const unsigned int add = 1u + ((dist_zero && local_zero) ? 0x10000u : 0u);
    const unsigned int old = atomicAdd(mw.counter, add);
    last_s = (old & 0xFFFFu) == (unsigned int)(mw.W - 1) ? (((old + add) >> 16) != 0u ? 2 : 1) : 0;
    if (last_s) {
      __threadfence();
      __hip_atomic_store(mw.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!last_s) return 0;
  if (last_s == 2) return 2;
  MISC_STAMP(10);
  {
    const uint4* src = reinterpret_cast<const uint4*>(mw.exits);       // 8 exit offsets per load (the scratch is padded)
    uint4* dst = reinterpret_cast<uint4*>(exits_lds);
    for (long i = tid; i < (nwk + 7) / 8; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  MISC_STAMP(11);
  if (tid == 0) {
    int o = 0;
    for (int k = 0; k < NG; ++k) {
      const long w = mw_first_walk(k, c1) + o - k * MW_G;      // the walk of group k that really happens
      entry_s[k] = (int)w;
      o = exits_lds[w];
    }
    *used_s = o;
  }
  __syncthreads();
  MISC_STAMP(12);
  for (int e = tid; e < N; e += 256) {
    const int k = e / MW_G;
    const int jh = __builtin_nontemporal_load(mw.rec + (long)entry_s[k] * MW_G + (e - k * MW_G));
    actions[e] = jh;                                     // jh == J  <=>  no hit  <=>  action A-1 = J
    if (act_lds) act_lds[e] = (int16_t)jh;
  }
  __syncthreads();
  return 1;
}

// probs_lds (LDSC > 0 only, nullable): the probabilities are already in LDS (written by this workgroup, barrier passed);
// stw_pre (with probs_lds): the caller requested the 625 state words (3 per thread, clamped index) before producing them.
// probs_hook (with probs_lds): produces the probabilities in probs_lds and ends with a barrier.  It is called AFTER the
// state blocks and the doubles (which need nothing but the state words) so that whatever the hook waits on -- the loads
// of the head partials it requested earlier -- travels while those phases compute.
// Phase 1 of sample_mt_body for one (environment, category j): the row's first j + 1 probabilities requested at once (one
// LDS round trip instead of j dependent ones), then the reference's sequential fp64 subtraction order over the first j of
// them (paac.py:42: p - epsneg in float32; numpy's multinomial: remaining -= p_i in float64).
template <int R>
__device__ __forceinline__ void mt_row_inputs(const float* row_p, const int j, double& remaining, double& p) {
  float row[R];
#pragma unroll
  for (int i = 0; i < R; ++i) row[i] = row_p[i <= j ? i : j];
  remaining = 1.0;
#pragma unroll
  for (int i = 0; i < R; ++i)
    if (i < j) remaining -= (double)(row[i] - 5.9604644775390625e-08f);
  float pj_f = row[0];
#pragma unroll
  for (int i = 1; i < R; ++i) pj_f = (i == j) ? row[i] : pj_f;
  p = (double)(pj_f - 5.9604644775390625e-08f);
}

// A probabilities hook: operator()(dst, scratch, lo, hi) leaves the probabilities of environments [lo, hi) at
// dst[e * A + a] (dst = the body's LDS array unless the hook has its own; scratch = LDS free at that point) and ends with a
// barrier; kOwnRows: the hook serves only the rows asked for (the multi-workgroup sampler: every workgroup finishes the heads
// of ITS environments), so the exact-zero scan is distributed (mt_multi_walks: dist_zero).
struct NoProbsHook {
  __device__ __forceinline__ void operator()() const {}
};
// test hook (paac_debug_report_zero): sampler workgroup `g_dbg_zero_wg` of the heads-folding launch reports an exactly-zero
// conditional probability it has not seen, so that the protocol around it (count in the ticket word, the last-ticket
// workgroup finishing all heads and walking serially) runs on ordinary probabilities; -1 = off
__device__ int g_dbg_zero_wg = -1;
// The multi-workgroup sampler's hook (class 2): finishes the heads of environments [lo, hi) from the fc kernel's per-tile
// partials (csrc/fc_heads.h: the same sums in the same order as heads_from_partials), probabilities to dst[e * A + a] in LDS
// and -- with the values -- to global memory; ends with a barrier.  scratch: (hi - lo) * (A + 1) floats of LDS.
struct RowsHeadsHook {
  const float* partial;
  int ntiles, N, A;
  const float *ba, *bc;
  float *probs_out, *values_out;
  __device__ __forceinline__ void operator()(float* dst, float* scratch, const int lo, const int hi) const {
    const int tid = threadIdx.x;
    const int n_total = N * (A + 1), n = (hi - lo) * (A + 1);
    for (int i = tid; i < n; i += 256) {
      const int a = i % (A + 1);
      float acc = (a < A) ? ba[a] : bc[0];
      for (int t0 = 0; t0 < ntiles; t0 += 32) {
        float v[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) v[t] = partial[(size_t)(t0 + t < ntiles ? t0 + t : 0) * n_total + (size_t)lo * (A + 1) + i];
#pragma unroll
        for (int t = 0; t < 32; ++t) acc += (t0 + t < ntiles) ? v[t] : 0.f;
      }
      scratch[i] = acc;
    }
    __syncthreads();
    heads_softmax_store(hi - lo, A, scratch, dst + (size_t)lo * A, nullptr, probs_out + (size_t)lo * A, values_out + lo, nullptr,
                        nullptr, nullptr);
  }
};
template <class H> struct HookOwnRows { static constexpr bool value = false; };
template <> struct HookOwnRows<RowsHeadsHook> { static constexpr bool value = true; };
// The small shards' hook (up to 32 environments x 7 actions, the headline shape): heads finish AND phase 1 without leaving
// the registers.  Thread (row = tid >> 3, a = tid & 7) owns output a of environment row (a < A: a logit, a == A: the value):
// it sums its partials (requested by the caller before the sampler's own phases), gathers its row's logits from the seven
// neighbouring lanes, takes the softmax exactly as heads_softmax_store does (same operations in the same order: same bits),
// gathers the row's probabilities the same way and forms its conditional probability exactly as phase 1 does -- no LDS
// round trip and no barrier between the partial sums and the thresholds (the separate hook + phase 1 take three).
// Leaves probs in probs_lds / probs_out, values in values_out, pj_buf / thr_s / *any_zero as phase 1 would; NO barrier.
struct FusedHeadsHook {
  const float (&pv)[64];
  float bias;
  int ntiles, N, A;
  float* probs_lds;
  float* probs_out;
  float* values_out;
  __device__ __forceinline__ void operator()(double* pj_buf, double* thr_s, int* any_zero) const {
    const int tid = threadIdx.x, lane = tid & 63;
    const int row = tid >> 3, a = tid & 7, J = A - 1;
    const bool rv = row < N;
    float acc = bias;
#pragma unroll
    for (int t = 0; t < 64; ++t) acc += (t < ntiles) ? pv[t] : 0.f;
    float lg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) lg[j] = __shfl(acc, (lane & ~7) | j, 64);
    float m = lg[0];
#pragma unroll
    for (int j = 1; j < 7; ++j)
      if (j < A) m = fmaxf(m, lg[j]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (j < A) sum += expf(lg[j] - m);
    const float pa = expf(acc - m) / sum;               // (meaningful in the lanes a < A)
    if (rv && a < A) {
      probs_lds[row * A + a] = pa;
      probs_out[row * A + a] = pa;
    }
    if (rv && a == A) values_out[row] = acc;
    float pr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pr[j] = __shfl(pa, (lane & ~7) | j, 64);
    if (rv && a < J) {
      double remaining = 1.0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if (i < a) remaining -= (double)(pr[i] - 5.9604644775390625e-08f);
      const double p = (double)(pa - 5.9604644775390625e-08f);
      const double cond = p / remaining;
      pj_buf[row * J + a] = cond;
      if (cond == 0.0) *any_zero = 1;
      if (thr_s) thr_s[row * J + a] = mt_pack_threshold(cond);
    }
  }
};
template <class H> struct HookFused { static constexpr bool value = false; };
template <> struct HookFused<FusedHeadsHook> { static constexpr bool value = true; };
template <int LDSC, class HOOK = NoProbsHook>
__device__ __forceinline__ bool sample_mt_body(const float* __restrict__ probs, int N, int A,
                                               uint32_t* __restrict__ mt_state, double* __restrict__ pj_g,
                                               double* __restrict__ u_g, uint32_t* __restrict__ blocks_g,
                                               int32_t* __restrict__ actions, int16_t* act_lds,
                                               const float* probs_lds = nullptr, const uint32_t* stw_pre = nullptr,
                                               const HOOK probs_hook = HOOK(), const MultiWalk mw = MultiWalk{nullptr, nullptr, nullptr, 0},
                                               const MtAhead* __restrict__ ahead = nullptr) {
  constexpr bool HOOKED = !__is_same(HOOK, NoProbsHook);
  constexpr bool OWNROWS = HookOwnRows<HOOK>::value;      // the hook finishes the heads of this workgroup's environments only
  constexpr bool FUSED = HookFused<HOOK>::value;          // the hook does phase 1 too
  MISC_STAMP(0);
  constexpr bool LDSPATH = LDSC > 0;
  constexpr int PRW = LDSC == 2 ? 18 : 8;             // probability floats per thread of the one-round-trip load
  __shared__ double pj_s[mt_lds_d(LDSC)];
  __shared__ double u_s[mt_lds_d(LDSC)];
  __shared__ uint32_t blocks_s[mt_lds_blk(LDSC) * 624];
  double* pj_buf = LDSPATH ? pj_s : pj_g;
  double* u_buf = LDSPATH ? u_s : u_g;
  uint32_t* blocks = LDSPATH ? blocks_s : blocks_g;
  const int tid = threadIdx.x;
  const int J = A - 1;
  const int D = N * J;
  __shared__ float probs_s[LDSPATH ? 2 * mt_lds_d(LDSC) : 1];   // N*A = D*A/(A-1) <= 2*D floats
  __shared__ uint32_t pos_s;
  __shared__ int any_zero;       // some conditional probability is exactly 0 (the table chase needs none); later
                                 // reused for the consumed-draw count
  uint32_t pos;
  // csrc/mt_ahead.h (class 2 only): the step's doubles, made one launch ahead, travel with the state words and the
  // probabilities; they are used only if their key matches the stream state found here (pre_ok, uniform)
  constexpr int AHW = LDSC == 2 ? (MT_LDS_D2 + 255) / 256 : (LDSC == 1 ? MT_LDS_D / 256 : 1);
  double ahu[AHW];
  uint32_t ahk[5] = {0u, 0u, 0u, 0u, 0u};
  bool pre_ok = false;
  if (tid == 0) any_zero = 0;    // ordered before phase 1 by the barrier below (LDSPATH) / after phase 2's copy
  if constexpr (LDSPATH) {
    // ONE memory round trip: the 625 state words and the N*A probabilities are all requested before anything
    // is consumed (unrolled, clamped indices), then parked in LDS.
    uint32_t stw[3];
    float prw[PRW];
    if (ahead) {
#pragma unroll
      for (int k = 0; k < 5; ++k) ahk[k] = ahead->hdr[k];
#pragma unroll
      for (int k = 0; k < AHW; ++k) ahu[k] = ahead->u[min(tid + k * 256, MT_LDS_D2 - 1)];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) stw[k] = stw_pre ? stw_pre[k] : mt_state[min(tid + k * 256, 624)];
    if (!probs_lds && !OWNROWS) {
#pragma unroll
      for (int k = 0; k < PRW; ++k) prw[k] = probs[min(tid + k * 256, N * A - 1)];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int idx = tid + k * 256;
      if (idx < 624) blocks[idx] = stw[k];
      if (idx == 624) pos_s = stw[k];
    }
    if (!probs_lds && !OWNROWS) {
#pragma unroll
      for (int k = 0; k < PRW; ++k)
        if (tid + k * 256 < N * A) probs_s[tid + k * 256] = prw[k];
    }
    __syncthreads();
    pos = pos_s;
    if (ahead) {
      pre_ok = ahk[0] == pos && ahk[1] == blocks[0] && ahk[2] == blocks[1] && ahk[3] == blocks[623] && (int)ahk[4] >= N * J;
      if (pre_ok) {
#pragma unroll
        for (int k = 0; k < AHW; ++k)
          if (tid + k * 256 < N * J) u_buf[tid + k * 256] = ahu[k];
      }
    }
  } else {
    pos = mt_state[624];
  }
  MISC_STAMP(1);
  const float* pr = LDSPATH ? (probs_lds ? probs_lds : probs_s) : probs;
  const int nblk = (int)((pos + 2u * (uint32_t)D) / 624u) + 1;
  // the walks of the small shards (mt_group_walks) want thresholds: phase 1 leaves them behind too
  __shared__ __attribute__((aligned(16))) unsigned char jh_tab[mt_tab_max(LDSC)];
  const bool multi = LDSC == 2 && mw.W > 0;          // group walks spread over mw.W workgroups of the launch
  const bool walk_ok = (LDSC == 1 && N <= 32 && 4 + 48 * (J - 1) <= 256) || multi;
  double* thr_s = reinterpret_cast<double*>(jh_tab);           // [D <= 1024]   (the first-hit table is not built then)
  unsigned char* inv_s = jh_tab + (LDSC == 0 ? 0 : 8 * mt_lds_d(LDSC));      // [D]
  static_assert(LDSC == 0 || mt_tab_max(LDSC) >= 9 * mt_lds_d(LDSC), "threshold arrays reuse the table's LDS");
  // phase 1: conditional probabilities p_j / remaining_j, one (env, category) per thread: the running
  // `remaining` is rebuilt with the reference's sequential fp64 subtraction order (cheap), so that only ONE fp64
  // division sits on each thread's critical path instead of J in a row
  auto phase1 = [&](const int e_lo, const int e_hi) {      // environments [e_lo, e_hi)
    for (int d = e_lo * J + tid; d < e_hi * J; d += 256) {
      const int e = d / J, j = d - e * J;
      double remaining = 1.0;
      double p;
      if (LDSPATH && J <= 4) mt_row_inputs<4>(pr + (long)e * A, j, remaining, p);
      else if (LDSPATH && J <= 17) mt_row_inputs<17>(pr + (long)e * A, j, remaining, p);
      else {
        for (int i = 0; i < j; ++i)
          remaining -= (double)(pr[(long)e * A + i] - 5.9604644775390625e-08f);
        p = (double)(pr[(long)e * A + j] - 5.9604644775390625e-08f);
      }
      const double cond = p / remaining;
      pj_buf[d] = cond;
      if (LDSPATH && cond == 0.0) any_zero = 1;
      if (walk_ok) thr_s[d] = mt_pack_threshold(cond);    // hit(U) == ((U > thr) != inv), as mt_fill_table
    }
  };
  // several sampler workgroups: each needs the thresholds of the environments ITS walks visit only; whether some
  // conditional probability is exactly zero (cond == 0  <=>  float32(p - epsneg) == 0) is read off the raw probabilities
  int p1_lo = 0, p1_hi = N;
  if (multi) {
    // (the last category of a row is never drawn for; it is tested like the others -- a division per element costs more
    // than the serial path taken once in 2^24 rows for nothing)
    if constexpr (!OWNROWS) {
      float pz[PRW];
#pragma unroll
      for (int k = 0; k < PRW; ++k) pz[k] = pr[min(tid + k * 256, N * A - 1)];     // N * A <= 2 D <= PRW * 256
      bool z = false;
#pragma unroll
      for (int k = 0; k < PRW; ++k) z = z || ((pz[k] - 5.9604644775390625e-08f) == 0.0f);
      if (z) any_zero = 1;
    }
    const int NGm = (N + MW_G - 1) / MW_G;
    const long nwk = mw_first_walk(NGm, J - 1);
    const long w0 = (long)blockIdx.x * 256, w1 = w0 + 255 < nwk - 1 ? w0 + 255 : nwk - 1;
    if (w0 < nwk) {
      p1_lo = mw_group_of(w0, J - 1, NGm) * MW_G;
      p1_hi = (mw_group_of(w1, J - 1, NGm) + 1) * MW_G;
      p1_hi = p1_hi < N ? p1_hi : N;
    } else {
      p1_hi = 0;
    }
  }
  MISC_STAMP(13);
  if constexpr (!HOOKED) phase1(p1_lo, p1_hi);
  MISC_STAMP(2);
  // phase 2: successive MT19937 state blocks
  if constexpr (!LDSPATH) {
    for (int i = tid; i < 624; i += 256) blocks[i] = mt_state[i];
  }
  __syncthreads();
  for (int b = 1; b < (pre_ok ? 1 : nblk); ++b) {      // (pre_ok: blocks and doubles came ready-made)
    const uint32_t* o = blocks + (long)(b - 1) * 624;
    uint32_t* nw = blocks + (long)b * 624;
    for (int k = tid; k < 227; k += 256) nw[k] = o[k + 397] ^ mt_mix(o[k], o[k + 1]);
    __syncthreads();
    for (int k = 227 + tid; k < 454; k += 256) nw[k] = nw[k - 227] ^ mt_mix(o[k], o[k + 1]);
    __syncthreads();
    for (int k = 454 + tid; k < 623; k += 256) nw[k] = nw[k - 227] ^ mt_mix(o[k], o[k + 1]);
    if (tid == 255) nw[623] = nw[396] ^ mt_mix(o[623], nw[0]);      // both inputs are older than this pass
    __syncthreads();
  }
  MISC_STAMP(3);
  // phase 3: the 53-bit doubles numpy would draw, in stream order
  for (int d = tid; d < (pre_ok ? 0 : D); d += 256) {
    const uint32_t q = pos + 2u * (uint32_t)d;
    const uint32_t a = mt_temper(blocks[q]) >> 5;
    const uint32_t b = mt_temper(blocks[q + 1]) >> 6;
    u_buf[d] = ((double)a * 67108864.0 + (double)b) * 1.1102230246251565404e-16;   // * 2^-53, exact
  }
  __syncthreads();
  if constexpr (HOOKED) {
    if constexpr (OWNROWS) {
      // the heads of this workgroup's environments (jh_tab is free until phase 1 parks the thresholds there), then the
      // exact-zero scan over them: what it finds travels with the ticket (mt_multi_walks: dist_zero)
      probs_hook(probs_s, reinterpret_cast<float*>(jh_tab), p1_lo, p1_hi);
      for (int i = p1_lo * A + tid; i < p1_hi * A; i += 256)
        if ((pr[i] - 5.9604644775390625e-08f) == 0.0f) any_zero = 1;
    } else if constexpr (FUSED) {
      probs_hook(pj_buf, walk_ok ? thr_s : nullptr, &any_zero);
    } else {
      probs_hook();          // ends with a barrier: the probabilities are in LDS
    }
    if constexpr (!FUSED) phase1(p1_lo, p1_hi);
    __syncthreads();
  }
  MISC_STAMP(4);
  // phase 4a (fast path): when every conditional probability is non-zero the stream offset after env e is
  // o + min(jh+1, J) with jh = first category hit when env e starts drawing at offset o.  jh is tabulated for
  // every reachable (e, o) in parallel (o <= e*J), then one lane chases the table: N dependent LDS byte reads
  // instead of N ballot/popcount/compare rounds.
  __shared__ unsigned short skip_tab[mt_skip_max(LDSC)];
  __shared__ int entry_s[33];
  bool chased = false;
  if constexpr (LDSPATH) {
    // env e can only start at offsets e .. e J (every earlier env drew between 1 and J doubles)
    const long tab_entries = (long)N + (long)(J - 1) * N * (N - 1) / 2;
    bool walked = false;
    if constexpr (LDSC == 2) {
      if (multi) {
        if (!OWNROWS && any_zero) {
          if (blockIdx.x != 0) return false;          // the rare exact-zero case: workgroup 0 alone, the serial way
          phase1(0, N);
          __syncthreads();
        } else {
          unsigned short* exits_lds = reinterpret_cast<unsigned short*>(jh_tab + 9 * mt_lds_d(2));
          static_assert(mt_tab_max(2) >= 9 * MT_LDS_D2 + 2 * 16384, "exit offsets of up to 16 k walks next to the thresholds");
          int owner;
          const bool lz = OWNROWS && (any_zero != 0 || (int)blockIdx.x == g_dbg_zero_wg);
          switch (J) {
            case 1: owner = mt_multi_walks<1>(mw, thr_s, inv_s, u_buf, exits_lds, entry_s, &any_zero, N, J, D, actions, act_lds, OWNROWS, lz); break;
            case 2: owner = mt_multi_walks<2>(mw, thr_s, inv_s, u_buf, exits_lds, entry_s, &any_zero, N, J, D, actions, act_lds, OWNROWS, lz); break;
            case 3: owner = mt_multi_walks<3>(mw, thr_s, inv_s, u_buf, exits_lds, entry_s, &any_zero, N, J, D, actions, act_lds, OWNROWS, lz); break;
            case 5: owner = mt_multi_walks<5>(mw, thr_s, inv_s, u_buf, exits_lds, entry_s, &any_zero, N, J, D, actions, act_lds, OWNROWS, lz); break;
            // the 9- and 18-action sets: every category of a hop requested at once -- ONE LDS round trip per hop instead of
            // up to three dependent chunks (a wave waits for its slowest lane anyway): walks 8.4 -> 3 us at 128 x 18
            case 8: owner = mt_multi_walks<8>(mw, thr_s, inv_s, u_buf, exits_lds, entry_s, &any_zero, N, J, D, actions, act_lds, OWNROWS, lz); break;
            case 17: owner = mt_multi_walks<17>(mw, thr_s, inv_s, u_buf, exits_lds, entry_s, &any_zero, N, J, D, actions, act_lds, OWNROWS, lz); break;
            default: owner = mt_multi_walks<0>(mw, thr_s, inv_s, u_buf, exits_lds, entry_s, &any_zero, N, J, D, actions, act_lds, OWNROWS, lz); break;
          }
          if (!owner) return false;
          if (owner == 2) {
            // some workgroup met an exactly-zero conditional probability among its environments: this one (the last
            // ticket) finishes ALL heads and walks the environments the serial way (phase 4b); the walks are discarded
            if constexpr (OWNROWS) probs_hook(probs_s, reinterpret_cast<float*>(jh_tab), 0, N);
            phase1(0, N);
            __syncthreads();
          } else {
            walked = true;
            chased = true;
          }
        }
      }
    }
    if constexpr (LDSC == 1) {
      if (!any_zero && walk_ok) {
        static_assert(mt_skip_max(1) >= 1024 + 256, "walk scratch");
        switch (J) {
#define PAAC_WALK_CASE(JJ) \
  case JJ: mt_group_walks<JJ, (mt_walks(4, JJ - 1) <= 256 ? 4 : 8)>(thr_s, u_buf, inv_s, skip_tab, entry_s, &any_zero, N, D, actions, act_lds); break;
          PAAC_WALK_CASE(1) PAAC_WALK_CASE(2) PAAC_WALK_CASE(3) PAAC_WALK_CASE(4) PAAC_WALK_CASE(5)
          default: mt_group_walks<6, 8>(thr_s, u_buf, inv_s, skip_tab, entry_s, &any_zero, N, D, actions, act_lds); break;
#undef PAAC_WALK_CASE
        }
        __syncthreads();
        walked = true;
        chased = true;
      }
    }
    if (!walked && tab_entries <= mt_tab_max(LDSC) && N <= 256) {
      if (!any_zero) {     // set in phase 1, visible since the barrier after phase 3
        // Table fill (mt_fill_table): the category count is dispatched to a compile-time constant, so the fill has
        // no branch per category
        switch (J) {
          case 1: mt_fill_table<1>(pj_buf, u_buf, jh_tab, N, D); break;
          case 2: mt_fill_table<2>(pj_buf, u_buf, jh_tab, N, D); break;
          case 3: mt_fill_table<3>(pj_buf, u_buf, jh_tab, N, D); break;
          case 4: mt_fill_table<4>(pj_buf, u_buf, jh_tab, N, D); break;
          case 5: mt_fill_table<5>(pj_buf, u_buf, jh_tab, N, D); break;
          case 6: mt_fill_table<6>(pj_buf, u_buf, jh_tab, N, D); break;
          case 7: mt_fill_table<7>(pj_buf, u_buf, jh_tab, N, D); break;
          case 8: mt_fill_table<8>(pj_buf, u_buf, jh_tab, N, D); break;
          default: mt_fill_table<0>(pj_buf, u_buf, jh_tab, N, D, J); break;
        }
        __syncthreads();
        MISC_STAMP(5);
        constexpr int G = 16;                           // environments per group of the two-level chase
        constexpr int MAXQ = (mt_skip_max(LDSC) / 8 + 31) / 32 + 1;   // walks per thread there (a pair of groups per 32 threads)
        const int NG = (N + G - 1) / G;
        const int c1 = J - 1, cg = G * (J - 1);
        // (at 32 environments the single-lane chase is the faster one: 5.3 k cycles against 7.5 k, tools/probe_sampler.py)
        const bool two_level = N > 64 && (NG - 1) * cg + 2 <= 32 * MAXQ && NG + cg * (NG * (NG - 1) / 2) <= mt_skip_max(LDSC);
        if (!two_level) {
          // one lane hops through the table: N dependent LDS byte reads
          if (tid == 0) {
            int o = 0, base = 0, ej = 0;               // base(e) - e = (J-1) e (e-1)/2, ej = e (J-1)
            for (int e = 0; e < N; ++e) {
              const int jh = jh_tab[base + o];         // slot base(e) + (o - e)
              actions[e] = jh;                            // jh == J  <=>  no hit  <=>  action A-1 = J
              if (act_lds) act_lds[e] = (int16_t)jh;
              o += (jh + 1 < J) ? jh + 1 : J;
              base += ej;
              ej += J - 1;
            }
            any_zero = o;                                 // reuse as the consumed-draw count
          }
        } else {
          // Two-level chase.  The stream offset after environment e is a function of the offset before it; environments are
          // cut into groups of G = 16, and for every group and every offset it can be entered at, the offset it is left at
          // is tabulated first: independent walks, each thread advancing its ~16 walks hop by hop without a branch so
          // that their table reads overlap (group k is paired with NG-1-k: every pair has the same number of entries).
          // Then one lane hops over the groups (N / 16 dependent reads instead of N) and one thread per group replays its
          // 16 environments from the true entry offset to emit the actions.
          auto tab_base = [&](const int e) { return e + c1 * (e * (e - 1) / 2); };      // slot of (e, o) = tab_base(e) + o - e
          auto skip_base = [&](const int k) { return k + cg * (k * (k - 1) / 2); };     // group k: entry offsets kG .. kGJ
          const int pr = tid >> 5, ln = tid & 31;         // 8 pairs of groups per pass
          for (int p0 = 0; p0 < (NG + 1) / 2; p0 += 8) {
            const int ka = p0 + pr, kb = NG - 1 - ka;
            const bool pair_ok = ka <= kb;
            const int cnt_a = pair_ok ? ka * cg + 1 : 0, cnt_b = (pair_ok && kb != ka) ? kb * cg + 1 : 0;
            int wo[MAXQ], we[MAXQ], wb[MAXQ], ws[MAXQ];    // offset, environment, table base of it, skip slot (-1: none)
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
              const int c = ln + 32 * q;
              const bool live = c < cnt_a + cnt_b, first = c < cnt_a;
              const int k = first ? ka : kb, idx = first ? c : c - cnt_a;
              we[q] = live ? k * G : 0;
              wo[q] = we[q] + (live ? idx : 0);
              wb[q] = tab_base(we[q]);
              ws[q] = live ? skip_base(k) + idx : -1;
            }
            for (int hop = 0; hop < G; ++hop) {
#pragma unroll
              for (int q = 0; q < MAXQ; ++q) {
                const bool ok = ws[q] >= 0 && we[q] < N;
                const int jh = jh_tab[ok ? wb[q] + wo[q] - we[q] : 0];
                const int st = (jh + 1 < J) ? jh + 1 : J;
                wo[q] += ok ? st : 0;
                wb[q] += 1 + c1 * we[q];                  // tab_base(e + 1) - tab_base(e)
                we[q] += 1;
              }
            }
#pragma unroll
            for (int q = 0; q < MAXQ; ++q)
              if (ws[q] >= 0) skip_tab[ws[q]] = (unsigned short)wo[q];
          }
          __syncthreads();
          if (tid == 0) {
            int o = 0;
            for (int k = 0; k < NG; ++k) {
              entry_s[k] = o;
              o = skip_tab[skip_base(k) + o - k * G];
            }
            any_zero = o;                                 // reuse as the consumed-draw count
          }
          __syncthreads();
          if (tid < NG) {
            int o = entry_s[tid];
            const int e1 = min(N, tid * G + G);
            for (int e = tid * G; e < e1; ++e) {
              const int jh = jh_tab[tab_base(e) + o - e];
              actions[e] = jh;                            // jh == J  <=>  no hit  <=>  action A-1 = J
              if (act_lds) act_lds[e] = (int16_t)jh;
              o += (jh + 1 < J) ? jh + 1 : J;
            }
          }
        }
        __syncthreads();
        chased = true;
      }
    }
  }
  MISC_STAMP(6);
  // phase 4b (only when the fast paths above did not run): one wavefront walks the envs in index order; lane j owns
  // category j
  if (!chased) {
    if (tid < 64) {
      const int lane = tid;
      int o = 0;
      for (int e = 0; e < N; ++e) {
        const bool mine = lane < J;
        const double pj = mine ? pj_buf[(long)e * J + lane] : 0.0;
        const bool nzl = mine && (pj != 0.0);
        const unsigned long long nz = __ballot(nzl);
        const unsigned long long below = (lane == 0) ? 0ull : (nz & ((1ull << lane) - 1ull));
        bool hit = false;
        if (nzl) {
          const double U = u_buf[o + __popcll(below)];
          if (pj <= 0.5) {
            hit = U > 1.0 - pj;
          } else {
            const double q = 1.0 - pj;
            hit = !(U > 1.0 - q);
          }
        }
        const unsigned long long hm = __ballot(hit);
        int act, used;
        if (hm) {
          act = __ffsll((long long)hm) - 1;
          used = __popcll(nz & ((2ull << act) - 1ull));
        } else {
          act = A - 1;
          used = __popcll(nz);
        }
        if (lane == 0) {
          actions[e] = act;
          if (act_lds) act_lds[e] = (int16_t)act;
        }
        o += used;
      }
      if (lane == 0) any_zero = o;                       // the consumed-draw count, like the fast paths leave it
    }
    __syncthreads();
  }
  // Every path has passed a barrier since the actions were written: the caller may read act_lds without another one.
  // The stream position goes back in numpy's convention (pos in [0,624], regenerate lazily) -- by the LAST wave, so that
  // the first ones are already back in the caller's bookkeeping meanwhile.
  if (tid >= 192) {
    const int lane = tid - 192;
    const int o = any_zero;
    const uint32_t abs_pos = pos + 2u * (uint32_t)o;
    int fb = (int)(abs_pos / 624u);
    uint32_t np = abs_pos % 624u;
    if (np == 0 && abs_pos > 0) {
      fb -= 1;
      np = 624;
    }
    if (fb > 0) {
      if (pre_ok) {       // the blocks stayed in the record that came with the doubles
        for (int i = lane; i < 624; i += 64) mt_state[i] = ahead->blocks[(long)fb * 624 + i];
      } else {
        for (int i = lane; i < 624; i += 64) mt_state[i] = blocks[(long)fb * 624 + i];
      }
    }
    if (lane == 0) mt_state[624] = np;
  }
  return true;
}

template <int LDSC>
__global__ __launch_bounds__(256) void sample_mt_kernel(const float* __restrict__ probs, int N, int A,
                                                        uint32_t* __restrict__ mt_state, double* __restrict__ pj_g,
                                                        double* __restrict__ u_g, uint32_t* __restrict__ blocks_g,
                                                        int32_t* __restrict__ actions) {
  sample_mt_body<LDSC>(probs, N, A, mt_state, pj_g, u_g, blocks_g, actions, nullptr);
}

// =============================================================================================
// Frame preprocessing + stacking.
struct Lut84 {
  int v[84];
};
// PIL ImagingScaleAffine (nearest): position accumulated in double, then truncated
// (oracle/preprocess.py:_pil_nearest_lut; verified against PIL there).
constexpr Lut84 make_lut(int src) {
  Lut84 l{};
  const double s = (double)src / 84.0;
  double xo = 0.0 + s * 0.5;
  for (int x = 0; x < 84; ++x) {
    l.v[x] = (int)xo;
    xo += s;
  }
  return l;
}
__constant__ Lut84 kRowLut = make_lut(210);
__constant__ Lut84 kColLut = make_lut(160);

__device__ __forceinline__ uint32_t gray601(uint32_t r, uint32_t g, uint32_t b) {
  return (r * 19595u + g * 38470u + b * 7471u + 32768u) >> 16;
}

constexpr int PRE_ROWS_PER_BAND = 12;  // 84 output rows = PRE_BANDS (7) bands of 12

// grid (N, 7), 256 threads: a workgroup builds 12 output rows of one environment in ONE pass.
//   stage   the 2 x 12 source rows the band's output rows map to (PIL nearest: kRowLut) go to LDS with one 16-byte load per
//           thread (gray: 240 of the 256 threads, 3,840 B; RGB: three loads per thread, 11,520 B) -- whole 160- / 480-byte
//           rows, so every 128-byte line fetched is used in full;
//   gather  after the one barrier, thread q < 252 owns output quad (row q / 21, pixels 4 (q % 21) ..+3): four column
//           gathers per frame from LDS (kColLut), max of the two frames (atari_emulator.py:72: max before resize == after),
//           and the 4-deep history shift on whole pixels: out = (old >> 8) | (new << 24)  (channel 0 = oldest = lowest
//           byte; environment.py:66-71) -- one 16-byte load of the old stack (requested before the staging loads, so it
//           travels with them) and one 16-byte store per thread.
// In-place (stack_out == stack_in) is safe: a thread reads and writes the same 16 bytes.
template <bool RGB>
__global__ __launch_bounds__(256) void preprocess_stack_kernel(const uint8_t* __restrict__ raw, int N,
                                                               const uint32_t* stack_in, uint32_t* stack_out,
                                                               uint32_t* __restrict__ stack_out2,
                                                               const uint8_t* __restrict__ push_mask,
                                                               const uint8_t* __restrict__ reset_mask,
                                                               const float* __restrict__ reset_when_zero) {
  constexpr int ROWB = RGB ? 480 : 160;           // bytes per source row
  constexpr int VPR = ROWB / 16;                  // 16-byte vectors per source row
  constexpr int SRC_VEC = PRE_ROWS_PER_BAND * 2 * VPR;      // 240 (gray) / 720 (RGB)
  constexpr int QPR = OBS_W / 4;                  // 21 output quads per row
  __shared__ uint4 rows[SRC_VEC];                 // [band row][frame][ROWB bytes]
  const int e = blockIdx.x;
  const int band = blockIdx.y;
  const int tid = threadIdx.x;
  const bool push = push_mask ? (push_mask[e] != 0) : true;
  bool reset = reset_mask ? (reset_mask[e] != 0) : false;
  if (reset_when_zero) reset = reset || (reset_when_zero[e] == 0.0f);
  const bool has_quad = tid < PRE_ROWS_PER_BAND * QPR;       // 252
  const int qy = tid / QPR, qx = tid - qy * QPR;
  const long quad = ((long)e * OBS_PIX + (band * PRE_ROWS_PER_BAND + qy) * OBS_W) / 4 + qx;
  uint4 old = make_uint4(0u, 0u, 0u, 0u);
  if (has_quad) old = reinterpret_cast<const uint4*>(stack_in)[quad];
  if (push) {
    const uint8_t* fr = raw + (long)e * 2 * PAAC_RAW_H * ROWB;
#pragma unroll
    for (int v = tid; v < SRC_VEC; v += 256) {
      const int r = v / (2 * VPR), rem = v - r * (2 * VPR);
      const int f = rem / VPR, c = rem - f * VPR;
      const int ry = kRowLut.v[band * PRE_ROWS_PER_BAND + r];
      rows[v] = *reinterpret_cast<const uint4*>(fr + ((long)f * PAAC_RAW_H + ry) * ROWB + c * 16);
    }
  }
  __syncthreads();
  if (!has_quad) return;
  uint4 outv = old;
  if (push) {
    const uint8_t* r0 = reinterpret_cast<const uint8_t*>(rows) + (qy * 2) * ROWB;
    const uint8_t* r1 = r0 + ROWB;
    const uint32_t o[4] = {old.x, old.y, old.z, old.w};
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cx = kColLut.v[qx * 4 + k];
      uint32_t v0, v1;
      if (RGB) {
        v0 = gray601(r0[cx * 3], r0[cx * 3 + 1], r0[cx * 3 + 2]);
        v1 = gray601(r1[cx * 3], r1[cx * 3 + 1], r1[cx * 3 + 2]);
      } else {
        v0 = r0[cx];
        v1 = r1[cx];
      }
      const uint32_t nv = v0 > v1 ? v0 : v1;
      w[k] = ((reset ? 0u : o[k]) >> 8) | (nv << 24);
    }
    outv = make_uint4(w[0], w[1], w[2], w[3]);
  }
  reinterpret_cast<uint4*>(stack_out)[quad] = outv;
  if (stack_out2) reinterpret_cast<uint4*>(stack_out2)[quad] = outv;
}

// =============================================================================================
// Synthetic environments (spec: paac_amd/synthetic.py); device helpers in synth_dev.h.
// Path A: one new 84x84 plane per step.  grid (N, 7), 256 threads; one dword (pixel x 4 channels) per thread-iteration.
__global__ __launch_bounds__(256) void synth_step_a_kernel(uint64_t seed, uint32_t env_offset, int N,
                                                           const int32_t* __restrict__ actions, uint32_t thresh,
                                                           const uint64_t* __restrict__ step_base, uint64_t step_off,
                                                           int force_reset, const uint32_t* stack_in,
                                                           uint32_t* stack_out,
                                                           uint32_t* __restrict__ stack_out2, float* rewards_out,
                                                           float* masks_out, float* ep_reward, int32_t* ep_len,
                                                           FinishedRing* fin) {
  const int e = blockIdx.x;
  const int band = blockIdx.y;
  const uint64_t id = force_reset ? 0ull : (step_base ? *step_base : 0ull) + step_off + 1ull;
  if (!force_reset && band == 0 && threadIdx.x == 0)
    synth_bookkeep(synth_key(seed, env_offset + (uint32_t)e, id), e, actions, thresh, rewards_out, masks_out, ep_reward, ep_len,
                   fin);
  synth_shift_band(seed, env_offset, id, thresh, e * PRE_BANDS + band, stack_in, stack_out, stack_out2, force_reset != 0);
}

// Path A with the numpy-parity sampler folded in: workgroup 0 runs the (inherently serial) MT19937 sampler and then
// the per-env bookkeeping, the other N*7 workgroups shift the observation stacks meanwhile -- the new frame and the
// terminal flag of the synthetic environments do not depend on the action, only reward bookkeeping does.  One launch
// per time step instead of two.  LDSC = 2 (large shards: up to 2304 draws): the sampler workgroup takes most of a CU's LDS,
// which every workgroup of the launch then reserves -- so the stacks are shifted by one workgroup per environment (all 7
// bands) instead of one per band.
template <int LDSC, bool FOLD = false>
__global__ __launch_bounds__(256) void synth_step_a_mt_kernel(const float* __restrict__ probs, int A,
                                                              uint32_t* __restrict__ mt_state,
                                                              int32_t* __restrict__ actions, uint64_t seed,
                                                              uint32_t env_offset, int N, uint32_t thresh,
                                                              const uint64_t* __restrict__ step_base, uint64_t step_off,
                                                              const uint32_t* __restrict__ stack_in,
                                                              uint32_t* __restrict__ stack_out,
                                                              uint32_t* __restrict__ stack_out2, float* rewards_out,
                                                              float* masks_out, float* ep_reward, int32_t* ep_len,
                                                              FinishedRing* fin, const MultiWalk mw,
                                                              uint32_t* __restrict__ raw, const MtAhead* __restrict__ ahead,
                                                              const RowsHeadsHook hs) {
  const uint64_t id = (step_base ? *step_base : 0ull) + step_off + 1ull;
  const int samplers = mw.W > 0 ? mw.W : 1;          // sampler workgroups in front of the shift workgroups
  if ((int)blockIdx.x < samplers) {
    __shared__ int16_t act_s[mt_lds_d(LDSC)];
    // the running episode totals do not depend on the sampler: request them before it (first 256 environments)
    const int e0 = threadIdx.x < N ? threadIdx.x : 0;
    const float ep_reward0 = ep_reward[e0];
    const int32_t ep_len0 = ep_len[e0];
    // (ends past a barrier; with several sampler workgroups only the one that finishes the walks goes on)
    // hs.partial: the heads were NOT finished by a launch of their own -- every sampler workgroup finishes those of the
    // environments its walks visit (paac_act_step_mt, large shards); otherwise `probs` holds all of them
    // (FOLD is a template argument: the two bodies' LDS arrays would otherwise both count against the workgroup)
    bool go_on;
    if constexpr (FOLD)
      go_on = sample_mt_body<LDSC, RowsHeadsHook>(nullptr, N, A, mt_state, nullptr, nullptr, nullptr, actions, act_s, nullptr, nullptr,
                                                  hs, mw, ahead);
    else
      go_on = sample_mt_body<LDSC>(probs, N, A, mt_state, nullptr, nullptr, nullptr, actions, act_s, nullptr, nullptr,
                                   NoProbsHook(), mw, ahead);
    if (!go_on) return;
    MISC_STAMP(7);
    for (int e = threadIdx.x; e < N; e += 256) {
      const uint32_t key = synth_key(seed, env_offset + (uint32_t)e, id);
      if (e < 256)
        synth_bookkeep_with(key, e, act_s[e], thresh, ep_reward0, ep_len0, rewards_out, masks_out, ep_reward, ep_len, fin);
      else
        synth_bookkeep_with(key, e, act_s[e], thresh, ep_reward[e], ep_len[e], rewards_out, masks_out, ep_reward, ep_len, fin);
    }
    MISC_STAMP(8);
    return;
  }
  if constexpr (LDSC == 2) {
    // every workgroup of this launch reserves the sampler's LDS, so a CU holds one: the (environment, band) units are dealt
    // over as many shift workgroups as fit beside the sampler workgroups in ONE round of the CUs
    const int nshift = (int)gridDim.x - samplers;
    for (int u = (int)blockIdx.x - samplers; u < N * PRE_BANDS; u += nshift) {
      if (raw) synth_raw_unit(seed, env_offset, id, u, raw);       // path B: the step's raw screen pair instead of the shift
      else synth_shift_band(seed, env_offset, id, thresh, u, stack_in, stack_out, stack_out2);
    }
  } else {
    if (raw) synth_raw_unit(seed, env_offset, id, (int)blockIdx.x - samplers, raw);
    else synth_shift_band(seed, env_offset, id, thresh, (int)blockIdx.x - samplers, stack_in, stack_out, stack_out2);
  }
}

// The same launch with the heads finish in front of the sampler: workgroup 0 sums the fc kernel's per-tile head partials
// (csrc/fc_heads.h), adds the biases, takes the softmax -- probabilities straight into LDS for the sampler (and to HBM for
// the learner's records) -- then samples and does the bookkeeping.  One launch per step for heads + sampler + env step.
__global__ __launch_bounds__(256) void synth_step_a_mth_kernel(const float* __restrict__ partial, int ntiles,
                                                               const float* __restrict__ ba, const float* __restrict__ bc,
                                                               float* __restrict__ probs_out, float* __restrict__ values_out,
                                                               int A, uint32_t* __restrict__ mt_state,
                                                               int32_t* __restrict__ actions, uint64_t seed,
                                                               uint32_t env_offset, int N, uint32_t thresh,
                                                               const uint64_t* __restrict__ step_base, uint64_t step_off,
                                                               const uint32_t* __restrict__ stack_in,
                                                               uint32_t* __restrict__ stack_out,
                                                               uint32_t* __restrict__ stack_out2, float* rewards_out,
                                                               float* masks_out, float* ep_reward, int32_t* ep_len,
                                                               FinishedRing* fin, uint32_t* __restrict__ raw,
                                                               const MtAhead* __restrict__ ahead) {
  const uint64_t id = (step_base ? *step_base : 0ull) + step_off + 1ull;
  if (blockIdx.x == 0) {
    __shared__ int16_t act_s[kFcHeadsMaxRows];
    __shared__ float lg_s[kFcHeadsMaxRows * 33];
    __shared__ float probs_sh[kFcHeadsMaxRows * 32];
    // nothing below depends on these: request them before the head sums
    uint32_t stw[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) stw[k] = mt_state[min((int)threadIdx.x + k * 256, 624)];
    const int e0 = threadIdx.x < N ? threadIdx.x : 0;
    const bool env_step = rewards_out != nullptr;          // false: policy + sampler only (host environments do the stepping)
    const float ep_reward0 = env_step ? ep_reward[e0] : 0.f;
    const int32_t ep_len0 = env_step ? ep_len[e0] : 0;
    // ... and the action-independent half of environment e0's bookkeeping
    const uint32_t key0 = synth_key(seed, env_offset + (uint32_t)e0, id);
    const uint32_t hr5 = synth_reward_slot(key0);
    const bool term0 = synth_terminal(key0, thresh);
    if (N <= 32 && A <= 7 && ntiles <= 64) {
      // (row = tid >> 3, output = tid & 7): the partials are requested now; summed, turned into probabilities and into
      // conditional probabilities in registers after the sampler's state blocks and doubles (FusedHeadsHook)
      float pv[64];
      const int row = (int)threadIdx.x >> 3, a = (int)threadIdx.x & 7;
      const int n = N * (A + 1);
      const int idx = (row < N && a <= A) ? row * (A + 1) + a : 0;
      const float bias = (a < A) ? ba[a] : bc[0];
#pragma unroll
      for (int t = 0; t < 64; ++t) pv[t] = partial[(size_t)(t < ntiles ? t : 0) * n + idx];
      sample_mt_body<1>(nullptr, N, A, mt_state, nullptr, nullptr, nullptr, actions, act_s, probs_sh, stw,
                        FusedHeadsHook{pv, bias, ntiles, N, A, probs_sh, probs_out, values_out},
                        MultiWalk{nullptr, nullptr, nullptr, 0}, ahead);
    } else {
      heads_from_partials(partial, ntiles, N, A, ba, bc, lg_s, probs_sh, nullptr, probs_out, values_out, nullptr, nullptr,
                          nullptr);
      sample_mt_body<1>(nullptr, N, A, mt_state, nullptr, nullptr, nullptr, actions, act_s, probs_sh, stw, NoProbsHook(),
                        MultiWalk{nullptr, nullptr, nullptr, 0}, ahead);
    }
    // (the sampler body ends past a barrier; its last wave is still writing the stream position back)
    if (env_step && (int)threadIdx.x < N)       // N <= 64 here: one environment per thread
      synth_bookkeep_hashed(hr5, term0, e0, act_s[e0], ep_reward0, ep_len0, rewards_out, masks_out, ep_reward, ep_len, fin);
    return;
  }
  if (raw) synth_raw_unit(seed, env_offset, id, (int)blockIdx.x - 1, raw);     // path B: the preprocess launch follows
  else synth_shift_band(seed, env_offset, id, thresh, (int)blockIdx.x - 1, stack_in, stack_out, stack_out2);
}

// Path B, stage 1: generate the two raw 210x160 gray frames of this step + bookkeeping.
// grid (N, 8), 256 threads; 16800 dwords per env.
__global__ __launch_bounds__(256) void synth_raw_kernel(uint64_t seed, uint32_t env_offset, int N,
                                                        const int32_t* __restrict__ actions, uint32_t thresh,
                                                        const uint64_t* __restrict__ step_base, uint64_t step_off,
                                                        int force_reset, uint32_t* __restrict__ raw, float* rewards_out,
                                                        float* masks_out, float* ep_reward, int32_t* ep_len,
                                                        FinishedRing* fin) {
  const int e = blockIdx.x;
  const uint64_t id = force_reset ? 0ull : (step_base ? *step_base : 0ull) + step_off + 1ull;
  const uint32_t key = synth_key(seed, env_offset + (uint32_t)e, id);
  if (!force_reset && blockIdx.y == 0 && threadIdx.x == 0)
    synth_bookkeep(key, e, actions, thresh, rewards_out, masks_out, ep_reward, ep_len, fin);
  const uint32_t rkey = key ^ 0x5bd1e995u;
  constexpr int WORDS = 2 * PAAC_RAW_H * PAAC_RAW_W / 4;  // 16800
  uint4* out = reinterpret_cast<uint4*>(raw + (long)e * WORDS);
  for (int q = blockIdx.y * 256 + threadIdx.x; q < WORDS / 4; q += 256 * gridDim.y)     // one 16-byte store per 4 words
    out[q] = make_uint4(synth_word(rkey, (uint32_t)(4 * q)), synth_word(rkey, (uint32_t)(4 * q + 1)),
                        synth_word(rkey, (uint32_t)(4 * q + 2)), synth_word(rkey, (uint32_t)(4 * q + 3)));
}

__global__ void fill_f32_kernel(float* p, float v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// =============================================================================================
// Global-norm clip + TF RMSProp over the flat parameter buffer (actor_learner.py:31-34,56-59,70).
constexpr int NORM_BLOCKS = 256;         // tail blocks of norm_kernel
constexpr int kNormPartialsMax = 1024;   // head blocks + tail blocks; unused slots stay 0 (the optimizer step sums all of them)
constexpr int NORM_LANES = 8;            // lanes that share one float4 of the head (the slab sums of net_bwd.hip's finalize)

// The flat gradient is [conv tensors | fc_w fc_b actor critic] (TF variable order).  The norm pass walks it as
//   head blocks: the conv tensors, 32 float4 per block, NORM_LANES adjacent lanes per float4.  After a backward that left
//                its slab reduction pending (paac_loss_backward phase 3) the lanes sum the split-K slabs -- lane r takes
//                slabs r, r + 8, ... (up to 8 loads in flight), then a fixed xor tree -- and lane 0 WRITES the gradient
//                value before using it; otherwise lane 0 reads the value grad_finalize_kernel wrote with the same
//                arithmetic.  Either way the same bits enter the same partial: the norm does not depend on the route.
//   tail blocks: NORM_BLOCKS blocks stride over the rest (fc / heads gradients, written by their kernels).
// partials layout (kNormPartialsMax floats each): [0] sum of squares (what the clip needs), then what the reference's
// gradient summaries need (logger_utils.py:23-33 over the flat gradient, actor_learner.py:85-87): [1] sum, [2] max and
// [3] min over the NONZERO elements, [4] number of exact zeros.  The flat buffer's alignment pads (all in the tail) are
// zeros the reference's flat gradient does not contain; the reader (grad_stats_kernel) knows how many pads there are, so
// zeros count for max / min only when more zeros were seen than there are pads.
struct NormSeg {
  const float* src;   // first slab (pending reduction) or nullptr
  long dst4;          // float4 offset of the tensor inside the flat gradient
  int count4;         // float4s
  int splits;
  long stride;        // floats between slabs
};
struct NormArgs {
  NormSeg seg[6];
  int nseg, head_blocks;
  long tail_begin4;
};

__global__ __launch_bounds__(256) void norm_kernel(float* __restrict__ g, long n4, float scale, const NormArgs a,
                                                   float* __restrict__ partials) {
  float acc = 0.f, sum = 0.f, mx = -INFINITY, mn = INFINITY, zeros = 0.f;
  auto take = [&](const float4 q) {
    const float x = q.x * scale, y = q.y * scale, z = q.z * scale, w = q.w * scale;
    acc += (x * x + y * y) + (z * z + w * w);
    sum += (x + y) + (z + w);
    const float e[4] = {x, y, z, w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool nz = e[c] != 0.f;
      mx = fmaxf(mx, nz ? e[c] : -INFINITY);
      mn = fminf(mn, nz ? e[c] : INFINITY);
      zeros += nz ? 0.f : 1.f;
    }
  };
  float4* g4 = reinterpret_cast<float4*>(g);
  if ((int)blockIdx.x < a.head_blocks) {
    const int r = threadIdx.x % NORM_LANES;
    long h = (long)blockIdx.x * (256 / NORM_LANES) + threadIdx.x / NORM_LANES;   // float4 inside the concatenated conv tensors
    NormSeg sg = a.seg[0];
#pragma unroll
    for (int j = 1; j < 6; ++j) {
      if (j < a.nseg && h >= sg.count4) {
        h -= sg.count4;
        sg = a.seg[j];
      }
    }
    const bool live = h < sg.count4;     // whole lane groups are live or not: the shuffles below stay inside a group
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (sg.src != nullptr) {
      for (int s0 = r; s0 < sg.splits; s0 += 8 * NORM_LANES) {
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int sp = s0 + u * NORM_LANES;
          t[u] = (live && sp < sg.splits) ? *reinterpret_cast<const f32x4*>(sg.src + (long)sp * sg.stride + 4 * h)
                                          : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
      }
#pragma unroll
      for (int off = 1; off < NORM_LANES; off <<= 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += __shfl_xor(v[c], off, 64);
      }
      if (live && r == 0) *reinterpret_cast<f32x4*>(g4 + sg.dst4 + h) = v;
    } else if (live && r == 0) {
      v = *reinterpret_cast<const f32x4*>(g4 + sg.dst4 + h);
    }
    if (live && r == 0) take(make_float4(v[0], v[1], v[2], v[3]));
  } else {
    // latency-bound (a few float4 per thread): every load of a pass is issued before the first one is consumed
    constexpr int U = 8;
    constexpr long STRIDE = (long)NORM_BLOCKS * 256;
    for (long i0 = a.tail_begin4 + ((int)blockIdx.x - a.head_blocks) * 256 + threadIdx.x; i0 < n4; i0 += U * STRIDE) {
      float4 v[U];
      bool live[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long i = i0 + u * STRIDE;
        live[u] = i < n4;
        v[u] = live[u] ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (live[u]) take(v[u]);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    acc += __shfl_down(acc, off, 64);
    sum += __shfl_down(sum, off, 64);
    mx = fmaxf(mx, __shfl_down(mx, off, 64));
    mn = fminf(mn, __shfl_down(mn, off, 64));
    zeros += __shfl_down(zeros, off, 64);
  }
  __shared__ float red[5][4];
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    red[0][w] = acc; red[1][w] = sum; red[2][w] = mx; red[3][w] = mn; red[4][w] = zeros;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partials[kNormPartialsMax + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    partials[2 * kNormPartialsMax + blockIdx.x] = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
    partials[3 * kNormPartialsMax + blockIdx.x] = fminf(fminf(red[3][0], red[3][1]), fminf(red[3][2], red[3][3]));
    partials[4 * kNormPartialsMax + blockIdx.x] = (red[4][0] + red[4][1]) + (red[4][2] + red[4][3]);
  }
}

// One workgroup: the partials of the last norm_kernel (np of them) -> out[8] = {sum, sum of squares, max, min, exact zeros
// among the real (unpadded) elements, 0, 0, 0}.  Launched by the host at the progress-record cadence only.
__global__ __launch_bounds__(256) void grad_stats_kernel(const float* __restrict__ partials, int np, float pads,
                                                         float* __restrict__ out) {
  const int t = threadIdx.x;
  float ss = 0.f, sum = 0.f, mx = -INFINITY, mn = INFINITY, zeros = 0.f;
  for (int i = t; i < np; i += 256) {
    ss += partials[i];
    sum += partials[kNormPartialsMax + i];
    mx = fmaxf(mx, partials[2 * kNormPartialsMax + i]);
    mn = fminf(mn, partials[3 * kNormPartialsMax + i]);
    zeros += partials[4 * kNormPartialsMax + i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ss += __shfl_down(ss, off, 64);
    sum += __shfl_down(sum, off, 64);
    mx = fmaxf(mx, __shfl_down(mx, off, 64));
    mn = fminf(mn, __shfl_down(mn, off, 64));
    zeros += __shfl_down(zeros, off, 64);
  }
  __shared__ float red[5][4];
  if ((t & 63) == 0) {
    const int w = t >> 6;
    red[0][w] = ss; red[1][w] = sum; red[2][w] = mx; red[3][w] = mn; red[4][w] = zeros;
  }
  __syncthreads();
  if (t == 0) {
    float a = 0.f, b = 0.f, c = -INFINITY, d = INFINITY, z = 0.f;
    for (int w = 0; w < 4; ++w) {
      a += red[0][w]; b += red[1][w]; c = fmaxf(c, red[2][w]); d = fminf(d, red[3][w]); z += red[4][w];
    }
    const float real_zeros = z - pads;
    if (real_zeros > 0.f) { c = fmaxf(c, 0.f); d = fminf(d, 0.f); }
    out[0] = b; out[1] = a; out[2] = c; out[3] = d; out[4] = real_zeros; out[5] = 0.f; out[6] = 0.f; out[7] = 0.f;
  }
}

constexpr int RMS_U = 4;   // float4 per thread and array in rmsprop_kernel

// What the optimizer step also maintains (csrc/tower.h, csrc/fc_heads.h): the pre-split bf16 planes of the Nature conv
// weights and the fragment-ordered copy of the fc weights -- written by the workgroup that has just updated the values, so
// no separate pack launch (and no second pass over 6.7 MB) follows an update.
struct PackSpec {
  // conv tensors [K, cout] at float offset conv_begin: tile blocks of 32 rows (one k-step) x cout columns
  int nconv;
  long conv_begin[3];
  int conv_cout[3], conv_tiles[3];   // tiles = K / 32
  long conv_dst[3];                  // offset of the layer inside the pack buffer, in 16-byte vectors
  bf16x8* conv_pack;                 // nullptr: no conv pack (NIPS, PAAC_TOWER=0)
  bf16x8* dgrad3_pack;               // conv3 / conv2 data-gradient planes (dgrad_tower.h), with conv_pack
  bf16x8* dgrad2_pack;
  // fc weights [K, H] at float offset fc_begin: tile blocks of 16 rows x 256 columns
  long fc_begin;
  int fc_K, fc_H;
  float* fc_pack;                    // nullptr: no fc pack
  // everything the tile blocks do not own, as float4 ranges walked by the flat blocks
  int nflat;
  long flat_begin4[5], flat_count4[5];
};

// One optimizer element update (TF ApplyRMSProp, actor_learner.py:31-34).
template <bool MOM>
__device__ __forceinline__ void rms_update4(float4& gv, float4& m, float4& mo, float4& v, const float f, const float lr,
                                            const float omd, const float momentum, const float eps) {
#define PAAC_RMS(c)                                        \
  {                                                        \
    const float gg = gv.c * f;                             \
    m.c = m.c + (gg * gg - m.c) * omd;                     \
    const float step = lr * gg / sqrtf(m.c + eps);         \
    mo.c = MOM ? momentum * mo.c + step : step;            \
    v.c = v.c - mo.c;                                      \
  }
  PAAC_RMS(x) PAAC_RMS(y) PAAC_RMS(z) PAAC_RMS(w)
#undef PAAC_RMS
}

// MOM = false: momentum == 0 (the reference's setting, actor_learner.py:31-34): the momentum slot is written (it
// is checkpointed as OptimizerVariables_1) but not read.
// Block classes by blockIdx: [0, fc_tiles) one 16 x 256 tile of the fc weights; then one 32 x cout tile of a conv weight
// tensor each; then flat blocks over everything else.  A tile block streams its rows coalesced like a flat block, then
// passes the updated values through LDS into the packed order (fragments of fc_heads.h / bf16 planes of tower.h).
template <bool MOM>
__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ var, const float* __restrict__ g,
                                                      float* __restrict__ ms, float* __restrict__ mom, long n4,
                                                      const float* __restrict__ lr_dev, float decay, float momentum,
                                                      float eps, float clip_norm, int clip_mode, float scale,
                                                      const float* __restrict__ partials, float* __restrict__ gnorm_out,
                                                      const PackSpec pk, const int fc_tiles, const int conv_tiles) {
  __shared__ float red[4];
  __shared__ float s_factor;
  __shared__ float tile[32 * 68 > 16 * 260 ? 32 * 68 : 16 * 260];
  const int bid = blockIdx.x, tid = threadIdx.x;
  const int cls = bid < fc_tiles ? 0 : (bid < fc_tiles + conv_tiles ? 1 : 2);
  // ---- which float4s does this thread own ----------------------------------------------------------------------------
  long idx[RMS_U];       // float4 index of each of this thread's elements, -1 = none
  int cl = 0, ct_i = 0, ccout = 64;          // conv tile: layer, tile (k-step), columns
  if (cls == 0) {
    const int cblocks = pk.fc_H / 256;
    const int gI = bid / cblocks, cb = bid - gI * cblocks;
#pragma unroll
    for (int u = 0; u < RMS_U; ++u) {
      const int f = tid + 256 * u;           // float4 inside the tile: row f / 64, column group f % 64
      idx[u] = (pk.fc_begin + (long)(16 * gI + (f >> 6)) * pk.fc_H + 256 * cb + 4 * (f & 63)) >> 2;
    }
  } else if (cls == 1) {
    int t = bid - fc_tiles;
    while (cl < pk.nconv - 1 && t >= pk.conv_tiles[cl]) t -= pk.conv_tiles[cl++];
    ct_i = t;
    ccout = pk.conv_cout[cl];
    const int f4_per_tile = 32 * ccout / 4;  // 512 (cout 64) or 256 (cout 32): rows are contiguous, so is the tile
#pragma unroll
    for (int u = 0; u < RMS_U; ++u) {
      const int f = tid + 256 * u;
      idx[u] = f < f4_per_tile ? ((pk.conv_begin[cl] + (long)ct_i * 32 * ccout) >> 2) + f : -1;
    }
  } else {
    const long i0 = (long)(bid - fc_tiles - conv_tiles) * (256 * RMS_U) + tid;
#pragma unroll
    for (int u = 0; u < RMS_U; ++u) {
      long i = i0 + u * 256;
      long at = -1;
#pragma unroll
      for (int sgm = 0; sgm < 5; ++sgm) {
        if (sgm < pk.nflat && at < 0) {
          if (i < pk.flat_count4[sgm]) at = pk.flat_begin4[sgm] + i;
          else i -= pk.flat_count4[sgm];
        }
      }
      idx[u] = at;
    }
  }
  // the streams do not depend on the norm: request them first (RMS_U float4 per thread and array), reduce the
  // partials while they are in flight
  float4 gv[RMS_U], m[RMS_U], mo[RMS_U], v[RMS_U];
#pragma unroll
  for (int u = 0; u < RMS_U; ++u) {
    const long il = idx[u] >= 0 ? idx[u] : 0;
    gv[u] = reinterpret_cast<const float4*>(g)[il];
    m[u] = reinterpret_cast<float4*>(ms)[il];
    mo[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (MOM) mo[u] = reinterpret_cast<float4*>(mom)[il];
    v[u] = reinterpret_cast<float4*>(var)[il];
  }
  const float lr = *lr_dev;
  // every block reduces the same partials in the same order -> identical norm everywhere (unused slots hold 0)
  float acc = (partials[tid] + partials[tid + 256]) + (partials[tid + 512] + partials[tid + 768]);
  static_assert(kNormPartialsMax == 1024, "four partials per thread");
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    const float gn = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    float f = 1.f;
    if (clip_mode == PAAC_CLIP_GLOBAL) f = clip_norm * fminf(1.0f / gn, 1.0f / clip_norm);
    s_factor = f;
    if (gnorm_out && bid == 0) *gnorm_out = gn;
  }
  __syncthreads();
  const float f = s_factor * scale;
  const float omd = 1.0f - decay;
#pragma unroll
  for (int u = 0; u < RMS_U; ++u) {
    if (idx[u] < 0) continue;
    rms_update4<MOM>(gv[u], m[u], mo[u], v[u], f, lr, omd, momentum, eps);
    reinterpret_cast<float4*>(ms)[idx[u]] = m[u];
    reinterpret_cast<float4*>(mom)[idx[u]] = mo[u];
    reinterpret_cast<float4*>(var)[idx[u]] = v[u];
  }
  if (cls == 0) {
    // fc tile -> fragments [tile nt][group g][lane = 16 kq + li][s] = W[16 g + 4 kq + s][16 nt + li]
    float (*tl)[260] = reinterpret_cast<float (*)[260]>(tile);
#pragma unroll
    for (int u = 0; u < RMS_U; ++u) {
      const int fidx = tid + 256 * u;
      *reinterpret_cast<float4*>(&tl[fidx >> 6][4 * (fidx & 63)]) = v[u];
    }
    __syncthreads();
    const int cblocks = pk.fc_H / 256, G = pk.fc_K / 16;
    const int gI = bid / cblocks, cb = bid - gI * cblocks;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = tid + 256 * j;           // fragment: local tile q / 64, lane q % 64
      const int ntl = q >> 6, lane = q & 63, li = lane & 15, kq = lane >> 4;
      float4 o;
      o.x = tl[4 * kq + 0][16 * ntl + li];
      o.y = tl[4 * kq + 1][16 * ntl + li];
      o.z = tl[4 * kq + 2][16 * ntl + li];
      o.w = tl[4 * kq + 3][16 * ntl + li];
      reinterpret_cast<float4*>(pk.fc_pack)[((long)(16 * cb + ntl) * G + gI) * 64 + lane] = o;
    }
  } else if (cls == 1) {
    // conv tile (k-step ct_i: 32 rows x cout) -> per 16-channel tile three planes of 64 lanes x 8 bf16 (tower.h):
    // lane (ch = l & 15, kq = l >> 4) holds W[32 s + 8 kq + j][16 ct + ch], j = 0..7
    float (*tl)[68] = reinterpret_cast<float (*)[68]>(tile);
    const int f4_per_row = ccout / 4;
#pragma unroll
    for (int u = 0; u < RMS_U; ++u) {
      if (idx[u] < 0) continue;
      const int fidx = tid + 256 * u;
      *reinterpret_cast<float4*>(&tl[fidx / f4_per_row][4 * (fidx % f4_per_row)]) = v[u];
    }
    __syncthreads();
    const int ctiles = ccout / 16;
    for (int q = tid; q < ctiles * 64; q += 256) {
      const int ct = q >> 6, lane = q & 63, ch = lane & 15, kq = lane >> 4;
      float x[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) x[jj] = tl[8 * kq + jj][16 * ct + ch];
      bf16x8 h, md, l;
      split3_bf16(x, h, md, l);
      bf16x8* dst = pk.conv_pack + pk.conv_dst[cl] + ((long)ct_i * ctiles + ct) * 192 + lane;
      dst[0] = h;
      dst[64] = md;
      dst[128] = l;
    }
    // the same tile in the data-gradient order (dgrad_tower.h): row r = input channel, 8 consecutive columns = 8
    // consecutive output channels = one vector of the transposed, tap-flipped weight matrix; 32 rows x 8 groups = one
    // vector per thread
    if (cl >= 1 && ccout == 64) {
      const int r = tid >> 3, q8 = tid & 7;
      float x[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) x[jj] = tl[r][8 * q8 + jj];
      bf16x8 h, md, l;
      split3_bf16(x, h, md, l);
      bf16x8* dst;
      if (cl == 2) {                       // conv3: tile = (tap kh*3+kw, input-channel half)
        const int tap_d = 8 - (ct_i >> 1), c2 = 32 * (ct_i & 1) + r;
        const int sI = 2 * tap_d + (q8 >> 2);
        dst = pk.dgrad3_pack + ((long)sI * 4 + (c2 >> 4)) * 192 + (q8 & 3) * 16 + (c2 & 15);
      } else {                             // conv2: tile = tap kh*4+kw, rows = its 32 input channels
        const int kh = ct_i >> 2, kw = ct_i & 3;
        const int clsI = (kh & 1) * 2 + (kw & 1), tap_d = (1 - (kh >> 1)) * 2 + (1 - (kw >> 1));
        const int sI = 2 * tap_d + (q8 >> 2);
        dst = pk.dgrad2_pack + ((long)(clsI * 8 + sI) * 2 + (r >> 4)) * 192 + (q8 & 3) * 16 + (r & 15);
      }
      dst[0] = h;
      dst[64] = md;
      dst[128] = l;
    }
  }
}

// The norm pass's view of the flat gradient: the conv tensors (head blocks; with their pending slab reduction when a
// phase-3 backward left one for `grad`) and where the tail starts.  Returns the number of blocks (= partials), 0 on a
// mismatch.  grad == nullptr: layout only (paac_grad_stats), nothing is consumed.
static int fill_norm_args(paac_ctx* ctx, const float* grad, NormArgs* na) {
  const paac_layout& L = ctx->layout;
  const int nconv_t = 2 * ctx->spec.nconv;            // conv weights and biases: the first tensors of the layout
  memset(na, 0, sizeof(*na));
  const bool pending = grad != nullptr && ctx->pending_fin_grad != nullptr;
  if (pending && ctx->pending_fin_grad != grad) return 0;
  long total4 = 0;
  for (int i = 0; i < nconv_t; ++i) {
    NormSeg& sg = na->seg[na->nseg++];
    sg.dst4 = L.offset[i] / 4;
    sg.count4 = (int)(L.size[i] / 4);
    if ((L.offset[i] % 4) != 0 || (L.size[i] % 4) != 0) return 0;
    if (pending) {
      for (int j = 0; j < ctx->pending_fin.nseg; ++j) {
        const FinalizeSeg& f = ctx->pending_fin.seg[j];
        if (f.dst == grad + L.offset[i]) {
          sg.src = f.src;
          sg.splits = f.splits;
          sg.stride = f.stride;
        }
      }
      if (sg.src == nullptr) return 0;
    }
    total4 += sg.count4;
  }
  na->head_blocks = (int)((total4 + 256 / NORM_LANES - 1) / (256 / NORM_LANES));
  na->tail_begin4 = L.offset[nconv_t] / 4;
  if (na->tail_begin4 != total4) return 0;            // the conv tensors are contiguous from 0 (no pads among them)
  if (na->head_blocks + NORM_BLOCKS > kNormPartialsMax) return 0;
  if (pending) ctx->pending_fin_grad = nullptr;
  return na->head_blocks + NORM_BLOCKS;
}

// The packed copies the optimizer step maintains and the block classes that go with them (from the ctx's layout).
static void fill_pack_spec(const paac_ctx* ctx, PackSpec* pk, int* fc_tiles, int* conv_tiles, long* flat4) {
  const paac_layout& L = ctx->layout;
  const ArchSpec& sp = ctx->spec;
  memset(pk, 0, sizeof(*pk));
  *fc_tiles = 0;
  *conv_tiles = 0;
  long owned_begin[4], owned_end[4];   // float ranges the tile blocks own, ascending
  int nowned = 0;
  if ((ctx->tower_on || ctx->tower2_on) && ctx->tower_pack) {
    pk->nconv = sp.nconv;
    long dst = 0;
    for (int i = 0; i < sp.nconv; ++i) {
      const long K = (long)sp.conv[i].k * sp.conv[i].k * sp.conv[i].cin;
      pk->conv_begin[i] = L.offset[2 * i];
      pk->conv_cout[i] = sp.conv[i].cout;
      pk->conv_tiles[i] = (int)(K / 32);
      pk->conv_dst[i] = dst;
      dst += (K / 32) * (sp.conv[i].cout / 16) * 192;
      *conv_tiles += pk->conv_tiles[i];
      owned_begin[nowned] = L.offset[2 * i];
      owned_end[nowned++] = L.offset[2 * i] + K * sp.conv[i].cout;
    }
    pk->conv_pack = reinterpret_cast<bf16x8*>(ctx->tower_pack);
    pk->dgrad3_pack = pk->conv_pack + kTowerPackVecs;
    pk->dgrad2_pack = pk->dgrad3_pack + kDgradW3Vecs;
  }
  if (ctx->fc_pack && sp.fc % 256 == 0) {
    pk->fc_begin = L.offset[2 * sp.nconv];
    pk->fc_K = sp.flat;
    pk->fc_H = sp.fc;
    pk->fc_pack = reinterpret_cast<float*>(ctx->fc_pack);
    *fc_tiles = (sp.flat / 16) * (sp.fc / 256);
    owned_begin[nowned] = pk->fc_begin;
    owned_end[nowned++] = pk->fc_begin + (long)sp.flat * sp.fc;
  }
  long at = 0;
  *flat4 = 0;
  for (int i = 0; i <= nowned; ++i) {
    const long end = i < nowned ? owned_begin[i] : L.total;
    if (end > at) {
      pk->flat_begin4[pk->nflat] = at / 4;
      pk->flat_count4[pk->nflat] = (end - at) / 4;
      *flat4 += (end - at) / 4;
      ++pk->nflat;
    }
    if (i < nowned) at = owned_end[i];
  }
}

// Path B, second launch of a step: the observation stacks from the raw screen pairs the step launch has just written
// (max of the two screens, PIL-nearest resize, history push; environments whose mask is 0 -- terminal -- restart from an
// empty history, like paac_synth_step's path B).
static void launch_preprocess_after_step(const uint8_t* raw_scratch, int N, const uint8_t* stack_in, uint8_t* stack_out,
                                         uint8_t* stack_out2, const float* masks, hipStream_t s) {
  ProfScope ps(g_prof_ctx, F_PREPROCESS_STACK, N, s);
  launch_k(preprocess_stack_kernel<false>, dim3(N, PRE_BANDS), dim3(256), s, PROF_WHOLE, raw_scratch, N,
           (const uint32_t*)stack_in, (uint32_t*)stack_out, (uint32_t*)stack_out2, (const uint8_t*)nullptr,
           (const uint8_t*)nullptr, masks);
}

int launch_sample_env_step_heads(const float* partial, int ntiles, const float* ba, const float* bc, float* probs_out,
                                 float* values_out, int A, uint32_t* mt_state, int32_t* actions, uint64_t seed,
                                 uint32_t env_offset, int N, uint32_t thresh, const uint64_t* step_base, uint64_t step_off,
                                 const uint8_t* stack_in, uint8_t* stack_out, uint8_t* stack_out2, float* rewards,
                                 float* masks, float* ep_reward, int32_t* ep_len, void* finished, uint8_t* raw_scratch,
                                 const void* mt_ahead, hipStream_t s) {
  {
    ProfScope ps(g_prof_ctx, F_SAMPLE_ENV_STEP, N, s);
    // (stack_out == nullptr: no environment step -- workgroup 0 alone: heads finish + sampler)
    launch_k(synth_step_a_mth_kernel, dim3(stack_out ? 1 + N * PRE_BANDS : 1), dim3(256), s, PROF_WHOLE, partial, ntiles, ba, bc, probs_out,
             values_out, A, mt_state, actions, seed, env_offset, N, thresh, step_base, step_off, (const uint32_t*)stack_in,
             (uint32_t*)stack_out, (uint32_t*)stack_out2, rewards, masks, ep_reward, ep_len, (FinishedRing*)finished,
             (uint32_t*)raw_scratch, reinterpret_cast<const MtAhead*>(mt_ahead));
  }
  if (raw_scratch) launch_preprocess_after_step(raw_scratch, N, stack_in, stack_out, stack_out2, masks, s);
  return 0;
}

}  // namespace paac

using namespace paac;

// =============================================================================================
// C-ABI wrappers
extern "C" {

#ifdef PAAC_DMM_STAMPS
void paac_debug_set_misc_stamps(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_misc_stamps), &p, sizeof(p)); }
#endif

int paac_nstep_returns(const float* v_boot, const float* rewards, const float* masks, const float* values, int T, int N,
                       double gamma, float* y, float* adv, paac_stream_t stream) {
  PAAC_REQUIRE(T > 0 && N > 0, "paac_nstep_returns: T=%d N=%d", T, N);
  CycleTick ct;
  memset(&ct, 0, sizeof(ct));
  ProfScope ps(g_prof_ctx, F_NSTEP_RETURNS, N * T, (hipStream_t)stream);
  launch_k(nstep_returns_kernel, dim3((N + 63) / 64), dim3(128), (hipStream_t)stream, PROF_WHOLE, v_boot, rewards, masks,
           values, T, N, gamma, y, adv, ct);
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_nstep_returns_tick(const float* v_boot, const float* rewards, const float* masks, const float* values, int T,
                            int N, double gamma, float* y, float* adv, int64_t* global_step_dev, int64_t increment,
                            double initial_lr, int64_t lr_annealing_steps, float* lr_out_dev, uint64_t* tick_dev,
                            uint64_t tick_inc, paac_stream_t stream) {
  PAAC_REQUIRE(T > 0 && N > 0, "paac_nstep_returns_tick: T=%d N=%d", T, N);
  PAAC_REQUIRE(global_step_dev && lr_out_dev && lr_annealing_steps > 0, "paac_nstep_returns_tick: bad arguments");
  CycleTick ct;
  ct.global_step = global_step_dev; ct.step_inc = increment; ct.lr0 = initial_lr; ct.anneal = lr_annealing_steps;
  ct.lr_out = lr_out_dev; ct.tick = tick_dev; ct.tick_inc = tick_inc;
  ProfScope ps(g_prof_ctx, F_NSTEP_RETURNS, N * T, (hipStream_t)stream);
  launch_k(nstep_returns_kernel, dim3((N + 63) / 64), dim3(128), (hipStream_t)stream, PROF_WHOLE, v_boot, rewards, masks,
           values, T, N, gamma, y, adv, ct);
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_lr_step(int64_t* global_step_dev, int64_t increment, double initial_lr, int64_t lr_annealing_steps,
                 float* lr_out_dev, paac_stream_t stream) {
  PAAC_REQUIRE(global_step_dev && lr_out_dev && lr_annealing_steps > 0, "paac_lr_step: bad arguments");
  hipLaunchKernelGGL(lr_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, global_step_dev, increment, initial_lr,
                     lr_annealing_steps, lr_out_dev);
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_counter_add(uint64_t* counter_dev, uint64_t inc, paac_stream_t stream) {
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter_dev, inc);
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_debug_clock(uint64_t* out2_dev, paac_stream_t stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, out2_dev);
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_sample_philox(const float* probs, int N, int A, uint64_t seed, const uint64_t* step_base_dev,
                       uint64_t step_offset, uint32_t env_offset, int32_t* actions, paac_stream_t stream) {
  PAAC_REQUIRE(N > 0 && A >= 2 && A <= 32, "paac_sample_philox: N=%d A=%d", N, A);
  ProfScope ps(g_prof_ctx, F_SAMPLE_PHILOX, N, (hipStream_t)stream);
  launch_k(sample_philox_kernel, dim3((N + 63) / 64), dim3(64), (hipStream_t)stream, PROF_WHOLE, probs, N, A, seed,
           step_base_dev, step_offset, env_offset, actions);
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int64_t paac_sample_mt_scratch_bytes(int N, int A) {
  const int64_t D = (int64_t)N * (A - 1);
  const int64_t nblk = (624 + 2 * D) / 624 + 2;
  return D * 8 * 2 + nblk * 624 * 4 + 64;
}

int paac_sample_mt(const float* probs, int N, int A, uint32_t* mt_state, void* scratch, int32_t* actions,
                   paac_stream_t stream) {
  PAAC_REQUIRE(N > 0 && A >= 2 && A <= 32, "paac_sample_mt: N=%d A=%d", N, A);
  PAAC_REQUIRE(scratch && mt_state, "paac_sample_mt: null scratch/state");
  const int64_t D = (int64_t)N * (A - 1);
  double* pj = (double*)scratch;
  double* u = pj + D;
  uint32_t* blocks = (uint32_t*)(u + D);
  ProfScope ps(g_prof_ctx, F_SAMPLE_MT, N, (hipStream_t)stream);
  if (D <= MT_LDS_D)
    launch_k(sample_mt_kernel<1>, dim3(1), dim3(256), (hipStream_t)stream, PROF_WHOLE, probs, N, A, mt_state, pj, u,
             blocks, actions);
  else if (D <= MT_LDS_D2)
    launch_k(sample_mt_kernel<2>, dim3(1), dim3(256), (hipStream_t)stream, PROF_WHOLE, probs, N, A, mt_state, pj, u,
             blocks, actions);
  else
    launch_k(sample_mt_kernel<0>, dim3(1), dim3(256), (hipStream_t)stream, PROF_WHOLE, probs, N, A, mt_state, pj, u,
             blocks, actions);
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_preprocess_stack(const uint8_t* raw, int is_rgb, int N, const uint8_t* stack_in, uint8_t* stack_out,
                          const uint8_t* push_mask, const uint8_t* reset_mask, paac_stream_t stream) {
  PAAC_REQUIRE(N > 0 && raw && stack_in && stack_out, "paac_preprocess_stack: bad arguments");
  dim3 grid(N, PRE_BANDS);
  ProfScope ps(g_prof_ctx, F_PREPROCESS_STACK, N, (hipStream_t)stream);
  if (is_rgb)
    launch_k(preprocess_stack_kernel<true>, grid, dim3(256), (hipStream_t)stream, PROF_WHOLE, raw, N,
             (const uint32_t*)stack_in, (uint32_t*)stack_out, (uint32_t*)nullptr, push_mask, reset_mask,
             (const float*)nullptr);
  else
    launch_k(preprocess_stack_kernel<false>, grid, dim3(256), (hipStream_t)stream, PROF_WHOLE, raw, N,
             (const uint32_t*)stack_in, (uint32_t*)stack_out, (uint32_t*)nullptr, push_mask, reset_mask,
             (const float*)nullptr);
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_synth_reset(uint64_t seed, uint32_t env_offset, int N, uint8_t* stack_out, uint8_t* raw_scratch,
                     paac_stream_t stream) {
  PAAC_REQUIRE(N > 0 && stack_out, "paac_synth_reset: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (!raw_scratch) {
    hipLaunchKernelGGL(synth_step_a_kernel, dim3(N, PRE_BANDS), dim3(256), 0, s, seed, env_offset, N,
                       (const int32_t*)nullptr, 0u, (const uint64_t*)nullptr, 0ull, 1, (const uint32_t*)stack_out,
                       (uint32_t*)stack_out, (uint32_t*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                       (int32_t*)nullptr, (FinishedRing*)nullptr);
  } else {
    hipLaunchKernelGGL(synth_raw_kernel, dim3(N, 8), dim3(256), 0, s, seed, env_offset, N, (const int32_t*)nullptr, 0u,
                       (const uint64_t*)nullptr, 0ull, 1, (uint32_t*)raw_scratch, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, (int32_t*)nullptr, (FinishedRing*)nullptr);
    // reset_mask = every env: reuse push semantics with an all-zero "mask" float is not available here,
    // so clear the stack first and push with reset handled by the zero fill.
    PAAC_CHECK_HIP(hipMemsetAsync(stack_out, 0, (size_t)N * PAAC_OBS_BYTES, s));
    hipLaunchKernelGGL((preprocess_stack_kernel<false>), dim3(N, PRE_BANDS), dim3(256), 0, s, raw_scratch, N,
                       (const uint32_t*)stack_out, (uint32_t*)stack_out, (uint32_t*)nullptr, (const uint8_t*)nullptr,
                       (const uint8_t*)nullptr, (const float*)nullptr);
  }
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_synth_step(uint64_t seed, uint32_t env_offset, int N, const int32_t* actions, uint32_t terminal_threshold,
                    const uint64_t* step_base_dev, uint64_t step_offset, const uint8_t* stack_in, uint8_t* stack_out,
                    uint8_t* stack_out2, float* rewards_out, float* masks_out, float* ep_reward, int32_t* ep_len,
                    void* finished, uint8_t* raw_scratch, paac_stream_t stream) {
  PAAC_REQUIRE(N > 0 && actions && stack_in && stack_out && rewards_out && masks_out && ep_reward && ep_len,
               "paac_synth_step: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (!raw_scratch) {
    ProfScope ps(g_prof_ctx, F_ENV_STEP, N, s);
    launch_k(synth_step_a_kernel, dim3(N, PRE_BANDS), dim3(256), s, PROF_WHOLE, seed, env_offset, N, actions,
             terminal_threshold, step_base_dev, step_offset, 0, (const uint32_t*)stack_in, (uint32_t*)stack_out,
             (uint32_t*)stack_out2, rewards_out, masks_out, ep_reward, ep_len, (FinishedRing*)finished);
  } else {
    {
      ProfScope ps(g_prof_ctx, F_ENV_STEP, N, s);
      launch_k(synth_raw_kernel, dim3(N, 8), dim3(256), s, PROF_WHOLE, seed, env_offset, N, actions, terminal_threshold,
               step_base_dev, step_offset, 0, (uint32_t*)raw_scratch, rewards_out, masks_out, ep_reward, ep_len,
               (FinishedRing*)finished);
    }
    ProfScope ps(g_prof_ctx, F_PREPROCESS_STACK, N, s);
    launch_k(preprocess_stack_kernel<false>, dim3(N, PRE_BANDS), dim3(256), s, PROF_WHOLE, raw_scratch, N,
             (const uint32_t*)stack_in, (uint32_t*)stack_out, (uint32_t*)stack_out2, (const uint8_t*)nullptr,
             (const uint8_t*)nullptr, (const float*)masks_out);
  }
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_debug_report_zero(int sampler_workgroup) {
  PAAC_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_zero_wg), &sampler_workgroup, sizeof(int)));
  return 0;
}

int64_t paac_walk_scratch_bytes(int N, int A) {
  if (N <= 0 || A < 2) return 0;
  const long nwk = mw_walks(N, A);
  return 64 + ((nwk * 2 + 63) / 64) * 64 + nwk * 8;
}

int paac_sample_mt_synth_step(const float* probs, int A, uint32_t* mt_state, int32_t* actions, uint64_t seed,
                              uint32_t env_offset, int N, uint32_t terminal_threshold, const uint64_t* step_base_dev,
                              uint64_t step_offset, const uint8_t* stack_in, uint8_t* stack_out, uint8_t* stack_out2,
                              float* rewards_out, float* masks_out, float* ep_reward, int32_t* ep_len, void* finished,
                              void* walk_scratch, int64_t walk_scratch_bytes, uint8_t* raw_scratch, paac_stream_t stream) {
  return launch_sample_mt_synth_step(probs, A, mt_state, actions, seed, env_offset, N, terminal_threshold, step_base_dev,
                                     step_offset, stack_in, stack_out, stack_out2, rewards_out, masks_out, ep_reward, ep_len,
                                     finished, walk_scratch, walk_scratch_bytes, raw_scratch, nullptr, (hipStream_t)stream, nullptr);
}

}  // extern "C"

namespace paac {
size_t mt_ahead_bytes() { return sizeof(MtAhead); }

// the step launch of launch_sample_mt_synth_step is the large-LDS sampler with its walks spread over several workgroups:
// only then can those workgroups finish the heads themselves (HeadsPartials)
static bool sampler_is_large(int N, int A) {
  return (int64_t)N * (A - 1) > MT_LDS_D || (long)N + (long)(A - 2) * N * (N - 1) / 2 > MT_TAB_MAX;
}
bool sampler_folds_heads(int N, int A, const void* walk_scratch) {
  return sampler_is_large(N, A) && walk_scratch != nullptr && mw_walks(N, A) <= 16384 && N <= 256 && A <= 32;
}

// mt_ahead (nullable): the record a spare workgroup of the preceding fc launch left (csrc/mt_ahead.h)
int launch_sample_mt_synth_step(const float* probs, int A, uint32_t* mt_state, int32_t* actions, uint64_t seed, uint32_t env_offset,
                                int N, uint32_t terminal_threshold, const uint64_t* step_base_dev, uint64_t step_offset,
                                const uint8_t* stack_in, uint8_t* stack_out, uint8_t* stack_out2, float* rewards_out,
                                float* masks_out, float* ep_reward, int32_t* ep_len, void* finished, void* walk_scratch,
                                int64_t walk_scratch_bytes, uint8_t* raw_scratch, const void* mt_ahead, hipStream_t stream,
                                const HeadsPartials* heads) {
  PAAC_REQUIRE(N > 0 && A >= 2 && A <= 32, "paac_sample_mt_synth_step: N=%d A=%d", N, A);
  PAAC_REQUIRE((int64_t)N * (A - 1) <= MT_LDS_D2, "paac_sample_mt_synth_step: N*(A-1)=%ld exceeds the fused kernel's limit %d "
               "(use paac_sample_mt + paac_synth_step)", (long)N * (A - 1), MT_LDS_D2);
  PAAC_REQUIRE(probs && mt_state && actions && stack_in && stack_out && rewards_out && masks_out && ep_reward && ep_len,
               "paac_sample_mt_synth_step: null argument");
  RowsHeadsHook hs{nullptr, 0, N, A, nullptr, nullptr, nullptr, nullptr};
  {
  ProfScope ps(g_prof_ctx, F_SAMPLE_ENV_STEP, N, (hipStream_t)stream);
  // small shards: the small-LDS sampler, one shift workgroup per band; beyond 1024 draws or 64 environments (where the
  // two-level chase needs the large first-hit table): the large-LDS sampler, one shift workgroup per environment
  const bool large = sampler_is_large(N, A);
  PAAC_REQUIRE(!(heads && heads->partial) || sampler_folds_heads(N, A, walk_scratch),
               "paac_sample_mt_synth_step: unfinished heads need the multi-workgroup sampler (N=%d A=%d)", N, A);
  MultiWalk mw{nullptr, nullptr, nullptr, 0};
  if (large && walk_scratch != nullptr) {
    // the caller lent a (zero-initialised, otherwise untouched) scratch: the walks go out over several workgroups
    const long nwk = mw_walks(N, A);
    PAAC_REQUIRE(walk_scratch_bytes >= paac_walk_scratch_bytes(N, A), "paac_sample_mt_synth_step: walk scratch of %ld bytes, "
                 "%ld needed (paac_walk_scratch_bytes)", (long)walk_scratch_bytes, (long)paac_walk_scratch_bytes(N, A));
    if (nwk <= 16384 && N <= 256 && A <= 32) {
      char* base = static_cast<char*>(walk_scratch);
      mw.counter = reinterpret_cast<unsigned int*>(base);
      mw.exits = reinterpret_cast<unsigned short*>(base + 64);
      mw.rec = reinterpret_cast<unsigned char*>(base + 64 + ((nwk * 2 + 63) / 64) * 64);
      mw.W = (int)((nwk + 255) / 256);
    }
  }
  if (heads && heads->partial && large)             // the sampler workgroups finish the heads themselves
    hs = RowsHeadsHook{heads->partial, heads->ntiles, N, A, heads->ba, heads->bc, const_cast<float*>(probs), heads->values_out};
  const int samplers = mw.W > 0 ? mw.W : 1;
  int nshift = N * PRE_BANDS;                       // large path: one round of the 256 CUs (one workgroup per CU there)
  if (nshift > 256 - samplers) nshift = 256 - samplers > 32 ? 256 - samplers : 32;
  if (large && hs.partial)
    launch_k((synth_step_a_mt_kernel<2, true>), dim3(samplers + nshift), dim3(256), (hipStream_t)stream, PROF_WHOLE, probs, A,
             mt_state, actions, seed, env_offset, N, terminal_threshold, step_base_dev, step_offset, (const uint32_t*)stack_in,
             (uint32_t*)stack_out, (uint32_t*)stack_out2, rewards_out, masks_out, ep_reward, ep_len, (FinishedRing*)finished, mw,
             (uint32_t*)raw_scratch, reinterpret_cast<const MtAhead*>(mt_ahead), hs);
  else if (large)
    launch_k(synth_step_a_mt_kernel<2>, dim3(samplers + nshift), dim3(256), (hipStream_t)stream, PROF_WHOLE, probs, A,
             mt_state, actions, seed, env_offset, N, terminal_threshold, step_base_dev, step_offset, (const uint32_t*)stack_in,
             (uint32_t*)stack_out, (uint32_t*)stack_out2, rewards_out, masks_out, ep_reward, ep_len, (FinishedRing*)finished, mw,
             (uint32_t*)raw_scratch, reinterpret_cast<const MtAhead*>(mt_ahead), hs);
  else
    launch_k(synth_step_a_mt_kernel<1>, dim3(1 + N * PRE_BANDS), dim3(256), (hipStream_t)stream, PROF_WHOLE, probs, A, mt_state,
             actions, seed, env_offset, N, terminal_threshold, step_base_dev, step_offset, (const uint32_t*)stack_in,
             (uint32_t*)stack_out, (uint32_t*)stack_out2, rewards_out, masks_out, ep_reward, ep_len, (FinishedRing*)finished, mw,
             (uint32_t*)raw_scratch, (const MtAhead*)nullptr, hs);
  }
  if (raw_scratch) launch_preprocess_after_step(raw_scratch, N, stack_in, stack_out, stack_out2, masks_out, (hipStream_t)stream);
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}
}  // namespace paac

extern "C" {

int paac_clip_rmsprop(paac_ctx* ctx, float* params, const float* grad, float* ms, float* mom, int64_t n,
                      const float* lr_dev, float decay, float momentum, float eps, float clip_norm, int clip_mode,
                      float grad_scale, float* gnorm_out, paac_stream_t stream) {
  PAAC_REQUIRE(ctx && params && grad && ms && mom && lr_dev, "paac_clip_rmsprop: null argument");
  PAAC_REQUIRE(n > 0 && (n % 4) == 0, "paac_clip_rmsprop: n=%ld must be a positive multiple of 4 (padded layout)", (long)n);
  PAAC_REQUIRE(clip_mode == PAAC_CLIP_IGNORE || clip_mode == PAAC_CLIP_GLOBAL,
               "paac_clip_rmsprop: clip mode %d (the reference's 'local' branch is undefined)", clip_mode);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(ctx, F_CLIP_RMSPROP, (int)(n / 4), s);
  const long n4 = n / 4;
  NormArgs na;
  const int np = fill_norm_args(ctx, grad, &na);
  PAAC_REQUIRE(np > 0, "paac_clip_rmsprop: a slab reduction is pending for another gradient buffer, or the conv tensors "
                       "need more than %d norm blocks", kNormPartialsMax - NORM_BLOCKS);
  launch_k(norm_kernel, dim3((unsigned)np), dim3(256), s, PROF_FIRST, const_cast<float*>(grad), n4, grad_scale, na,
           ctx->partials);
  PackSpec pk;
  int fc_tiles, conv_tiles;
  long flat4;
  fill_pack_spec(ctx, &pk, &fc_tiles, &conv_tiles, &flat4);
  const dim3 grid((unsigned)(fc_tiles + conv_tiles + (flat4 + 256 * RMS_U - 1) / (256 * RMS_U)));
  if (momentum != 0.f)
    launch_k(rmsprop_kernel<true>, grid, dim3(256), s, PROF_LAST, params, grad, ms, mom, n4, lr_dev, decay, momentum, eps,
             clip_norm, clip_mode, grad_scale, (const float*)ctx->partials, gnorm_out, pk, fc_tiles, conv_tiles);
  else
    launch_k(rmsprop_kernel<false>, grid, dim3(256), s, PROF_LAST, params, grad, ms, mom, n4, lr_dev, decay, momentum, eps,
             clip_norm, clip_mode, grad_scale, (const float*)ctx->partials, gnorm_out, pk, fc_tiles, conv_tiles);
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_grad_stats(paac_ctx* ctx, float* stats_out, paac_stream_t stream) {
  PAAC_REQUIRE(ctx && stats_out, "paac_grad_stats: null argument");
  const float pads = (float)(ctx->layout.total - ctx->layout.total_unpadded);
  NormArgs na;
  const int np = fill_norm_args(ctx, nullptr, &na);
  PAAC_REQUIRE(np > 0, "paac_grad_stats: no norm layout");
  hipLaunchKernelGGL(grad_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ctx->partials, np, pads,
                     stats_out);
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
