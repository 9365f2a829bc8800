// Internal definitions shared by the kernel translation units of libpaac_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/paac_hip.h"

namespace paac {

void set_error(const char* fmt, ...);

#define PAAC_CHECK_HIP(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      paac::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return -2;                                                                          \
    }                                                                                     \
  } while (0)

#define PAAC_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      paac::set_error(__VA_ARGS__);      \
      return -1;                         \
    }                                    \
  } while (0)

constexpr int OBS_H = 84, OBS_W = 84, OBS_C = 4;
constexpr int OBS_PIX = OBS_H * OBS_W;  // 7056

// Architecture description (networks.py:138-169).
struct ConvSpec {
  int ih, iw, cin, oh, ow, cout, k, stride;
};
struct ArchSpec {
  int nconv;
  ConvSpec conv[3];
  int flat;  // fc input features
  int fc;    // fc width H
};
ArchSpec arch_spec(int arch);

// Kernel families for the timing hooks (paac_prof_*).
enum Family {
  F_CONV1_FWD = 0, F_CONV2_FWD, F_CONV3_FWD, F_FC_FWD, F_HEADS_FWD,
  F_HEADS_BWD, F_FC_WGRAD, F_FC_DGRAD, F_CONV3_WGRAD, F_CONV3_DGRAD, F_CONV2_WGRAD, F_CONV2_DGRAD, F_CONV1_WGRAD,
  F_GRAD_FINALIZE, F_CLIP_RMSPROP, F_MISC,
  // the kernels around the network (entry points without a ctx: timed through the ctx profiling was enabled on)
  F_ENV_STEP, F_SAMPLE_ENV_STEP, F_SAMPLE_MT, F_SAMPLE_PHILOX, F_NSTEP_RETURNS, F_PREPROCESS_STACK, F_CONV_TOWER,
  // two contractions in one launch (dmm_pair_kernel) and the conv3 + conv2 data-gradient tower: named for what ran
  F_FC_CONV3_WGRAD, F_CONV2_CONV1_WGRAD, F_DGRAD_TOWER
};
static_assert(F_DGRAD_TOWER + 1 == PAAC_PROF_FAMILIES, "family count");

}  // namespace paac

namespace paac {
// Forward activations of one batch.  Two sets per ctx: [0] acting / bootstrap (paac_forward*), [1] training
// (paac_train_forward + paac_loss_backward) -- so the bootstrap inference of a cycle can run concurrently with
// the training forward on another stream without clobbering the activations the backward pass needs.
struct Workspace {
  float* act[3];   // conv outputs a1..a3 (fp32 NHWC)
  float* fc_slab;  // [FC_SPLITS_MAX][max_batch][H] split-K partials of the fc layer
  float* h;        // [max_batch][H] fc activations
  float* probs;    // [max_batch][A]
  float* values;   // [max_batch]
  float* logits;   // [max_batch][A]
};
}  // namespace paac

namespace paac {
// GEMM ops of the network (tuning table index) and their launch tuning record.
enum Op { OP_CONV1_FWD = 0, OP_CONV2_FWD, OP_CONV3_FWD, OP_FC_FWD, OP_FC_WGRAD, OP_FC_DGRAD, OP_CONV3_WGRAD,
          OP_CONV3_DGRAD, OP_CONV2_WGRAD, OP_CONV2_DGRAD, OP_CONV1_WGRAD, OP_CONV_TOWER, OP_COUNT };
struct Tune {
  int cfg;     // index into the family's configuration table, -1 = size heuristic
  int ksplit;  // blockIdx.z K split (slab epilogues), 0 = heuristic
  int xcd;     // grid dimension tied to the XCD (0 M tiles, 1 N tiles, 2 z), -1 none
};
inline int batch_class(int batch) { return batch > 512 ? 2 : batch > 64 ? 1 : 0; }

constexpr int kDlStride = 36;   // floats per row of paac_ctx::dl_buf (heads.h)

// Split-K slab reduction of the conv weight gradients into the flat gradient (net_bwd.hip): segments of one backward.
struct FinalizeSeg {
  const float* src;  // first slab
  float* dst;
  int count;         // floats
  int splits;
  long stride;       // floats between slabs
};
struct FinalizeArgs {
  FinalizeSeg seg[8];
  int nseg;
};

}  // namespace paac

struct paac_ctx {
  paac_cfg cfg;
  paac::Tune tune[paac::OP_COUNT][3];   // [op][batch class: 0 = batch <= 64, 1 = batch <= 512, 2 = larger]
  paac::ArchSpec spec;
  paac_layout layout;
  int max_batch;
  paac::Workspace ws[2];
  int last_ws;
  float* dact[3];  // gradients wrt conv outputs (post ReLU mask)
  float* dh;       // [max_batch][H]
  float* wslab;    // wgrad split-K slabs (all layers)
  int64_t wslab_floats;
  float* partials; // norm / gradient-summary partials of the last paac_clip_rmsprop (5 x kNormPartialsMax floats)
  // paac_loss_backward(phase = 3) leaves the slab reduction of the conv weight gradients to the next paac_clip_rmsprop on
  // the same gradient buffer (its norm pass does it): the segments, and the buffer they belong to
  paac::FinalizeArgs pending_fin;
  const float* pending_fin_grad;   // nullptr: nothing pending
  // paac_train_forward_trunk stopped the training forward (ws[1]) after the fc layer's split-K slabs: rows covered and
  // slab count; the next backward finishes the heads (fused into its first launch where it can)
  int heads_pending_rows, heads_pending_splits;
  // paac_keep_next_forward: the next acting forward (ws[0] routes of the conv tower + fc_heads_kernel) also leaves its rows'
  // activations -- conv1 / conv2 / conv3 outputs and the fc activations -- at rows [keep_row, keep_row + batch) of the
  // TRAINING activation set, so that an update over rows the acting steps have already computed needs no training forward
  // (weights are frozen inside a cycle).  -1 = off (one shot: the forward that honours it resets it).
  int keep_row;
  int keep_h_only;                 // with keep_row: only the fc activations are kept (bootstrap rows: the update reads nothing else of them)
  int heads_pending_h;             // 1: the pending "slab" holds finished fc activations (bias + ReLU applied), one split
  float* zeros;                    // max(H) zero floats (the bias of heads that read finished activations)
  // csrc/mt_ahead.h: the record a spare workgroup of the acting forward's fc launch leaves for the sampling step behind it
  void* mt_ahead;                  // MtAhead, owned by the ctx
  const uint32_t* ahead_state;     // one shot, set by paac_act_step_mt: the MT19937 state to work ahead from (nullptr: off)
  int ahead_D;                     // doubles the step can consume at most: N * (A - 1)
  float* dl_buf;                   // [max_batch][kDlStride] per-row head gradients + loss terms (heads.h)
  int fc_splits_max;
  // conv tower (csrc/tower.h, Nature only): conv weights pre-split into bf16 planes in MFMA operand order
  void* tower_pack;      // kTowerPackVecs x 16 bytes, nullptr when the tower is off
  void* fc_pack;         // fc weights in fc_heads_kernel's fragment order (flat * H floats)
  int tower_on;          // PAAC_TOWER (default 1)
  int heads_ntiles;      // column groups of the last fc + head-partials launch (fc_heads.h: 16- or 8-wide tiles)
  int no_quarter_tiles;  // PAAC_FC_QUARTER=0: always whole 16 x 16 tiles (A/B measurements)
  int tower2_on;         // the two-conv tower of the stock NIPS geometry (csrc/tower2.h), same knob
  int managed_weights;   // paac_set_managed_weights: 1 = the caller keeps tower_pack current (paac_clip_rmsprop / paac_pack_weights
                         // re-pack) and acting forwards keep no conv1 / conv2 activations; 0 = every forward re-packs first
  // profiling hooks
  int prof_on;
  static constexpr int PROF_MAX_EVENTS = 8192;
  hipEvent_t* ev_start;
  hipEvent_t* ev_stop;
  int* ev_family;
  int* ev_batch;
  int* ev_mix;     // bf16 products per fp32 multiply of the launch's contraction bodies, one byte each (prof_mix)
  int ev_count;
};

namespace paac {

// Timing hooks: while a ProfScope is alive, launch_k() attaches its start/stop events to the kernel dispatch itself
// (hipExtLaunchKernelGGL: the events carry the dispatch's own begin/end timestamps, like rocprofv3's kernel trace),
// so the elapsed time is the kernel's GPU duration without marker-packet overhead.  A scope with several launches
// (first_only / last_only) puts the start on the first and the stop on the last.
struct ProfEvents {
  hipEvent_t start, stop;
  int mix, nmix;
};
extern thread_local ProfEvents g_prof;
extern paac_ctx* g_prof_ctx;   // the ctx paac_prof_enable(ctx, 1) was last called on (nullptr: none)

struct ProfScope {
  paac_ctx* ctx;
  int idx;
  ProfScope(paac_ctx* c, int family, int batch, hipStream_t) : ctx(c), idx(-1) {
    if (c && c->prof_on && c->ev_count < paac_ctx::PROF_MAX_EVENTS) {
      idx = c->ev_count++;
      c->ev_family[idx] = family;
      c->ev_batch[idx] = batch;
      g_prof.start = c->ev_start[idx];
      g_prof.stop = c->ev_stop[idx];
    }
  }
  ~ProfScope() {
    if (idx >= 0) ctx->ev_mix[idx] = g_prof.mix;
    g_prof.start = nullptr;
    g_prof.stop = nullptr;
    g_prof.mix = 0;
    g_prof.nmix = 0;
  }
};

// Instruction mix of a contraction body launched inside the current ProfScope: MFMA products issued per fp32 multiply --
// 1 = fp32 MFMA, 3 = exact-bf16 path (u8 operand x fp32 split into 3 bf16 terms), 6 = split-bf16 path.  bench.py prices
// the family against the ceiling of what it actually ran.  Up to four bodies per launch, in launch order.
inline void prof_mix(int products) {
  if (g_prof.start && g_prof.nmix < 4) g_prof.mix |= (products & 255) << (8 * g_prof.nmix++);
}

enum { PROF_WHOLE = 0, PROF_FIRST = 1, PROF_LAST = 2, PROF_NONE = 3 };
template <class K, class... Args>
inline void launch_k(K kernel, dim3 grid, dim3 block, hipStream_t s, int part, Args... args) {
  hipEvent_t st = (part == PROF_WHOLE || part == PROF_FIRST) ? g_prof.start : nullptr;
  hipEvent_t sp = (part == PROF_WHOLE || part == PROF_LAST) ? g_prof.stop : nullptr;
  if (st || sp)
    hipExtLaunchKernelGGL(kernel, grid, block, 0, s, st, sp, 0, args...);
  else
    hipLaunchKernelGGL(kernel, grid, block, 0, s, args...);
}

// launchers implemented in the kernel translation units
int launch_forward_trunk_train(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, hipStream_t s);
int launch_deferred_heads(paac_ctx* ctx, const float* params, hipStream_t s);
int launch_forward(paac_ctx* ctx, int ws, const float* params, const uint8_t* states, int batch, float* logits,
                   float* probs, float* values, hipStream_t s);
struct SynthStepArgs;
int launch_forward_sample(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, float* probs,
                          float* values, uint64_t seed, const uint64_t* step_base, uint64_t step_off,
                          uint32_t env_offset, int32_t* actions, const SynthStepArgs* step, hipStream_t s);
int launch_forward_sample_step(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, float* probs,
                               float* values, uint64_t seed, const uint64_t* step_base, uint64_t step_off,
                               uint32_t env_offset, int32_t* actions, uint64_t env_seed, uint32_t thresh,
                               uint8_t* stack_out, float* rewards, float* masks, float* ep_reward, int32_t* ep_len,
                               void* finished, hipStream_t s);
int launch_pack_weights(paac_ctx* ctx, const float* params, hipStream_t s);
int launch_pack_dgrad(paac_ctx* ctx, const float* params, hipStream_t s);
// the fc kernel's per-tile head partials of an acting forward whose heads a later launch finishes (csrc/fc_heads.h)
struct HeadsPartials {
  const float* partial;
  int ntiles;
  const float *ba, *bc;
  float* values_out;
};
size_t mt_ahead_bytes();
bool sampler_folds_heads(int N, int A, const void* walk_scratch);
bool tower2_available();
int launch_sample_mt_synth_step(const float* probs, int A, uint32_t* mt_state, int32_t* actions, uint64_t seed, uint32_t env_offset,
                                int N, uint32_t terminal_threshold, const uint64_t* step_base_dev, uint64_t step_offset,
                                const uint8_t* stack_in, uint8_t* stack_out, uint8_t* stack_out2, float* rewards_out,
                                float* masks_out, float* ep_reward, int32_t* ep_len, void* finished, void* walk_scratch,
                                int64_t walk_scratch_bytes, uint8_t* raw_scratch, const void* mt_ahead, hipStream_t stream,
                                const HeadsPartials* heads = nullptr);
int launch_bootstrap_trunk(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, int train_row, hipStream_t s);
int launch_forward_trunk(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, const float** partial,
                         int* ntiles, const float** ba, const float** bc, hipStream_t s);
int launch_sample_env_step_heads(const float* partial, int ntiles, const float* ba, const float* bc, float* probs_out,
                                 float* values_out, int A, uint32_t* mt_state, int32_t* actions, uint64_t seed,
                                 uint32_t env_offset, int N, uint32_t thresh, const uint64_t* step_base, uint64_t step_off,
                                 const uint8_t* stack_in, uint8_t* stack_out, uint8_t* stack_out2, float* rewards,
                                 float* masks, float* ep_reward, int32_t* ep_len, void* finished, uint8_t* raw_scratch,
                                 const void* mt_ahead, hipStream_t s);
int launch_backward(paac_ctx* ctx, const float* params, const uint8_t* states, const int32_t* actions, const float* y,
                    const float* adv, int batch, float beta, float* grad, float* loss_out, int phase, hipStream_t s,
                    const paac_returns* ret = nullptr);

}  // namespace paac
