// C-ABI: context, parameter layout, forward/backward entry points, hipGraph helpers, timing hooks.
#include <stdarg.h>
#include <stdlib.h>

#include <new>

#include "common.h"

namespace paac {

static thread_local char g_err[512] = "";
thread_local ProfEvents g_prof = {nullptr, nullptr, 0, 0};
paac_ctx* g_prof_ctx = nullptr;

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

ArchSpec arch_spec(int arch) {
  ArchSpec s;
  memset(&s, 0, sizeof(s));
  if (arch == PAAC_ARCH_NATURE) {
    s.nconv = 3;
    s.conv[0] = ConvSpec{84, 84, 4, 20, 20, 32, 8, 4};
    s.conv[1] = ConvSpec{20, 20, 32, 9, 9, 64, 4, 2};
    s.conv[2] = ConvSpec{9, 9, 64, 7, 7, 64, 3, 1};
    s.flat = 3136;
    s.fc = 512;
#ifdef PAAC_USER_ARCH
  } else if (arch == PAAC_ARCH_USER) {
    // layer shapes from the build's -DPAAC_USER_* flags, like UserNet (net_common.h): VALID convolutions over 84 x 84 x 4
#ifndef PAAC_USER_K1
#define PAAC_USER_K1 8
#define PAAC_USER_S1 4
#define PAAC_USER_K2 4
#define PAAC_USER_S2 2
#define PAAC_USER_K3 3
#define PAAC_USER_S3 1
#endif
    constexpr int o1 = (84 - PAAC_USER_K1) / PAAC_USER_S1 + 1, o2 = (o1 - PAAC_USER_K2) / PAAC_USER_S2 + 1;
    constexpr int o3 = PAAC_USER_NCONV == 3 ? (o2 - PAAC_USER_K3) / PAAC_USER_S3 + 1 : o2;
    s.nconv = PAAC_USER_NCONV;
    s.conv[0] = ConvSpec{84, 84, 4, o1, o1, PAAC_USER_C1, PAAC_USER_K1, PAAC_USER_S1};
    s.conv[1] = ConvSpec{o1, o1, PAAC_USER_C1, o2, o2, PAAC_USER_C2, PAAC_USER_K2, PAAC_USER_S2};
    if (PAAC_USER_NCONV == 3) s.conv[2] = ConvSpec{o2, o2, PAAC_USER_C2, o3, o3, PAAC_USER_C3, PAAC_USER_K3, PAAC_USER_S3};
    s.flat = PAAC_USER_NCONV == 3 ? o3 * o3 * PAAC_USER_C3 : o2 * o2 * PAAC_USER_C2;
    s.fc = PAAC_USER_H;
#endif
  } else {
    s.nconv = 2;
    s.conv[0] = ConvSpec{84, 84, 4, 20, 20, 16, 8, 4};
    s.conv[1] = ConvSpec{20, 20, 16, 9, 9, 32, 4, 2};
    s.flat = 2592;
    s.fc = 256;
  }
  return s;
}

// which of the library's two compiled geometries besides Nature: the reference's NIPS trunk, or a user architecture
bool arch_supported(int arch) {
#ifdef PAAC_USER_ARCH
  return arch == PAAC_ARCH_NATURE || arch == PAAC_ARCH_USER;
#else
  return arch == PAAC_ARCH_NATURE || arch == PAAC_ARCH_NIPS;
#endif
}

int64_t wslab_floats_needed(int arch);
size_t tower_pack_bytes();
void default_tuning(paac_ctx* c);
int fc_splits_max();

static const char* kFamilyNames[PAAC_PROF_FAMILIES] = {
    "conv1_fwd", "conv2_fwd", "conv3_fwd", "fc_fwd", "heads_fwd", "heads_bwd", "fc_wgrad", "fc_dgrad",
    "conv3_wgrad", "conv3_dgrad", "conv2_wgrad", "conv2_dgrad", "conv1_wgrad", "grad_finalize", "clip_rmsprop", "misc",
    "env_step", "sample_env_step", "sample_mt", "sample_philox", "nstep_returns", "preprocess_stack", "conv_tower",
    "fc_conv3_wgrad", "conv2_conv1_wgrad", "dgrad_tower"};

}  // namespace paac

using namespace paac;

struct paac_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};

extern "C" {

const char* paac_last_error(void) { return g_err; }
int paac_version(void) { return 100; }

int paac_param_layout(int arch, int num_actions, paac_layout* out) {
  PAAC_REQUIRE(out, "paac_param_layout: null out");
  PAAC_REQUIRE(arch_supported(arch), "paac_param_layout: arch %d is not compiled into this library (a library holds Nature "
               "and either the NIPS geometry or one user architecture, paac_user_arch)", arch);
  PAAC_REQUIRE(num_actions >= 2 && num_actions <= 32, "paac_param_layout: num_actions %d not in [2,32]", num_actions);
  memset(out, 0, sizeof(*out));
  const ArchSpec s = arch_spec(arch);
  int t = 0;
  int64_t off = 0, unp = 0;
  auto add = [&](const char* name, int rank, int d0, int d1, int d2, int d3) {
    int64_t size = (int64_t)d0 * (rank > 1 ? d1 : 1) * (rank > 2 ? d2 : 1) * (rank > 3 ? d3 : 1);
    snprintf(out->name[t], sizeof(out->name[t]), "%s", name);
    out->rank[t] = rank;
    out->shape[t][0] = d0; out->shape[t][1] = rank > 1 ? d1 : 0;
    out->shape[t][2] = rank > 2 ? d2 : 0; out->shape[t][3] = rank > 3 ? d3 : 0;
    out->offset[t] = off;
    out->size[t] = size;
    off += (size + 3) / 4 * 4;
    unp += size;
    ++t;
  };
  char nm[32];
  for (int i = 0; i < s.nconv; ++i) {
    snprintf(nm, sizeof(nm), "conv%d_weights", i + 1);
    add(nm, 4, s.conv[i].k, s.conv[i].k, s.conv[i].cin, s.conv[i].cout);
    snprintf(nm, sizeof(nm), "conv%d_biases", i + 1);
    add(nm, 1, s.conv[i].cout, 0, 0, 0);
  }
  snprintf(nm, sizeof(nm), "fc%d_weights", s.nconv + 1);
  add(nm, 2, s.flat, s.fc, 0, 0);
  snprintf(nm, sizeof(nm), "fc%d_biases", s.nconv + 1);
  add(nm, 1, s.fc, 0, 0, 0);
  add("actor_output_weights", 2, s.fc, num_actions, 0, 0);
  add("actor_output_biases", 1, num_actions, 0, 0, 0);
  add("critic_output_weights", 2, s.fc, 1, 0, 0);
  add("critic_output_biases", 1, 1, 0, 0, 0);
  out->num_tensors = t;
  out->total = off;
  out->total_unpadded = unp;
  return 0;
}

int paac_create(const paac_cfg* cfg, paac_ctx** out) {
  PAAC_REQUIRE(cfg && out, "paac_create: null argument");
  PAAC_REQUIRE(cfg->max_batch > 0, "paac_create: max_batch %d", cfg->max_batch);
  int ndev = 0;
  PAAC_CHECK_HIP(hipGetDeviceCount(&ndev));
  PAAC_REQUIRE(ndev > 0, "paac_create: no HIP device visible (this library is MI355X-only; there is no CPU path)");
  PAAC_REQUIRE(cfg->device >= 0 && cfg->device < ndev, "paac_create: device %d of %d", cfg->device, ndev);
  PAAC_CHECK_HIP(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  PAAC_CHECK_HIP(hipGetDeviceProperties(&prop, cfg->device));
  PAAC_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
               "paac_create: device %d is %s; libpaac_hip is built for gfx950 (MI355X) only", cfg->device,
               prop.gcnArchName);
  paac_ctx* c = new (std::nothrow) paac_ctx;
  PAAC_REQUIRE(c, "paac_create: out of host memory");
  memset(c, 0, sizeof(*c));
  c->cfg = *cfg;
  if (paac_param_layout(cfg->arch, cfg->num_actions, &c->layout) != 0) {
    delete c;
    return -1;
  }
  c->spec = arch_spec(cfg->arch);
  c->max_batch = cfg->max_batch;
  for (int o = 0; o < OP_COUNT; ++o)
    for (int k = 0; k < 3; ++k) c->tune[o][k] = Tune{-1, 0, -1};
  default_tuning(c);
  // diagnostic: PAAC_TUNE_OVERRIDE="op:class:cfg:ksplit:xcd[,...]" replaces entries of the table (the numbering of
  // paac_debug_set_tuning), so that a whole bench.py run can be taken with another launch configuration
  if (const char* ov = getenv("PAAC_TUNE_OVERRIDE")) {
    int op, cls, cf, ks, xc, used = 0;
    while (sscanf(ov, "%d:%d:%d:%d:%d%n", &op, &cls, &cf, &ks, &xc, &used) == 5) {
      if (op >= 0 && op < OP_COUNT && cls >= 0 && cls < 3) c->tune[op][cls] = Tune{cf, ks, xc};
      ov += used;
      if (*ov != ',') break;
      ++ov;
    }
  }
  const int64_t B = cfg->max_batch;
  const int A = cfg->num_actions;
  c->fc_splits_max = fc_splits_max();
  for (int w = 0; w < 2; ++w) {
    Workspace& W = c->ws[w];
    for (int i = 0; i < c->spec.nconv; ++i) {
      const ConvSpec& cs = c->spec.conv[i];
      // rows rounded up to 16: the packed conv3 -> fc hand-off (tower.h / fc_heads.h) addresses whole 16-row tiles
      PAAC_CHECK_HIP(hipMalloc(&W.act[i], (size_t)((B + 15) / 16 * 16) * cs.oh * cs.ow * cs.cout * sizeof(float)));
    }
    PAAC_CHECK_HIP(hipMalloc(&W.fc_slab, (size_t)c->fc_splits_max * B * c->spec.fc * sizeof(float)));
    PAAC_CHECK_HIP(hipMalloc(&W.h, (size_t)B * c->spec.fc * sizeof(float)));
    PAAC_CHECK_HIP(hipMalloc(&W.probs, (size_t)B * A * sizeof(float)));
    PAAC_CHECK_HIP(hipMalloc(&W.logits, (size_t)B * A * sizeof(float)));
    PAAC_CHECK_HIP(hipMalloc(&W.values, (size_t)B * sizeof(float)));
  }
  for (int i = 0; i < c->spec.nconv; ++i) {
    const ConvSpec& cs = c->spec.conv[i];
    PAAC_CHECK_HIP(hipMalloc(&c->dact[i], (size_t)B * cs.oh * cs.ow * cs.cout * sizeof(float)));
  }
  PAAC_CHECK_HIP(hipMalloc(&c->dh, (size_t)B * c->spec.fc * sizeof(float)));
  PAAC_CHECK_HIP(hipMalloc(&c->dl_buf, (size_t)B * paac::kDlStride * sizeof(float)));
  c->wslab_floats = wslab_floats_needed(cfg->arch);
  PAAC_CHECK_HIP(hipMalloc(&c->wslab, (size_t)c->wslab_floats * sizeof(float)));
  PAAC_CHECK_HIP(hipMalloc(&c->partials, 8192 * sizeof(float)));
  PAAC_CHECK_HIP(hipMemset(c->partials, 0, 8192 * sizeof(float)));
  PAAC_CHECK_HIP(hipMalloc(&c->zeros, (size_t)(c->spec.fc > 1024 ? c->spec.fc : 1024) * sizeof(float)));
  PAAC_CHECK_HIP(hipMemset(c->zeros, 0, (size_t)(c->spec.fc > 1024 ? c->spec.fc : 1024) * sizeof(float)));
  c->keep_row = -1;
  c->keep_h_only = 0;
  c->heads_pending_h = 0;
  PAAC_CHECK_HIP(hipMalloc(&c->mt_ahead, mt_ahead_bytes()));
  PAAC_CHECK_HIP(hipMemset(c->mt_ahead, 0, mt_ahead_bytes()));
  c->ahead_state = nullptr;
  c->ahead_D = 0;
  {
    const char* v = getenv("PAAC_TOWER");
    c->tower_on = (cfg->arch == PAAC_ARCH_NATURE) && !(v && *v && atoi(v) == 0);
    { const char* q = getenv("PAAC_FC_QUARTER"); c->no_quarter_tiles = (q && *q && atoi(q) == 0) ? 1 : 0; }
    c->tower2_on = (cfg->arch == PAAC_ARCH_NIPS) && tower2_available() && !(v && *v && atoi(v) == 0);
    c->managed_weights = 0;
    c->tower_pack = nullptr;
    if (c->tower_on || c->tower2_on) PAAC_CHECK_HIP(hipMalloc(&c->tower_pack, tower_pack_bytes()));
    PAAC_CHECK_HIP(hipMalloc(&c->fc_pack, (size_t)c->spec.flat * c->spec.fc * sizeof(float)));
  }
  c->ev_start = new hipEvent_t[paac_ctx::PROF_MAX_EVENTS];
  c->ev_stop = new hipEvent_t[paac_ctx::PROF_MAX_EVENTS];
  c->ev_family = new int[paac_ctx::PROF_MAX_EVENTS];
  c->ev_batch = new int[paac_ctx::PROF_MAX_EVENTS];
  c->ev_mix = new int[paac_ctx::PROF_MAX_EVENTS];
  c->ev_count = 0;
  c->prof_on = 0;
  for (int i = 0; i < paac_ctx::PROF_MAX_EVENTS; ++i) {
    c->ev_start[i] = nullptr;
    c->ev_stop[i] = nullptr;
  }
  *out = c;
  return 0;
}

int paac_destroy(paac_ctx* c) {
  if (!c) return 0;
  for (int w = 0; w < 2; ++w) {
    Workspace& W = c->ws[w];
    for (int i = 0; i < 3; ++i)
      if (W.act[i]) (void)hipFree(W.act[i]);
    float* bufs[] = {W.fc_slab, W.h, W.probs, W.logits, W.values};
    for (float* b : bufs)
      if (b) (void)hipFree(b);
  }
  for (int i = 0; i < 3; ++i)
    if (c->dact[i]) (void)hipFree(c->dact[i]);
  float* bufs[] = {c->dh, c->wslab, c->partials, c->dl_buf, c->zeros};
  for (float* b : bufs)
    if (b) (void)hipFree(b);
  if (c->mt_ahead) (void)hipFree(c->mt_ahead);
  if (c->tower_pack) (void)hipFree(c->tower_pack);
  if (c->fc_pack) (void)hipFree(c->fc_pack);
  for (int i = 0; i < paac_ctx::PROF_MAX_EVENTS; ++i) {
    if (c->ev_start[i]) (void)hipEventDestroy(c->ev_start[i]);
    if (c->ev_stop[i]) (void)hipEventDestroy(c->ev_stop[i]);
  }
  delete[] c->ev_start;
  delete[] c->ev_stop;
  delete[] c->ev_family;
  delete[] c->ev_batch;
  delete[] c->ev_mix;
  delete c;
  return 0;
}

int paac_forward(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, float* logits, float* probs,
                 float* values, paac_stream_t stream) {
  PAAC_REQUIRE(ctx && params && states, "paac_forward: null argument");
  PAAC_REQUIRE(batch > 0 && batch <= ctx->max_batch, "paac_forward: batch %d outside (0, max_batch=%d]", batch,
               ctx->max_batch);
  const int rc = launch_forward(ctx, 0, params, states, batch, logits, probs, values, (hipStream_t)stream);
  if (rc) return rc;
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_pack_weights(paac_ctx* ctx, const float* params, paac_stream_t stream) {
  PAAC_REQUIRE(ctx && params, "paac_pack_weights: null argument");
  const int rc = launch_pack_weights(ctx, params, (hipStream_t)stream);
  if (rc) return rc;
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_set_managed_weights(paac_ctx* ctx, int on) {
  PAAC_REQUIRE(ctx, "paac_set_managed_weights: null ctx");
  ctx->managed_weights = on ? 1 : 0;
  return 0;
}

int paac_train_forward(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, float* values,
                       paac_stream_t stream) {
  PAAC_REQUIRE(ctx && params && states, "paac_train_forward: null argument");
  PAAC_REQUIRE(batch > 0 && batch <= ctx->max_batch, "paac_train_forward: batch %d outside (0, max_batch=%d]", batch,
               ctx->max_batch);
  const int rc = launch_forward(ctx, 1, params, states, batch, nullptr, nullptr, values, (hipStream_t)stream);
  if (rc) return rc;
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_train_forward_trunk(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, paac_stream_t stream) {
  PAAC_REQUIRE(ctx && params && states, "paac_train_forward_trunk: null argument");
  PAAC_REQUIRE(batch > 0 && batch <= ctx->max_batch, "paac_train_forward_trunk: batch %d outside (0, max_batch=%d]", batch,
               ctx->max_batch);
  const int rc = launch_forward_trunk_train(ctx, params, states, batch, (hipStream_t)stream);
  if (rc) return rc;
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_keep_next_forward(paac_ctx* ctx, int train_row) {
  PAAC_REQUIRE(ctx, "paac_keep_next_forward: null ctx");
  PAAC_REQUIRE(train_row >= -1 && train_row < ctx->max_batch, "paac_keep_next_forward: row %d outside [-1, max_batch=%d)",
               train_row, ctx->max_batch);
  ctx->keep_row = train_row;
  return 0;
}

int paac_bootstrap_forward_trunk(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, int train_row,
                                 paac_stream_t stream) {
  PAAC_REQUIRE(ctx && params && states, "paac_bootstrap_forward_trunk: null argument");
  PAAC_REQUIRE(batch > 0 && train_row >= 0 && train_row + batch <= ctx->max_batch,
               "paac_bootstrap_forward_trunk: rows [%d, %d) outside max_batch=%d", train_row, train_row + batch, ctx->max_batch);
  const int rc = launch_bootstrap_trunk(ctx, params, states, batch, train_row, (hipStream_t)stream);
  if (rc) return rc;
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_forward_sample(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, float* probs,
                        float* values, uint64_t seed, const uint64_t* step_base_dev, uint64_t step_offset,
                        uint32_t env_offset, int32_t* actions, paac_stream_t stream) {
  PAAC_REQUIRE(ctx && params && states && actions, "paac_forward_sample: null argument");
  PAAC_REQUIRE(batch > 0 && batch <= ctx->max_batch, "paac_forward_sample: batch %d outside (0, max_batch=%d]", batch,
               ctx->max_batch);
  const int rc = launch_forward_sample(ctx, params, states, batch, probs, values, seed, step_base_dev, step_offset,
                                       env_offset, actions, nullptr, (hipStream_t)stream);
  if (rc) return rc;
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_forward_sample_synth_step(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, float* probs,
                                   float* values, uint64_t seed, const uint64_t* step_base_dev, uint64_t step_offset,
                                   uint32_t env_offset, int32_t* actions, uint64_t env_seed, uint32_t terminal_threshold,
                                   uint8_t* stack_out, float* rewards_out, float* masks_out, float* ep_reward,
                                   int32_t* ep_len, void* finished, paac_stream_t stream) {
  PAAC_REQUIRE(ctx && params && states && actions && stack_out && rewards_out && masks_out && ep_reward && ep_len,
               "paac_forward_sample_synth_step: null argument");
  PAAC_REQUIRE(batch > 0 && batch <= ctx->max_batch, "paac_forward_sample_synth_step: batch %d outside (0, max_batch=%d]",
               batch, ctx->max_batch);
  PAAC_REQUIRE(states != stack_out, "paac_forward_sample_synth_step: the step cannot shift the stacks in place");
  const int rc = launch_forward_sample_step(ctx, params, states, batch, probs, values, seed, step_base_dev, step_offset,
                                            env_offset, actions, env_seed, terminal_threshold, stack_out, rewards_out,
                                            masks_out, ep_reward, ep_len, finished, (hipStream_t)stream);
  if (rc) return rc;
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_act_step_mt(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, uint32_t* mt_state,
                     int32_t* actions, float* probs_out, float* values_out, uint64_t env_seed, uint32_t env_offset,
                     uint32_t terminal_threshold, const uint64_t* step_base_dev, uint64_t step_offset, uint8_t* stack_out,
                     uint8_t* stack_out2, float* rewards_out, float* masks_out, float* ep_reward, int32_t* ep_len,
                     void* finished, uint8_t* raw_scratch, void* walk_scratch, int64_t walk_scratch_bytes,
                     paac_stream_t stream) {
  PAAC_REQUIRE(ctx && params && states && mt_state && actions && probs_out && values_out, "paac_act_step_mt: null argument");
  if (stack_out == nullptr) {
    // policy forward + sampler only (the environments live on the host: paac.py:104-110 without :115-116): three launches up
    // to PAAC_ACT_STEP_MAX_ENVS environments, paac_forward + paac_sample_mt beyond
    PAAC_REQUIRE(!stack_out2 && !raw_scratch, "paac_act_step_mt: no environment step (stack_out == NULL) takes no stack_out2 / "
                 "raw_scratch");
    PAAC_REQUIRE(batch > 0 && batch <= ctx->max_batch && batch <= PAAC_ACT_STEP_MAX_ENVS &&
                 (int64_t)batch * (ctx->cfg.num_actions - 1) <= 1024,
                 "paac_act_step_mt: sampling without an environment step covers up to %d environments and 1024 draws (use "
                 "paac_forward + paac_sample_mt); got %d x %d actions", PAAC_ACT_STEP_MAX_ENVS, batch, ctx->cfg.num_actions);
    const float *partial, *ba, *bc;
    int ntiles;
    int rc = launch_forward_trunk(ctx, params, states, batch, &partial, &ntiles, &ba, &bc, (hipStream_t)stream);
    if (rc) return rc;
    rc = launch_sample_env_step_heads(partial, ntiles, ba, bc, probs_out, values_out, ctx->cfg.num_actions, mt_state, actions, 0, 0,
                                      batch, 0, nullptr, 0, states, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                      nullptr, nullptr, (hipStream_t)stream);
    if (rc) return rc;
    PAAC_CHECK_HIP(hipGetLastError());
    return 0;
  }
  PAAC_REQUIRE(rewards_out && masks_out && ep_reward && ep_len, "paac_act_step_mt: null argument");
  PAAC_REQUIRE(batch > 0 && batch <= ctx->max_batch && batch <= PAAC_ACT_STEP_MAX_ENVS_LARGE,
               "paac_act_step_mt: batch %d outside (0, min(max_batch=%d, %d)]", batch, ctx->max_batch,
               PAAC_ACT_STEP_MAX_ENVS_LARGE);
  PAAC_REQUIRE((int64_t)batch * (ctx->cfg.num_actions - 1) <= PAAC_FUSED_SAMPLE_MAX_DRAWS,
               "paac_act_step_mt: N*(A-1) = %ld exceeds %d (use paac_forward + paac_sample_mt + paac_synth_step)",
               (long)batch * (ctx->cfg.num_actions - 1), PAAC_FUSED_SAMPLE_MAX_DRAWS);
  PAAC_REQUIRE(states != stack_out && states != stack_out2, "paac_act_step_mt: the step cannot shift the stacks in place");
  if (batch > PAAC_ACT_STEP_MAX_ENVS || (int64_t)batch * (ctx->cfg.num_actions - 1) > 1024) {
    // the large shards (128 x 18, 256 x 4): policy forward with its heads finish, then the sampler + environment-step launch
    // with the walks spread over several workgroups -- whose MT19937 doubles a spare workgroup of the fc launch makes
    // meanwhile (csrc/mt_ahead.h)
    const bool tail = (ctx->tower_on || ctx->tower2_on) && ctx->managed_weights && batch <= PAAC_ACT_STEP_MAX_ENVS_LARGE;
    if (tail) {
      ctx->ahead_state = mt_state;
      ctx->ahead_D = batch * (ctx->cfg.num_actions - 1);
    }
    // (three launches when the walks can be spread -- walk_scratch lent -- and the action set is small: the sampler workgroups
    // then also finish the heads of the environments they visit, from the fc kernel's per-tile partials.  Measured: 256 x 4
    // 16.2 + 4.3 us -> 18.7 us per step, 1.898 -> 1.927 M env-steps/s; 128 x 18: 18.4 + 4.5 -> 24.0 us, SLOWER -- up to 24
    // rows x 19 outputs x 32 partials and an 18-way softmax at the head of every sampler workgroup's chain cost more than the
    // launch they replace -- so the fold is taken up to 8 actions.  PAAC_HEADS_IN_SAMPLER=0 / =1 force it off / on.)
    static const int fold_knob = [] {
      const char* v = getenv("PAAC_HEADS_IN_SAMPLER");
      return (v && *v) ? (atoi(v) != 0 ? 1 : 0) : -1;
    }();
    const bool fold = fold_knob >= 0 ? fold_knob == 1 : ctx->cfg.num_actions <= 8;
    HeadsPartials hp{nullptr, 0, nullptr, nullptr, values_out};
    int rc;
    if (tail && fold && sampler_folds_heads(batch, ctx->cfg.num_actions, walk_scratch))
      rc = launch_forward_trunk(ctx, params, states, batch, &hp.partial, &hp.ntiles, &hp.ba, &hp.bc, (hipStream_t)stream);
    else
      rc = launch_forward(ctx, 0, params, states, batch, nullptr, probs_out, values_out, (hipStream_t)stream);
    ctx->ahead_state = nullptr;
    if (rc) return rc;
    rc = launch_sample_mt_synth_step(probs_out, ctx->cfg.num_actions, mt_state, actions, env_seed, env_offset, batch,
                                     terminal_threshold, step_base_dev, step_offset, states, stack_out, stack_out2, rewards_out,
                                     masks_out, ep_reward, ep_len, finished, walk_scratch, walk_scratch_bytes, raw_scratch,
                                     tail ? ctx->mt_ahead : nullptr, (hipStream_t)stream, hp.partial ? &hp : nullptr);
    if (rc) return rc;
    PAAC_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const float *partial, *ba, *bc;
  int ntiles;
  // Up to 64 environments the sampler's state blocks and doubles already hide behind the loads of the head partials
  // (misc.hip: synth_step_a_mth_kernel): making them one launch ahead (csrc/mt_ahead.h) measured SLOWER here -- 658 k against
  // 669 k env-steps/s at the headline shape, the extra loads sit at the head of the sampler workgroup's chain --, so it is off
  // unless PAAC_MT_AHEAD_SMALL=1 asks for it.  The large shards (above) use it.
  static const bool ahead_on = [] {
    const char* v = getenv("PAAC_MT_AHEAD_SMALL");
    return v && *v && atoi(v) != 0;
  }();
  if (ahead_on) {
    ctx->ahead_state = mt_state;
    ctx->ahead_D = batch * (ctx->cfg.num_actions - 1);
  }
  int rc = launch_forward_trunk(ctx, params, states, batch, &partial, &ntiles, &ba, &bc, (hipStream_t)stream);
  ctx->ahead_state = nullptr;
  if (rc) return rc;
  rc = launch_sample_env_step_heads(partial, ntiles, ba, bc, probs_out, values_out, ctx->cfg.num_actions, mt_state, actions,
                                    env_seed, env_offset, batch, terminal_threshold, step_base_dev, step_offset, states,
                                    stack_out, stack_out2, rewards_out, masks_out, ep_reward, ep_len, finished, raw_scratch,
                                    ahead_on ? ctx->mt_ahead : nullptr, (hipStream_t)stream);
  if (rc) return rc;
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_loss_backward(paac_ctx* ctx, const float* params, const uint8_t* states, const int32_t* actions, const float* y,
                       const float* adv, int batch, float entropy_beta, float* grad, float* loss_out,
                       int forward_done, int phase, paac_stream_t stream) {
  PAAC_REQUIRE(ctx && params && states && actions && y && adv && grad, "paac_loss_backward: null argument");
  PAAC_REQUIRE(batch > 0 && batch <= ctx->max_batch, "paac_loss_backward: batch %d outside (0, max_batch=%d]", batch,
               ctx->max_batch);
  PAAC_REQUIRE(phase >= 0 && phase <= 3, "paac_loss_backward: phase %d", phase);
  int rc = 0;
  if (!forward_done && phase != 2) {
    rc = launch_forward(ctx, 1, params, states, batch, nullptr, nullptr, nullptr, (hipStream_t)stream);
    if (rc) return rc;
  }
  ctx->last_ws = 1;
  rc = launch_backward(ctx, params, states, actions, y, adv, batch, entropy_beta, grad, loss_out, phase,
                       (hipStream_t)stream);
  if (rc) return rc;
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int paac_loss_backward_returns(paac_ctx* ctx, const float* params, const uint8_t* states, const int32_t* actions,
                               const paac_returns* ret, int batch, float entropy_beta, float* grad, float* loss_out,
                               int forward_done, int phase, paac_stream_t stream) {
  PAAC_REQUIRE(ctx && params && states && actions && ret && grad, "paac_loss_backward_returns: null argument");
  PAAC_REQUIRE(batch > 0 && batch <= ctx->max_batch, "paac_loss_backward_returns: batch %d outside (0, max_batch=%d]", batch,
               ctx->max_batch);
  PAAC_REQUIRE(phase >= 0 && phase <= 3, "paac_loss_backward_returns: phase %d", phase);
  PAAC_REQUIRE(ret->T > 0 && ret->N > 0 && ret->T * ret->N == batch, "paac_loss_backward_returns: T*N = %d*%d != batch %d",
               ret->T, ret->N, batch);
  PAAC_REQUIRE(ret->rewards && ret->masks && ret->values && ret->y_out && ret->adv_out,
               "paac_loss_backward_returns: null rollout record");
  PAAC_REQUIRE(ret->v_boot || (forward_done && batch + ret->N <= ctx->max_batch),
               "paac_loss_backward_returns: v_boot == NULL takes the bootstrap values from rows [batch, batch + N) of a "
               "training forward that has already run over batch + N rows");
  PAAC_REQUIRE(!ret->global_step_dev || (ret->lr_out_dev && ret->lr_annealing_steps > 0),
               "paac_loss_backward_returns: schedule bookkeeping needs lr_out_dev and lr_annealing_steps");
  int rc = 0;
  if (!forward_done && phase != 2) {
    rc = launch_forward(ctx, 1, params, states, batch, nullptr, nullptr, nullptr, (hipStream_t)stream);
    if (rc) return rc;
  }
  ctx->last_ws = 1;
  rc = launch_backward(ctx, params, states, actions, ret->y_out, ret->adv_out, batch, entropy_beta, grad, loss_out, phase,
                       (hipStream_t)stream, ret);
  if (rc) return rc;
  PAAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int64_t paac_debug_activation(paac_ctx* ctx, int what, int batch, float* out, paac_stream_t stream) {
  PAAC_REQUIRE(ctx && out && batch > 0 && batch <= ctx->max_batch, "paac_debug_activation: bad arguments");
  const float* src = nullptr;
  int64_t n = 0;
  const bool training_set = what >= 21 && what <= 24;      // 21..24: a1..a3 / h of the TRAINING set whichever was used last
  if (training_set) what -= 20;
  const Workspace& W = ctx->ws[training_set ? 1 : ctx->last_ws];
  if (what >= 1 && what <= ctx->spec.nconv) {
    const ConvSpec& cs = ctx->spec.conv[what - 1];
    src = W.act[what - 1];
    n = (int64_t)batch * cs.oh * cs.ow * cs.cout;
  } else if (what == 4) {
    src = W.h;
    n = (int64_t)batch * ctx->spec.fc;
  } else if (what >= 11 && what <= 10 + ctx->spec.nconv) {
    const ConvSpec& cs = ctx->spec.conv[what - 11];
    src = ctx->dact[what - 11];
    n = (int64_t)batch * cs.oh * cs.ow * cs.cout;
  } else if (what == 14) {
    src = ctx->dh;
    n = (int64_t)batch * ctx->spec.fc;
  } else {
    set_error("paac_debug_activation: what=%d", what);
    return -1;
  }
  PAAC_CHECK_HIP(hipMemcpyAsync(out, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return n;
}

int paac_debug_set_tuning(paac_ctx* ctx, int op, int batch_class, int cfg, int ksplit, int xcd_dim) {
  PAAC_REQUIRE(ctx && op >= 0 && op < OP_COUNT && batch_class >= 0 && batch_class <= 2, "paac_debug_set_tuning: bad op/class");
  ctx->tune[op][batch_class] = Tune{cfg, ksplit, xcd_dim};
  return 0;
}

int paac_debug_get_tuning(paac_ctx* ctx, int op, int batch_class, int* cfg, int* ksplit, int* xcd_dim) {
  PAAC_REQUIRE(ctx && op >= 0 && op < OP_COUNT && batch_class >= 0 && batch_class <= 2 && cfg && ksplit && xcd_dim,
               "paac_debug_get_tuning: bad argument");
  const Tune t = ctx->tune[op][batch_class];
  *cfg = t.cfg;
  *ksplit = t.ksplit;
  *xcd_dim = t.xcd;
  return 0;
}

// ---- hipGraph helpers -------------------------------------------------------------------------
int paac_graph_begin(paac_stream_t stream) {
  PAAC_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return 0;
}

int paac_graph_end(paac_stream_t stream, paac_graph** out) {
  PAAC_REQUIRE(out, "paac_graph_end: null out");
  hipGraph_t g = nullptr;
  PAAC_CHECK_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t ex = nullptr;
  hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    set_error("hipGraphInstantiate -> %s", hipGetErrorString(e));
    return -2;
  }
  paac_graph* pg = new paac_graph{g, ex};
  *out = pg;
  return 0;
}

int paac_graph_launch(paac_graph* g, paac_stream_t stream) {
  PAAC_REQUIRE(g, "paac_graph_launch: null graph");
  PAAC_CHECK_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
  return 0;
}

int paac_graph_destroy(paac_graph* g) {
  if (!g) return 0;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
  return 0;
}

// ---- timing hooks -----------------------------------------------------------------------------
int paac_prof_enable(paac_ctx* ctx, int on) {
  PAAC_REQUIRE(ctx, "paac_prof_enable: null ctx");
  if (on) {
    for (int i = 0; i < paac_ctx::PROF_MAX_EVENTS; ++i) {
      if (!ctx->ev_start[i]) PAAC_CHECK_HIP(hipEventCreate(&ctx->ev_start[i]));
      if (!ctx->ev_stop[i]) PAAC_CHECK_HIP(hipEventCreate(&ctx->ev_stop[i]));
    }
  }
  ctx->prof_on = on ? 1 : 0;
  if (on) g_prof_ctx = ctx;
  else if (g_prof_ctx == ctx) g_prof_ctx = nullptr;
  return 0;
}

int paac_prof_read(paac_ctx* ctx, int32_t* family_out, int32_t* batch_out, float* ms_out, int max_events) {
  PAAC_REQUIRE(ctx, "paac_prof_read: null ctx");
  int n = 0;
  for (int i = 0; i < ctx->ev_count; ++i) {
    PAAC_CHECK_HIP(hipEventSynchronize(ctx->ev_stop[i]));
    float ms = 0.f;
    PAAC_CHECK_HIP(hipEventElapsedTime(&ms, ctx->ev_start[i], ctx->ev_stop[i]));
    if (n < max_events) {
      if (family_out) family_out[n] = ctx->ev_family[i];
      if (batch_out) batch_out[n] = ctx->ev_batch[i];
      if (ms_out) ms_out[n] = ms;
      ++n;
    }
  }
  ctx->ev_count = 0;
  return n;
}

int paac_prof_read_mix(paac_ctx* ctx, int32_t* mix_out, int max_events) {
  PAAC_REQUIRE(ctx && mix_out, "paac_prof_read_mix: null argument");
  int n = 0;
  for (int i = 0; i < ctx->ev_count && n < max_events; ++i) mix_out[n++] = ctx->ev_mix[i];
  return n;
}

int paac_user_arch(int32_t* nconv, int32_t* filters3, int32_t* fc_width) {
#ifdef PAAC_USER_ARCH
  if (nconv) *nconv = PAAC_USER_NCONV;
  if (filters3) {
    filters3[0] = PAAC_USER_C1;
    filters3[1] = PAAC_USER_C2;
    filters3[2] = PAAC_USER_NCONV == 3 ? PAAC_USER_C3 : 0;
  }
  if (fc_width) *fc_width = PAAC_USER_H;
  return 1;
#else
  if (nconv) *nconv = 0;
  if (fc_width) *fc_width = 0;
  if (filters3) filters3[0] = filters3[1] = filters3[2] = 0;
  return 0;
#endif
}

int paac_user_arch_layers(int32_t* sizes3, int32_t* strides3) {
#ifdef PAAC_USER_ARCH
  const ArchSpec s = arch_spec(PAAC_ARCH_USER);
  for (int i = 0; i < 3; ++i) {
    if (sizes3) sizes3[i] = i < s.nconv ? s.conv[i].k : 0;
    if (strides3) strides3[i] = i < s.nconv ? s.conv[i].stride : 0;
  }
  return 1;
#else
  for (int i = 0; i < 3; ++i) {
    if (sizes3) sizes3[i] = 0;
    if (strides3) strides3[i] = 0;
  }
  return 0;
#endif
}

const char* paac_prof_name(int family) {
  if (family < 0 || family >= PAAC_PROF_FAMILIES) return "";
  return kFamilyNames[family];
}

}  // extern "C"
