/*
 * paac_hip.h -- C-ABI of libpaac_hip.so: the MI355X (gfx950) hot path of PAAC.
 *
 * The reference (arjunchandra/paac) is pure Python/TensorFlow-1 and has no FFI; the entry points
 * below are what a binding for its hot path would call, one per kernel family.  Each entry cites
 * the reference code it replaces (file:line relative to the reference root).
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; paac_last_error() gives the message
 *     (thread-local).  No exceptions cross the ABI.
 *   - the CALLER owns every tensor (device pointers + explicit dims); the library owns only the
 *     opaque paac_ctx (activation/slab workspace) and paac_graph handles.
 *   - every launch goes to the caller-supplied hipStream_t (passed as void*); no hidden host
 *     synchronisation, no allocation after paac_create -> every entry is hipGraph-capturable.
 *   - a ctx is not thread-safe; use one per process/GPU.
 *   - all floating point is fp32 (the reference graph is fp32); the n-step return scan is fp64
 *     like the reference's numpy buffers.
 */
#ifndef PAAC_HIP_H
#define PAAC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct paac_ctx paac_ctx;
typedef struct paac_graph paac_graph;
typedef void* paac_stream_t; /* hipStream_t */

enum { PAAC_ARCH_NIPS = 0, PAAC_ARCH_NATURE = 1,     /* networks.py:138-151 / :154-169 */
       /* a user architecture (networks.py:117-120, README.md:80-83: "subclass the trunk"): the same trunk family -- conv
        * 8x8 / 4, conv 4x4 / 2 [, conv 3x3 / 1], fc -- with the user's filter counts (multiples of 16) and fc width (a
        * multiple of 256).  An architecture here is a compiled geometry: paac_amd/build.py builds a library for it on
        * demand (-DPAAC_USER_ARCH ...), which then serves PAAC_ARCH_NATURE and PAAC_ARCH_USER (not PAAC_ARCH_NIPS). */
       PAAC_ARCH_USER = 2 };
enum { PAAC_CLIP_IGNORE = 0, PAAC_CLIP_GLOBAL = 1 }; /* actor_learner.py:51-59 ('local' is broken upstream) */

#define PAAC_MAX_TENSORS 12
#define PAAC_OBS_BYTES 28224 /* 84*84*4 */
#define PAAC_RAW_H 210
#define PAAC_RAW_W 160

/* Flat parameter layout, TF variable-creation order (actor_learner.py:44 grads_and_vars order;
 * pretrained checkpoints .index): conv1_w [8,8,4,C1], conv1_b, conv2_w, conv2_b, (conv3_w, conv3_b,)
 * fcN_w [K,H], fcN_b, actor_w [H,A], actor_b, critic_w [H,1], critic_b.  Every tensor starts on a
 * 4-float boundary; pad floats are zero and stay zero. */
typedef struct {
  int32_t num_tensors;
  int64_t total;          /* floats, padded */
  int64_t total_unpadded; /* the reference's parameter count P */
  int64_t offset[PAAC_MAX_TENSORS];
  int64_t size[PAAC_MAX_TENSORS];
  int32_t rank[PAAC_MAX_TENSORS];
  int32_t shape[PAAC_MAX_TENSORS][4];
  char name[PAAC_MAX_TENSORS][32];
} paac_layout;

typedef struct {
  int32_t device;      /* HIP device ordinal */
  int32_t arch;        /* PAAC_ARCH_* */
  int32_t num_actions; /* A, 2..32 */
  int32_t max_batch;   /* largest batch any forward/backward will see (N*T for training) */
} paac_cfg;

const char* paac_last_error(void);
int paac_version(void);

int paac_param_layout(int arch, int num_actions, paac_layout* out);

int paac_create(const paac_cfg* cfg, paac_ctx** out);
int paac_destroy(paac_ctx* ctx);

/* Policy/value inference: networks.py:100-169 + policy_v_network.py:24-37 (what
 * paac.py:20-23 and :140-142 fetch).  states u8 [batch,84,84,4] NHWC.  Any of logits/probs/values
 * may be NULL.  Activations stay in the ctx for a following paac_backward on the same batch. */
int paac_forward(paac_ctx* ctx, const float* params, const uint8_t* states, int batch,
                 float* logits, float* probs, float* values, paac_stream_t stream);

/* paac_forward fused with the counter-based categorical sampler (paac_sample_philox semantics) in the heads
 * kernel: what one PAACLearner.choose_next_actions call (paac.py:18-29) costs on the device-resident loop.
 * probs/values nullable; actions int32[batch]. */
int paac_forward_sample(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, float* probs,
                        float* values, uint64_t seed, const uint64_t* step_base_dev, uint64_t step_offset,
                        uint32_t env_offset, int32_t* actions, paac_stream_t stream);

/* paac_forward_sample with the step of the device-resident synthetic environments (paac_synth_step, path A) inside
 * the heads launch: row i's workgroup samples action i and does environment i's bookkeeping, extra workgroups shift
 * the observation stacks `states` -> `stack_out`.  One launch less per rollout step; same results as
 * paac_forward_sample followed by paac_synth_step(env_seed, ...). */
int paac_forward_sample_synth_step(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, float* probs,
                                   float* values, uint64_t seed, const uint64_t* step_base_dev, uint64_t step_offset,
                                   uint32_t env_offset, int32_t* actions, uint64_t env_seed, uint32_t terminal_threshold,
                                   uint8_t* stack_out, float* rewards_out, float* masks_out, float* ep_reward,
                                   int32_t* ep_len, void* finished, paac_stream_t stream);

/* Training forward alone (into the ctx's TRAINING activation set, separate from the one paac_forward* use).
 * Follow with paac_loss_backward(forward_done=1) on the same states with the same OR A SMALLER batch: activations
 * are row-major per sample, so a caller may append the bootstrap observations (paac.py:140-142) as extra rows,
 * read their values here, compute the returns, and run the backward over the first N*T rows only.
 * values: nullable f32[batch]. */
int paac_train_forward(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, float* values,
                       paac_stream_t stream);
/* The same without the heads (it stops after the fc layer): the next paac_loss_backward[_returns](forward_done=1) on this
 * ctx finishes them -- inside its first launch when it is a whole backward (phase 0 / 3) of the three-conv network (one
 * launch less per update; every value bit-identical), as a launch of their own otherwise.  Batches of up to 64 rows run
 * the whole forward.  With paac_loss_backward_returns and ret->v_boot == NULL the bootstrap values are taken from rows
 * [batch, batch + N) of this forward (paac.py:140-142: the bootstrap observations appended to the rollout rows). */
int paac_train_forward_trunk(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, paac_stream_t stream);

/* An update without a training forward.  Weights are frozen inside a cycle (paac.py:99-165), so the T acting forwards of a
 * rollout have already computed the activations the update's forward (paac.py:163-165) would recompute.
 * paac_keep_next_forward(ctx, r): the NEXT acting forward on this ctx (paac_forward / paac_act_step_mt of at most 256 rows,
 * three-conv network, managed weights) also leaves its rows' conv1 / conv2 / conv3 outputs and fc activations at rows
 * [r, r + batch) of the ctx's training activation set (one shot; r = -1 cancels).
 * paac_bootstrap_forward_trunk: an acting-shaped forward (conv tower + fc, no heads) of the N bootstrap observations
 * (paac.py:140-142), kept at rows [train_row, train_row + batch) -- after the T acting steps were kept at rows t*N and
 * this call at T*N, paac_loss_backward[_returns](forward_done = 1) runs on the kept rows exactly as after
 * paac_train_forward_trunk (v_boot == NULL: bootstrap values from rows [batch, batch + N)).  Same mathematics as the
 * reference's second session.run over the same observations; the values differ from a recomputed forward only by the
 * summation order of the kernels involved. */
int paac_keep_next_forward(paac_ctx* ctx, int train_row);
int paac_bootstrap_forward_trunk(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, int train_row,
                                 paac_stream_t stream);

/* One whole acting step of the device-resident loop in three launches (paac.py:104-127 for N <= 64 environments with
 * the reference's numpy sampler): policy forward on `states` (conv tower, fc with the head contractions in its epilogue),
 * then ONE launch that finishes the heads (bias, softmax; probabilities and values also written to probs_out [N,A] /
 * values_out [N]), samples the actions exactly like paac_sample_mt (np.random.multinomial(1, p - epsneg) per
 * environment on the MT19937 state, advanced in place) and steps the synthetic environments like paac_synth_step
 * (stack_out = shifted stacks with the new frame, stack_out2 (nullable) = a second copy of them; rewards / masks /
 * episode bookkeeping).
 * raw_scratch (nullable, [N,2,210,160] u8): path B like paac_synth_step's -- the step launch writes the step's raw screen
 * pairs there instead of shifting the stacks, and one more launch (paac_preprocess_stack's kernel: max of the two screens,
 * PIL-nearest resize, history push, reset on terminal) builds stack_out / stack_out2 from them: four launches.
 * stack_out == NULL (with it stack_out2, raw_scratch and the four record pointers): no environment step -- forward + heads
 * finish + sampler only, for environments that live on the host (paac.py:104-110; up to PAAC_ACT_STEP_MAX_ENVS of them).
 * Up to PAAC_ACT_STEP_MAX_ENVS environments and N*(A-1) <= 1024 draws: the three launches above.  Beyond, up to
 * PAAC_ACT_STEP_MAX_ENVS_LARGE environments and PAAC_FUSED_SAMPLE_MAX_DRAWS draws (the 128 x 18 and 256 x 4 shards):
 * paac_forward + paac_sample_mt_synth_step in one call (four launches; lend walk_scratch as there), with the sampler's
 * MT19937 doubles made one launch ahead by a spare workgroup of the fc launch (they depend on nothing but the stream
 * position), so that the sampler workgroups load them instead of each rebuilding the state blocks. */
#define PAAC_ACT_STEP_MAX_ENVS 64
#define PAAC_ACT_STEP_MAX_ENVS_LARGE 256
int paac_act_step_mt(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, uint32_t* mt_state,
                     int32_t* actions, float* probs_out, float* values_out, uint64_t env_seed, uint32_t env_offset,
                     uint32_t terminal_threshold, const uint64_t* step_base_dev, uint64_t step_offset, uint8_t* stack_out,
                     uint8_t* stack_out2, float* rewards_out, float* masks_out, float* ep_reward, int32_t* ep_len,
                     void* finished, uint8_t* raw_scratch, void* walk_scratch, int64_t walk_scratch_bytes,
                     paac_stream_t stream);

/* Conv-weight packing.  The Nature conv layers run as one fused launch that reads the conv weights pre-split into bf16
 * planes (an internal copy owned by the ctx).  By default every paac_forward* / paac_train_forward / paac_loss_backward
 * call refreshes that copy from `params` first (one small extra launch), so a caller may change `params` at any time.
 * A caller that owns every write to `params` can switch to managed mode: paac_set_managed_weights(ctx, 1) -- then the
 * copy is refreshed only by paac_clip_rmsprop (right behind the optimizer step) and by an explicit paac_pack_weights
 * (call it after initialising, restoring or broadcasting `params`); in managed mode the acting forwards also stop
 * keeping the conv1 / conv2 activations (only the training forward keeps them, for the backward pass). */
int paac_pack_weights(paac_ctx* ctx, const float* params, paac_stream_t stream);
int paac_set_managed_weights(paac_ctx* ctx, int on);

/* Loss + gradients of policy_v_network.py:29-57 through the whole network (what
 * optimizer.compute_gradients(loss), actor_learner.py:44, evaluates): runs the training forward
 * on `states` (unless forward_done != 0: paac_train_forward already ran on the same batch), then backward.
 * phase: 0 = everything; 1 = forward (unless done) + heads + fc layer -> the gradients of fc_w .. critic_b, i.e. the
 * contiguous tail [offset(fc_w), total) of the flat buffer (95 % of its bytes); 2 = conv layers -> the head
 * [0, offset(fc_w)).  A data-parallel caller all-reduces the tail while phase 2 still runs.  3 = everything, except that
 * the split-K slabs of the conv weight gradients are summed into `grad` by the NEXT paac_clip_rmsprop on this ctx and
 * this `grad` (its norm pass does it, in the same order, so norm and update are bit-identical to phase 0; one launch
 * less): until then the conv part of `grad` is not valid -- for a caller that goes straight to the optimizer step.
 * actions = sampled action index per row (the one-hot's argmax,
 * paac.py:27), y = critic target, adv = advantage, batch rows t-major (paac.py:151-154).
 * grad: flat, padded layout.  loss_out (nullable, device float[4]) = {loss, actor, critic, mean entropy}. */
int paac_loss_backward(paac_ctx* ctx, const float* params, const uint8_t* states, const int32_t* actions,
                       const float* y, const float* adv, int batch, float entropy_beta,
                       float* grad, float* loss_out, int forward_done, int phase, paac_stream_t stream);

/* paac_nstep_returns_tick + paac_loss_backward with the returns computed inside the backward's first launch (the heads
 * gradient kernel derives y / adv of every row from the rollout records; its last workgroup writes y_out / adv_out and
 * does the global_step / lr / frame-counter bookkeeping): one launch less per update, same values bit for bit.
 * batch must equal T*N (rows t-major, paac.py:151-154).  With phase == 2 (conv part only) `ret` is not used. */
typedef struct {
  const float* v_boot;        /* [N] bootstrap values (float32 network output, paac.py:140-142) */
  const float* rewards;       /* [T,N] clipped rewards */
  const float* masks;         /* [T,N] 1 - terminal */
  const float* values;        /* [T,N] values of the acting forwards */
  int32_t T, N;
  double gamma;
  float* y_out;               /* [T*N] */
  float* adv_out;             /* [T*N] */
  int64_t* global_step_dev;   /* nullable: no schedule bookkeeping */
  int64_t increment;
  double initial_lr;
  int64_t lr_annealing_steps;
  float* lr_out_dev;
  uint64_t* tick_dev;         /* nullable */
  uint64_t tick_inc;
} paac_returns;
int paac_loss_backward_returns(paac_ctx* ctx, const float* params, const uint8_t* states, const int32_t* actions,
                               const paac_returns* ret, int batch, float entropy_beta, float* grad, float* loss_out,
                               int forward_done, int phase, paac_stream_t stream);

/* tf.clip_by_global_norm + RMSPropOptimizer.apply_gradients (actor_learner.py:31-34,56-59,70):
 *   g <- grad * grad_scale           (grad_scale = 1/world_size after the sum all-reduce)
 *   gn = sqrt(sum g^2); g <- g * clip_norm*min(1/gn, 1/clip_norm)  (mode GLOBAL)
 *   ms += (g^2 - ms)(1-decay); mom = momentum*mom + lr*g/sqrt(ms+eps); var -= mom
 * lr is read from device memory (*lr_dev) so the call can sit in a replayed graph.
 * gnorm_out (nullable): device float receiving gn.  After paac_loss_backward(phase = 3) on the same `grad` the norm pass
 * first completes the conv part of `grad` (which is therefore written although the parameter is const). */
int paac_clip_rmsprop(paac_ctx* ctx, float* params, const float* grad, float* ms, float* mom, int64_t n,
                      const float* lr_dev, float decay, float momentum, float eps, float clip_norm,
                      int clip_mode, float grad_scale, float* gnorm_out, paac_stream_t stream);

/* The reference's gradient summaries (actor_learner.py:85-87 -> logger_utils.py:23-33: mean, stddev, max, min of the
 * flat raw gradient and of the flat clipped gradient, plus global_norm): the reductions ride along the norm pass of
 * the LAST paac_clip_rmsprop on this ctx (no extra pass over the gradient); this call only folds its per-block
 * partials.  stats_out: device float[8] = {sum, sum of squares, max, min, number of exact zeros, 0, 0, 0} of the raw
 * flat gradient (g * grad_scale) over the reference's P elements (alignment pads excluded).  The clipped gradient is
 * the raw one times clip_norm*min(1/gn, 1/clip_norm), so its statistics follow.  Call at the logging cadence. */
int paac_grad_stats(paac_ctx* ctx, float* stats_out, paac_stream_t stream);

/* actor_learner.py:119-123 + paac.py:127: *global_step += increment; *lr_out = f32(lr0 - step*lr0/anneal)
 * (0 beyond anneal); evaluated in fp64 like the reference's Python float. */
int paac_lr_step(int64_t* global_step_dev, int64_t increment, double initial_lr, int64_t lr_annealing_steps,
                 float* lr_out_dev, paac_stream_t stream);

/* paac.py:140-149: R = v_boot; for t = T-1..0: R = r_t + gamma*R*m_t; y_t = R; adv_t = R - V_t.
 * fp64 scan, fp32 in/out.  rewards/masks/values/y/adv are [T,N] (t-major). */
int paac_nstep_returns(const float* v_boot, const float* rewards, const float* masks, const float* values,
                       int T, int N, double gamma, float* y, float* adv, paac_stream_t stream);

/* paac_nstep_returns + paac_lr_step (+ an optional counter bump) in ONE launch: the end-of-rollout bookkeeping of
 * paac.py:127,140-156.  tick_dev may be NULL. */
int paac_nstep_returns_tick(const float* v_boot, const float* rewards, const float* masks, const float* values,
                            int T, int N, double gamma, float* y, float* adv, int64_t* global_step_dev,
                            int64_t increment, double initial_lr, int64_t lr_annealing_steps, float* lr_out_dev,
                            uint64_t* tick_dev, uint64_t tick_inc, paac_stream_t stream);

/* paac.py:34-45 bit-exact: probs - float32.epsneg, then numpy legacy multinomial(1, p) per env in
 * index order on ONE MT19937 stream.  mt_state: device uint32[625] = numpy key[624] + pos, advanced
 * in place (import/export with np.random.get_state()/set_state()).  scratch: device, >=
 * paac_sample_mt_scratch_bytes(N, A). */
int64_t paac_sample_mt_scratch_bytes(int N, int A);
int paac_sample_mt(const float* probs, int N, int A, uint32_t* mt_state, void* scratch, int32_t* actions,
                   paac_stream_t stream);

/* Throughput sampler (build's own spec, oracle/sampler.py:sample_philox): u = philox4x32-10
 * (ctr = {env_offset+e, step lo, step hi, 0}; key = seed) with step = *step_base_dev + step_offset,
 * action = inverse CDF on fp32 running sums.  The base lives in device memory and the offset is an
 * immediate so a captured graph of T steps replays with a fresh base (paac_counter_add, once per cycle). */
int paac_sample_philox(const float* probs, int N, int A, uint64_t seed, const uint64_t* step_base_dev,
                       uint64_t step_offset, uint32_t env_offset, int32_t* actions, paac_stream_t stream);
int paac_counter_add(uint64_t* counter_dev, uint64_t inc, paac_stream_t stream);

/* emulator_runner.py:18-33 frame path on device: FramePool max over 2 raw frames
 * (atari_emulator.py:72), PIL-nearest resize 210x160 -> 84x84 (:73), ObservationPool push + rotated
 * read-out (environment.py:66-71).  raw: u8 [N,2,210,160] (gray) or [N,2,210,160,3] (rgb, converted with
 * the ITU-R 601 fixed-point luma).  stack_in/stack_out: u8 [N,84,84,4] oldest..newest (may alias).
 * push_mask (nullable): u8[N], 0 = copy the env's stack through unchanged.
 * reset_mask (nullable): u8[N], !=0 = the three older channels are cleared before the push. */
int paac_preprocess_stack(const uint8_t* raw, int is_rgb, int N, const uint8_t* stack_in, uint8_t* stack_out,
                          const uint8_t* push_mask, const uint8_t* reset_mask, paac_stream_t stream);

/* Device-resident synthetic environments (the metric's "synthetic 84x84x4 uint8 frames"; spec in
 * paac_amd/synthetic.py, a BaseEnvironment plugin producing the same numbers on the host).
 * One call replaces one Runners.update_environments()/wait_updated() round (runners.py:44-50) plus the
 * per-env bookkeeping of paac.py:119-138 for N envs.  The frame id of the step is
 * *step_base_dev + step_offset + 1 (id 0 is the reset frame):
 *   stack_in -> stack_out (and stack_out2 if non-NULL); auto-reset on terminal like
 *   emulator_runner.py:26-27; rewards_out f32[N] = clipped reward (actor_learner.py:95-101);
 *   masks_out f32[N] = 1 - terminal; ep_reward f32[N] / ep_len i32[N] running episode totals;
 *   finished: {i32 count; i32 pad; f32 reward[4096]; i32 len[4096]} ring of finished episodes.
 * raw_scratch: NULL = path A (one new 84x84 plane per step); else u8 [N,2,210,160] = path B (two raw
 * frames are generated there, then max + resize + stack via the preprocess kernel). */
int paac_synth_reset(uint64_t seed, uint32_t env_offset, int N, uint8_t* stack_out, uint8_t* raw_scratch,
                     paac_stream_t stream);
int paac_synth_step(uint64_t seed, uint32_t env_offset, int N, const int32_t* actions, uint32_t terminal_threshold,
                    const uint64_t* step_base_dev, uint64_t step_offset, const uint8_t* stack_in, uint8_t* stack_out,
                    uint8_t* stack_out2, float* rewards_out, float* masks_out, float* ep_reward, int32_t* ep_len,
                    void* finished, uint8_t* raw_scratch, paac_stream_t stream);

/* paac_sample_mt + paac_synth_step (path A) in ONE launch: workgroup 0 samples (numpy-parity MT19937 stream) and does
 * the per-env bookkeeping while the other workgroups shift the observation stacks (stack_out2, nullable: a second copy
 * of the new stacks, like paac_synth_step's).  Limit: N*(A-1) <= 2304 (covers 256 environments x 4 actions and
 * 128 x 18).
 * walk_scratch (nullable): device memory of paac_walk_scratch_bytes(N, A) bytes, ZERO-INITIALISED once by the caller, then
 * left to the library and lent to every call of the same (N, A) in one stream order.  With it the large shards (more than
 * 64 environments or 1024 draws) spread the sampler's walk over several workgroups of the launch (same actions, same
 * stream position; 49 -> 14 us at 256 environments x 4 actions, 41 -> 24 us at 128 x 18); without it one workgroup walks
 * all environments.
 * raw_scratch (nullable, [N,2,210,160] u8): path B -- the launch's other workgroups write the step's raw screen pairs there
 * instead of shifting, and the preprocess launch (max, PIL-nearest resize, history push) follows, like paac_synth_step's. */
#define PAAC_FUSED_SAMPLE_MAX_DRAWS 2304
int64_t paac_walk_scratch_bytes(int N, int A);
int paac_sample_mt_synth_step(const float* probs, int A, uint32_t* mt_state, int32_t* actions, uint64_t seed,
                              uint32_t env_offset, int N, uint32_t terminal_threshold, const uint64_t* step_base_dev,
                              uint64_t step_offset, const uint8_t* stack_in, uint8_t* stack_out, uint8_t* stack_out2,
                              float* rewards_out, float* masks_out, float* ep_reward, int32_t* ep_len, void* finished,
                              void* walk_scratch, int64_t walk_scratch_bytes, uint8_t* raw_scratch, paac_stream_t stream);

/* hipGraph helpers: capture every launch issued on `stream` between begin/end, replay with launch. */
int paac_graph_begin(paac_stream_t stream);
int paac_graph_end(paac_stream_t stream, paac_graph** out);
int paac_graph_launch(paac_graph* g, paac_stream_t stream);
int paac_graph_destroy(paac_graph* g);

/* Test/debug: copy an internal activation to a caller device buffer (async on stream).
 * what: 1..3 = conv outputs a1..a3 [batch,OH,OW,C], 4 = fc activations h [batch,H] of the activation set used
 * last (acting or training); 21..23 / 24 = the same of the TRAINING set explicitly (rows kept by paac_keep_next_forward);
 * 11..13 / 14 = the gradients wrt them.  Returns the element count. */
int64_t paac_debug_activation(paac_ctx* ctx, int what, int batch, float* out, paac_stream_t stream);

/* Test/debug: sampler workgroup `sampler_workgroup` of paac_act_step_mt's large-shard step launch reports an exactly-zero
 * conditional probability it has not seen (-1: off, the default) -- exercises the rare serial path of the distributed zero
 * detection on ordinary probabilities.  Process-wide; synchronises the device. */
int paac_debug_report_zero(int sampler_workgroup);

/* Tuning (tools/tune_gemm.py): override the launch configuration of GEMM op `op` (0 conv1_fwd, 1 conv2_fwd,
 * 2 conv3_fwd, 3 fc_fwd, 4 fc_wgrad, 5 fc_dgrad, 6 conv3_wgrad, 7 conv3_dgrad, 8 conv2_wgrad, 9 conv2_dgrad,
 * 10 conv1_wgrad, 11 conv tower: cfg = regions per sample, 1 / 2 / 4, -1 = by batch) for batch class 0 (batch <= 64), 1 (batch <= 512) or 2: cfg = index into the family's configuration table
 * (-1 = size heuristic), ksplit = blockIdx.z K split (0 = heuristic), xcd_dim = grid dimension tied to the XCD. */
int paac_debug_set_tuning(paac_ctx* ctx, int op, int batch_class, int cfg, int ksplit, int xcd_dim);
int paac_debug_get_tuning(paac_ctx* ctx, int op, int batch_class, int* cfg, int* ksplit, int* xcd_dim);

/* The user architecture compiled into this library: returns 1 and fills nconv (2 or 3), filters3[3] (0 for an absent third
 * layer) and fc_width; returns 0 (and zeros) for the stock library. */
int paac_user_arch(int32_t* nconv, int32_t* filters3, int32_t* fc_width);
/* ... and its layers' kernel sizes and strides (VALID convolutions, networks.py:12-21; 0 for an absent layer). */
int paac_user_arch_layers(int32_t* sizes3, int32_t* strides3);

/* Diagnostic: writes {s_memtime shader-clock ticks, s_memrealtime 100 MHz ticks} to out2_dev[0..1]. */
int paac_debug_clock(uint64_t* out2_dev, paac_stream_t stream);

/* Per-kernel timing hooks for bench.py's roofline object: when enabled, every network / optimizer kernel
 * launch is bracketed by hipEvents on the launch stream (do not enable inside graph capture).
 * paac_prof_read synchronises the recorded events and returns, per launch, its kernel family, the batch it
 * processed and its duration in ms (up to max_events; the internal table holds 8192 launches); returns the
 * number of records written and clears the table.  The entry points that take no ctx (environment step, samplers,
 * n-step returns, preprocessing) are recorded in the table of the ctx profiling was last enabled on.
 * paac_prof_read_mix (call it BEFORE paac_prof_read, which clears the table; no synchronisation) returns per launch the
 * instruction mix of its contraction bodies, one byte per body in launch order: MFMA products issued per fp32 multiply
 * (1 = fp32 MFMA, 3 = exact-bf16 path, 6 = split-bf16 path; 0 = no contraction) -- so that a family is priced against the
 * ceiling of what it ran. */
#define PAAC_PROF_FAMILIES 26
int paac_prof_enable(paac_ctx* ctx, int on);
int paac_prof_read_mix(paac_ctx* ctx, int32_t* mix_out, int max_events);
int paac_prof_read(paac_ctx* ctx, int32_t* family_out, int32_t* batch_out, float* ms_out, int max_events);
const char* paac_prof_name(int family);

#ifdef __cplusplus
}
#endif
#endif /* PAAC_HIP_H */
