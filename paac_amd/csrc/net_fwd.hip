// Actor-critic network of the PAAC hot path on gfx950: forward, loss, backward.
//   networks.py:100-169 (trunk), policy_v_network.py:6-57 (heads + loss), and the gradient graph
//   optimizer.compute_gradients(loss) builds from them (actor_learner.py:44).
// Layout contract: activations NHWC fp32, conv weights HWIO, fc weights [in,out], flatten in HWC order
// (networks.py:6-9) -- so every weight tensor is already the row-major [K,N] B-matrix of its GEMM.
#include "net_common.h"
#include "tower.h"
#include "tower2.h"
#include "fc_heads.h"
#include "gemm3.h"

namespace paac {

#ifdef PAAC_DMM_STAMPS
unsigned long long* g_tower_stamps = nullptr;
extern "C" void paac_debug_set_tower_stamps(unsigned long long* p) { g_tower_stamps = p; }
unsigned long long* g_stamps = nullptr;
int g_stamp_which = -1, g_stamp_calls = 0;
extern "C" void paac_debug_set_stamps(unsigned long long* p, int which) {
  g_stamps = p;
  g_stamp_which = which;
  g_stamp_calls = 0;
}
#endif

// Forward conv/fc: A = FRAG_K patches, B = FRAG_MN weights [K,N].  N per wave = 16*VN.
template <class G, bool U8, int NDIM, int EPI>
static int launch_fwd(const GemmArgs& g, Tune t, hipStream_t s) {
  constexpr int VN = (NDIM % 64 == 0) ? 4 : (NDIM % 32 == 0) ? 2 : 1;
  int cfg = t.cfg, ksplit = 1, xcd = t.xcd;
  if constexpr (EPI == EPI_SLAB) {
    ksplit = t.ksplit > 0 ? t.ksplit : 0;
    if (ksplit <= 0) {   // heuristic: fill ~768 waves
      const long tiles = (long)((g.M + 15) / 16) * ((g.N + 16 * VN - 1) / (16 * VN));
      ksplit = pick_ksplit(tiles, 4, (g.K + 15) / 16, FC_SPLITS_MAX, 768);
    }
    if (ksplit > FC_SPLITS_MAX) ksplit = FC_SPLITS_MAX;
  }
  if (cfg < 0) {
    const long w = (long)((g.M + 15) / 16) * ((g.N + 16 * VN - 1) / (16 * VN)) * ksplit;
    cfg = (w <= 256) ? 0 : (w <= 1024) ? 3 : (w <= 4096) ? 4 : 5;
    if (U8) cfg += kExactBf16;   // the u8 operand always takes the exact-bf16 path (faster at every size measured)
    xcd = -1;
  }
  if constexpr (U8) {
    if (cfg >= kExactBf16) {   // conv1 on the bf16 MFMA with exactly split weights (dmm.h: XB), same tile table
      switch (cfg - kExactBf16) {
#define X(id, TM, NWM, WK, PF) \
  case id: launch_dmm<Dmm<G, U8, FRAG_K, FRAG_MN, TM, VN, NWM, 1, WK, 1, EPI, false, PF, 1>>(g, ksplit, ksplit, xcd, s); break;
        PAAC_FWD_CFGS(X)
#undef X
        default: break;
      }
      return ksplit;
    }
  }
  if constexpr (!U8 && VN == 4 && EPI == EPI_BIAS_RELU) {
    if (cfg >= kNarrow) {
      switch (cfg - kNarrow) {
#define X(id, TM, NWM, WK, PF) \
  case id: launch_dmm<Dmm<G, U8, FRAG_K, FRAG_MN, TM, 2, NWM, 1, WK, 1, EPI, false, PF>>(g, ksplit, ksplit, xcd, s); return ksplit;
        PAAC_FWD_NARROW_CFGS(X)
#undef X
        default: cfg -= kNarrow; break;
      }
    }
  }
  if (cfg >= kNarrow) cfg -= kNarrow;
  if constexpr (!U8) {
    if (cfg >= kSplitBf16) {
      switch (cfg - kSplitBf16) {
#define X(id, TM, NWM, WK, PF) \
  case id: launch_dmm<Dmm<G, U8, FRAG_K, FRAG_MN, TM, VN, NWM, 1, WK, 1, EPI, false, split_pf(PF), 2>>(g, ksplit, ksplit, xcd, s); return ksplit;
        PAAC_FWD_SPLIT_CFGS(X)
#undef X
        default: cfg -= kSplitBf16; break;   // not instantiated on the split path: its fp32 form
      }
    }
  }
  switch (cfg) {
#define X(id, TM, NWM, WK, PF) \
  case id: launch_dmm<Dmm<G, U8, FRAG_K, FRAG_MN, TM, VN, NWM, 1, WK, 1, EPI, false, PF>>(g, ksplit, ksplit, xcd, s); break;
    PAAC_FWD_CFGS(X)
#undef X
    default: break;
  }
  return ksplit;
}

// fc forward at large training batches on the LDS-tiled split-bf16 GEMM (gemm3.h): split-K slabs like launch_fwd's
// EPI_SLAB form.  Returns the number of slabs, 0 when the shape does not qualify (the caller takes the dmm path).
static int launch_fc_gemm3(const float* x, const float* wf, float* slab, int batch, int K, int H, hipStream_t s) {
  static const int min_rows = env_int("PAAC_GEMM3_MIN_ROWS", 513);
  if (batch < min_rows || (K % 32) != 0) return 0;
  Gemm3Args a;
  memset(&a, 0, sizeof(a));
  a.A = x; a.lda = K;
  a.B = wf; a.ldb = H;
  a.out = slab; a.ldo = H;
  a.M = batch; a.N = H; a.K = K;
  using D = Gemm3<true, false, EPI_SLAB>;
  a.MB = (batch + D::TM - 1) / D::TM;
  a.NB = (H + D::TN - 1) / D::TN;
  // K splits: enough workgroups for about one round of the 256 CUs (one workgroup per CU: 96 KB of LDS)
  const int stages = K / 32, tiles = a.MB * a.NB;
  int S = (256 + tiles / 2) / tiles;
  S = S < 1 ? 1 : (S > FC_SPLITS_MAX ? FC_SPLITS_MAX : S);
  a.stages_per_split = (stages + S - 1) / S;
  a.S = (stages + a.stages_per_split - 1) / a.stages_per_split;
  a.slab_stride = (long)batch * H;
  prof_mix(6);
  launch_k(gemm3_kernel<D>, dim3((unsigned)(a.MB * a.NB * a.S)), dim3(D::THREADS), s, PROF_WHOLE, a);
  return a.S;
}

// ---------------------------------------------------------------------------------------------
// Conv tower (tower.h): Nature conv1 -> conv2 -> conv3 in one launch.
size_t tower_pack_bytes() { return (size_t)kTowerPackAllVecs * sizeof(bf16x8); }      // (the two-conv tower's planes fit too)
// the two-conv tower is written for the stock NIPS geometry (16 / 32 filters, fc 256)
constexpr bool kTower2 = OtherNet::NCONV == 2 && OtherNet::C1 == 16 && OtherNet::C2 == 32 && OtherNet::FLAT == kT2Flat;
bool tower2_available() { return kTower2; }

int launch_pack_weights(paac_ctx* ctx, const float* params, hipStream_t s) {
  const paac_layout& L = ctx->layout;
  {   // fc weights in fragment order for the small-batch fc + heads kernel (fc_heads.h)
    const float* wf = params + L.offset[2 * ctx->spec.nconv];
    f32x4* out = reinterpret_cast<f32x4*>(ctx->fc_pack);
    if (ctx->cfg.arch == PAAC_ARCH_NATURE)
      launch_k(pack_fc_kernel<NatureNet::FLAT, NatureNet::H>, dim3((NatureNet::FLAT / 16) * (NatureNet::H / 16) * 64 / 256),
               dim3(256), s, PROF_NONE, wf, out);
    else
      launch_k(pack_fc_kernel<OtherNet::FLAT, OtherNet::H>, dim3((OtherNet::FLAT / 16) * (OtherNet::H / 16) * 64 / 256), dim3(256),
               s, PROF_NONE, wf, out);
  }
  if (ctx->tower2_on) {
    constexpr int threads2 = (8 * 1 + 8 * 2) * 64;
    launch_k(pack_tower2_kernel, dim3((threads2 + 255) / 256), dim3(256), s, PROF_NONE, params + L.offset[0], params + L.offset[2],
             reinterpret_cast<bf16x8*>(ctx->tower_pack));
    return 0;
  }
  if (!ctx->tower_on) return 0;
  constexpr int threads = (8 * 2 + 16 * 4 + 18 * 4) * 64;
  launch_k(pack_tower_kernel, dim3((threads + 255) / 256), dim3(256), s, PROF_NONE, params + L.offset[0], params + L.offset[2],
           params + L.offset[4], reinterpret_cast<bf16x8*>(ctx->tower_pack));
  return launch_pack_dgrad(ctx, params, s);
}

template <class G, bool KEEP>
static void launch_tower_variant(const TowerArgs& a, hipStream_t s) {
  launch_k(tower_kernel<G, KEEP>, dim3((unsigned)(a.batch * G::NR)), dim3(512), s, PROF_WHOLE, a);
}

// keep = also write the fp32 conv1 / conv2 activations (the backward pass and the debug read-back need them)
// packed3: conv3's output goes out in fc_heads_kernel's A-fragment order (acting rows that nothing else reads)
// keepW / keep_row: the rows are ALSO kept in another activation set (the training set) at row offset keep_row -- fp32 conv1
// / conv2 outputs and a plain-row copy of conv3's -- while the fragment-order conv3 output for the fc kernel goes to W
static void launch_tower(paac_ctx* ctx, Workspace& W, const float* params, const uint8_t* states, int batch, bool keep,
                         bool packed3, hipStream_t s, Workspace* keepW = nullptr, int keep_row = 0) {
  const paac_layout& L = ctx->layout;
  TowerArgs a;
  a.states = states;
  const bf16x8* pk = reinterpret_cast<const bf16x8*>(ctx->tower_pack);
  a.w1p = pk;
  a.w2p = pk + kTowerW1Vecs;
  a.w3p = pk + kTowerW1Vecs + kTowerW2Vecs;
  a.b1 = params + L.offset[1];
  a.b2 = params + L.offset[3];
  a.b3 = params + L.offset[5];
  a.act1 = W.act[0];
  a.act2 = W.act[1];
  a.act3 = W.act[2];
  a.batch = batch;
  a.act3_packed = packed3 ? 1 : 0;
  a.act3_rows = nullptr;
  if (keepW) {
    keep = true;
    a.act1 = keepW->act[0] + (size_t)keep_row * 400 * NatureNet::C1;
    a.act2 = keepW->act[1] + (size_t)keep_row * 81 * NatureNet::C2;
    if (packed3) a.act3_rows = keepW->act[2] + (size_t)keep_row * NatureNet::FLAT;      // a second copy beside the fragments
    else a.act3 = keepW->act[2] + (size_t)keep_row * NatureNet::FLAT;                   // plain rows: the kept ones serve the fc too
  }
#ifdef PAAC_DMM_STAMPS
  a.stamps = g_tower_stamps;
#endif
  // regions per sample: as many as keep the launch within ONE round of the 256 CUs (a workgroup takes a CU's LDS: workgroup
  // 257 waits for the first to finish -- tools/probe_regions.py: 68 rows as 4 regions = 272 workgroups 30.5 us per forward
  // against 25.3 as 2 regions, 136 rows as 2 regions 42.2 against 35.7 as 1)
  const int force = ctx->tune[OP_CONV_TOWER][batch_class(batch)].cfg;
  // (every workgroup streams all the weights, so more regions is more L2 traffic: at 32 rows 8 regions of 4x2 -- 256
  // workgroups -- take 11.0 us as a launch of their own against 10.7 for 4 of 4x4, but the replayed cycle is 2 us shorter
  // with them, 653 k against 648 k env-steps/s in two A/B pairs: the launch ramps and drains faster on all 256 CUs)
  int regions = (8 * batch <= 256) ? 8 : (4 * batch <= 256) ? 4 : (2 * batch <= 256) ? 2 : 1;
  if (force == 1 || force == 2 || force == 4 || force == 8) regions = force;
  if (regions == 8) {
    if (keep) launch_tower_variant<TowerGeom<4, 2>, true>(a, s);
    else launch_tower_variant<TowerGeom<4, 2>, false>(a, s);
  } else if (regions == 4) {
    if (keep) launch_tower_variant<TowerGeom<4, 4>, true>(a, s);
    else launch_tower_variant<TowerGeom<4, 4>, false>(a, s);
  } else if (regions == 2) {
    if (keep) launch_tower_variant<TowerGeom<4, 7>, true>(a, s);
    else launch_tower_variant<TowerGeom<4, 7>, false>(a, s);
  } else {
    if (keep) launch_tower_variant<TowerGeom<7, 7>, true>(a, s);
    else launch_tower_variant<TowerGeom<7, 7>, false>(a, s);
  }
}

template <class G, bool KEEP>
static void launch_tower2_variant(const Tower2Args& a, hipStream_t s) {
  launch_k(tower2_kernel<G, KEEP>, dim3((unsigned)(a.batch * G::NR)), dim3(512), s, PROF_WHOLE, a);
}

// The two-conv tower (tower2.h): same roles of keep / packed / keepW as launch_tower's.
static void launch_tower2(paac_ctx* ctx, Workspace& W, const float* params, const uint8_t* states, int batch, bool keep,
                          bool packed2, hipStream_t s, Workspace* keepW = nullptr, int keep_row = 0) {
  const paac_layout& L = ctx->layout;
  Tower2Args a;
  a.states = states;
  const bf16x8* pk = reinterpret_cast<const bf16x8*>(ctx->tower_pack);
  a.w1p = pk;
  a.w2p = pk + kT2W1Vecs;
  a.b1 = params + L.offset[1];
  a.b2 = params + L.offset[3];
  a.act1 = W.act[0];
  a.act2 = W.act[1];
  a.batch = batch;
  a.act2_packed = packed2 ? 1 : 0;
  a.act2_rows = nullptr;
  if (keepW) {
    keep = true;
    a.act1 = keepW->act[0] + (size_t)keep_row * 400 * 16;
    if (packed2) a.act2_rows = keepW->act[1] + (size_t)keep_row * kT2Flat;
    else a.act2 = keepW->act[1] + (size_t)keep_row * kT2Flat;
  }
  const int force = ctx->tune[OP_CONV_TOWER][batch_class(batch)].cfg;
  // regions per sample: nine 3x3 regions while that stays within one round of the 256 CUs (28 rows), four 5x5 up to 64 rows,
  // else one (tools/probe_regions.py: at 32 rows nine regions = 288 workgroups 17.0 us per forward against 14.2 with four)
  int regions = (9 * batch <= 256) ? 9 : (4 * batch <= 256) ? 4 : 1;
  if (force == 1 || force == 4 || force == 9) regions = force;
  if (regions == 9) {
    if (keep) launch_tower2_variant<Tower2Geom<3, 3>, true>(a, s);
    else launch_tower2_variant<Tower2Geom<3, 3>, false>(a, s);
  } else if (regions == 4) {
    if (keep) launch_tower2_variant<Tower2Geom<5, 5>, true>(a, s);
    else launch_tower2_variant<Tower2Geom<5, 5>, false>(a, s);
  } else {
    if (keep) launch_tower2_variant<Tower2Geom<9, 9>, true>(a, s);
    else launch_tower2_variant<Tower2Geom<9, 9>, false>(a, s);
  }
}

template <class NT>
static int forward_impl(paac_ctx* ctx, int wsi, const float* params, const uint8_t* states, int batch, float* logits,
                        float* probs, float* values, const PhiloxArgs& ph, const SynthStepArgs& st, hipStream_t s,
                        bool defer_heads = false, bool trunk_only = false) {
  const paac_layout& L = ctx->layout;
  Workspace& W = ctx->ws[wsi];
  ctx->last_ws = wsi;
  const int cls = batch_class(batch);
  const int A = ctx->cfg.num_actions;
  int t = 0;
  const float* w1 = params + L.offset[t++];
  const float* b1 = params + L.offset[t++];
  const float* w2 = params + L.offset[t++];
  const float* b2 = params + L.offset[t++];
  const float* w3 = nullptr;
  const float* b3 = nullptr;
  if constexpr (NT::NCONV == 3) {
    w3 = params + L.offset[t++];
    b3 = params + L.offset[t++];
  }
  const float* wf = params + L.offset[t++];
  const float* bf = params + L.offset[t++];
  const float* wa = params + L.offset[t++];
  const float* ba = params + L.offset[t++];
  const float* wc = params + L.offset[t++];
  const float* bc = params + L.offset[t++];

  bool tower = false;
  if constexpr (NT::NCONV == 3) tower = ctx->tower_on != 0;
  bool tower2 = false;                       // the two-conv network's tower (tower2.h)
  if constexpr (NT::NCONV == 2 && kTower2) tower2 = ctx->tower2_on != 0;
  // paac_keep_next_forward (one shot): this acting forward's rows are also kept in the training activation set
  int keep_row = -1;
  bool keep_h_only = false;
  if (wsi == 0) {
    keep_row = ctx->keep_row;
    keep_h_only = ctx->keep_h_only != 0;
    ctx->keep_row = -1;
    ctx->keep_h_only = 0;
  }
  const bool keep_acts = wsi == 1 || !ctx->managed_weights;          // fp32 conv activations / h kept for backward and read-back
  constexpr int NW = fc_heads_waves(NT::FLAT);      // 0: this fc geometry has no fc + head partials kernel
  const bool small_tail = NW > 0 && batch <= kFcHeadsMaxRows && !ph.enabled && !st.enabled;   // fc + head partials kernel (fc_heads.h)
  // conv3 -> fc hand-off in the fc kernel's fragment order: rows are padded to 16 inside the activation buffer (max_batch
  // is rounded up at allocation)
  // ... and acting batches of up to 256 rows (the 128- / 256-environment shards) take the same fc kernel, finished by a
  // few workgroups (heads_finish_rows_kernel) instead of split-K slabs + a per-row heads launch
  const bool mid_tail = !small_tail && (tower || tower2) && !keep_acts && wsi == 0 && batch <= kFcHeadsMidRows && !ph.enabled &&
                        !st.enabled && !trunk_only;
  Workspace* keepW = nullptr;
  if (keep_row >= 0) {
    // (the fc + head partials routes, or -- the counter-based sampler's forwards -- the split-K fc with its per-row heads
    // launch, which then leaves the finished fc activations in the training set instead of the acting one)
    const bool generic_ok = !small_tail && !mid_tail && (ph.enabled || st.enabled) && !defer_heads && !trunk_only;
    if (!((tower || tower2) && !keep_acts && (small_tail || mid_tail || generic_ok) && keep_row + batch <= ctx->max_batch)) {
      set_error("paac_keep_next_forward: rows [%d, %d) cannot be kept -- needs a stock trunk's conv tower, managed weights, an "
                "acting forward of at most %d rows (any size with the counter-based sampler) and keep_row + batch <= max_batch "
                "(%d)", keep_row, keep_row + batch, kFcHeadsMidRows, ctx->max_batch);
      return -1;
    }
    keepW = &ctx->ws[1];
  }
  // (measured at 128 rows: fragment-order hand-off tower 16.5 + fc 8.7 us, plain rows 15.2 + 10.4)
  const bool packed3 = (tower || tower2) && (small_tail || mid_tail) && !keep_acts;
  if (!ctx->managed_weights && (tower || tower2 || batch <= kFcHeadsMaxRows)) launch_pack_weights(ctx, params, s);
  if (tower) {
    ProfScope ps(ctx, F_CONV_TOWER, batch, s);
    prof_mix(3);       // conv1: u8 pixels x weights split into 3 bf16 terms
    prof_mix(6);       // conv2, conv3: six-product split-bf16
    launch_tower(ctx, W, params, states, batch, keep_acts, packed3, s, keep_h_only ? nullptr : keepW, keep_row < 0 ? 0 : keep_row);
  }
  if (tower2) {
    ProfScope ps(ctx, F_CONV_TOWER, batch, s);
    prof_mix(3);       // conv1: u8 pixels x weights split into 3 bf16 terms
    prof_mix(6);       // conv2: six-product split-bf16
    launch_tower2(ctx, W, params, states, batch, keep_acts, packed3, s, keep_h_only ? nullptr : keepW, keep_row < 0 ? 0 : keep_row);
  }
  if (!tower && !tower2) {
    ProfScope ps(ctx, F_CONV1_FWD, batch, s);
    constexpr int P1 = NT::G1::OPIX, F1 = NT::G1::FEATS;
    GemmArgs g = make_args(states, (size_t)batch * 28224, w1, (size_t)F1 * NT::C1 * 4, W.act[0], b1, batch * P1, NT::C1, F1, NT::C1, NT::C1);
    launch_fwd<typename NT::G1, true, NT::C1, EPI_BIAS_RELU>(g, ctx->tune[OP_CONV1_FWD][cls], s);
  }
  if (!tower && !tower2) {
    ProfScope ps(ctx, F_CONV2_FWD, batch, s);
    constexpr int P1 = NT::G1::OPIX, P2 = NT::G2::OPIX, F2 = NT::G2::FEATS;
    GemmArgs g = make_args(W.act[0], (size_t)batch * P1 * NT::C1 * 4, w2, (size_t)F2 * NT::C2 * 4, W.act[1], b2, batch * P2, NT::C2, F2, NT::C2, NT::C2);
    launch_fwd<typename NT::G2, false, NT::C2, EPI_BIAS_RELU>(g, ctx->tune[OP_CONV2_FWD][cls], s);
  }
  const float* last = W.act[1];
  if constexpr (NT::NCONV == 3) {
    last = W.act[2];
  }
  if (keepW && !keep_h_only && !packed3)      // the towers wrote their plain-row output into the training set
    last = keepW->act[NT::NCONV - 1] + (size_t)keep_row * NT::FLAT;
  if constexpr (NT::NCONV == 3) if (!tower) {
    ProfScope ps(ctx, F_CONV3_FWD, batch, s);
    constexpr int P2 = NT::G2::OPIX, P3 = NT::G3::OPIX, F3 = NT::G3::FEATS;
    GemmArgs g = make_args(W.act[1], (size_t)batch * P2 * NT::C2 * 4, w3, (size_t)F3 * NT::C3 * 4, W.act[2], b3, batch * P3, NT::C3, F3, NT::C3, NT::C3);
    launch_fwd<typename NT::G3, false, NT::C3, EPI_BIAS_RELU>(g, ctx->tune[OP_CONV3_FWD][cls], s);
    last = W.act[2];
  }
  // Small acting / evaluation batches: fc with the head contractions folded into its epilogue (fc_heads.h), then the
  // head finish -- as its own one-workgroup launch here, or (defer_heads) inside the caller's sampler launch.
  if constexpr (NW > 0) if (small_tail || mid_tail) {
    if (wsi == 1) ctx->heads_pending_rows = 0;
    // quarter tiles (fc_heads.h) while they fit one workgroup per CU: 32 rows of the stock fc width = 256 workgroups
    const bool quarter = fc_heads_quarter_ok(NT::FLAT) && ((batch + 7) / 8) * (NT::H / 8) <= 256 && !ctx->no_quarter_tiles &&
                         ctx->ahead_state == nullptr;
    const int NTILES = quarter ? NT::H / 8 : NT::H / 16;
    ctx->heads_ntiles = NTILES;
    float* partial = W.fc_slab;      // [NTILES][batch][A + 1]: fits the split-K slab buffer
    const bool keep_h = keep_acts;
    // kept rows: the finished fc activations (bias + ReLU applied) go where the training forward's fc slabs would be
    float* h_keep = keepW ? keepW->fc_slab + (size_t)keep_row * NT::H : nullptr;
    {
      ProfScope ps(ctx, F_FC_FWD, batch, s);
      prof_mix(1);     // fp32 MFMA
      // one-shot (paac_act_step_mt): a spare workgroup makes the next sampling step's MT19937 doubles meanwhile (mt_ahead.h)
      MtAheadArgs ah{ctx->ahead_state, ctx->ahead_state ? reinterpret_cast<MtAhead*>(ctx->mt_ahead) : nullptr, ctx->ahead_D};
      ctx->ahead_state = nullptr;
      const unsigned grid = NTILES * ((batch + 15) / 16) + (ah.out ? 1 : 0);
      if constexpr (fc_heads_quarter_ok(NT::FLAT)) if (quarter) {
        if (packed3)
          launch_k(fc_heads_q_kernel<NT::FLAT, NT::H, NW, true>, dim3(NTILES * ((batch + 7) / 8)), dim3(64 * NW), s, PROF_WHOLE,
                   last, reinterpret_cast<const f32x4*>(ctx->fc_pack), bf, wa, wc, A, batch, partial, h_keep);
        else
          launch_k(fc_heads_q_kernel<NT::FLAT, NT::H, NW, false>, dim3(NTILES * ((batch + 7) / 8)), dim3(64 * NW), s, PROF_WHOLE,
                   last, reinterpret_cast<const f32x4*>(ctx->fc_pack), bf, wa, wc, A, batch, partial,
                   keep_h ? W.h : (float*)nullptr);
      }
      if (quarter) {
      } else if (packed3)
        launch_k(fc_heads_kernel<NT::FLAT, NT::H, NW, true>, dim3(grid), dim3(64 * NW), s, PROF_WHOLE,
                 last, reinterpret_cast<const f32x4*>(ctx->fc_pack), bf, wa, wc, A, batch, partial, h_keep, ah);
      else
        launch_k(fc_heads_kernel<NT::FLAT, NT::H, NW, false>, dim3(grid), dim3(64 * NW), s, PROF_WHOLE,
                 last, reinterpret_cast<const f32x4*>(ctx->fc_pack), bf, wa, wc, A, batch, partial,
                 keep_h ? W.h : (float*)nullptr, ah);
    }
    if (defer_heads) return 0;       // the caller's launch finishes the heads (or, for bootstrap rows, the backward's)
    if (mid_tail) {
      ProfScope ps(ctx, F_HEADS_FWD, batch, s);
      const int rpw = 256 / (A + 1);
      launch_k(heads_finish_rows_kernel, dim3((batch + rpw - 1) / rpw), dim3(256), s, PROF_WHOLE, (const float*)partial, NTILES,
               batch, A, rpw, ba, bc, W.logits, W.probs, W.values, logits, probs, values);
    } else {
      ProfScope ps(ctx, F_HEADS_FWD, batch, s);
      launch_k(heads_finish_kernel, dim3(1), dim3(256), s, PROF_WHOLE, (const float*)partial, NTILES, batch, A, ba, bc,
               W.logits, W.probs, W.values, logits, probs, values);
    }
    return 0;
  }
  if (defer_heads) {
    set_error("forward: deferred heads need batch <= %d", kFcHeadsMaxRows);
    return -1;
  }
  int splits = 1;
  {
    ProfScope ps(ctx, F_FC_FWD, batch, s);
    GemmArgs g = make_args(last, (size_t)batch * NT::FLAT * 4, wf, (size_t)NT::FLAT * NT::H * 4, W.fc_slab, nullptr, batch, NT::H, NT::FLAT, NT::H, NT::H);
    g.slab_rows = batch;
    splits = launch_fc_gemm3(last, wf, W.fc_slab, batch, NT::FLAT, NT::H, s);
    if (splits == 0) splits = launch_fwd<typename NT::GFC, false, NT::H, EPI_SLAB>(g, ctx->tune[OP_FC_FWD][cls], s);
  }
  if (wsi == 1) ctx->heads_pending_rows = 0;
  if (trunk_only && wsi == 1) {      // the backward's first launch finishes the heads (heads.h: heads_train_kernel)
    ctx->heads_pending_rows = batch;
    ctx->heads_pending_splits = splits;
    ctx->heads_pending_h = 0;
    return 0;
  }
  {
    ProfScope ps(ctx, F_HEADS_FWD, batch, s);
    SynthStepArgs stl = st;
    if (stl.enabled) stl.stack_in = reinterpret_cast<const uint32_t*>(states);   // the stacks just observed
    launch_heads_fwd<NT::H>(A, dim3(stl.enabled ? batch + batch * PRE_BANDS : batch), s, (const float*)W.fc_slab, splits,
                            (long)batch * NT::H, bf, wa, ba, wc, bc, A, keepW ? keepW->fc_slab + (size_t)keep_row * NT::H : W.h,
                            W.logits, W.probs, W.values, logits, probs, values, ph, batch, stl);
  }
  return 0;
}

int launch_forward(paac_ctx* ctx, int ws, const float* params, const uint8_t* states, int batch, float* logits,
                   float* probs, float* values, hipStream_t s) {
  PhiloxArgs ph;
  memset(&ph, 0, sizeof(ph));
  SynthStepArgs st;
  memset(&st, 0, sizeof(st));
  if (ctx->cfg.arch == PAAC_ARCH_NATURE)
    return forward_impl<NatureNet>(ctx, ws, params, states, batch, logits, probs, values, ph, st, s);
  return forward_impl<OtherNet>(ctx, ws, params, states, batch, logits, probs, values, ph, st, s);
}

// Training forward up to the fc layer's split-K slabs (batches above the small-batch tail only; smaller ones run the
// whole forward): the heads are finished by the backward's first launch, or by launch_deferred_heads.
int launch_forward_trunk_train(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, hipStream_t s) {
  PhiloxArgs ph;
  memset(&ph, 0, sizeof(ph));
  SynthStepArgs st;
  memset(&st, 0, sizeof(st));
  if (ctx->cfg.arch == PAAC_ARCH_NATURE)
    return forward_impl<NatureNet>(ctx, 1, params, states, batch, nullptr, nullptr, nullptr, ph, st, s, false, true);
  return forward_impl<OtherNet>(ctx, 1, params, states, batch, nullptr, nullptr, nullptr, ph, st, s, false, true);
}

// The heads launch a trunk-only training forward left out (a backward that cannot fuse it runs it first).
int launch_deferred_heads(paac_ctx* ctx, const float* params, hipStream_t s) {
  const int rows = ctx->heads_pending_rows;
  if (rows <= 0) return 0;
  ctx->heads_pending_rows = 0;
  const paac_layout& L = ctx->layout;
  const int nt = L.num_tensors;
  Workspace& W = ctx->ws[1];
  PhiloxArgs ph;
  memset(&ph, 0, sizeof(ph));
  SynthStepArgs st;
  memset(&st, 0, sizeof(st));
  const int A = ctx->cfg.num_actions, H = ctx->spec.fc;
  // (rows kept by acting forwards hold finished fc activations: a zero bias and one "slab" reproduce them exactly)
  const float *bf = ctx->heads_pending_h ? ctx->zeros : params + L.offset[nt - 5], *wa = params + L.offset[nt - 4],
              *ba = params + L.offset[nt - 3], *wc = params + L.offset[nt - 2], *bc = params + L.offset[nt - 1];
  ctx->heads_pending_h = 0;
  ProfScope ps(ctx, F_HEADS_FWD, rows, s);
  if (ctx->cfg.arch == PAAC_ARCH_NATURE)
    launch_heads_fwd<NatureNet::H>(A, dim3(rows), s, (const float*)W.fc_slab, ctx->heads_pending_splits, (long)rows * H, bf, wa,
                                   ba, wc, bc, A, W.h, W.logits, W.probs, W.values, (float*)nullptr, (float*)nullptr,
                                   (float*)nullptr, ph, rows, st);
  else
    launch_heads_fwd<OtherNet::H>(A, dim3(rows), s, (const float*)W.fc_slab, ctx->heads_pending_splits, (long)rows * H, bf, wa,
                                  ba, wc, bc, A, W.h, W.logits, W.probs, W.values, (float*)nullptr, (float*)nullptr,
                                  (float*)nullptr, ph, rows, st);
  return 0;
}

// Acting trunk up to the per-tile head partials (fc_heads.h); the caller's sampler launch finishes the heads.
int launch_forward_trunk(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, const float** partial,
                         int* ntiles, const float** ba, const float** bc, hipStream_t s) {
  PhiloxArgs ph;
  memset(&ph, 0, sizeof(ph));
  SynthStepArgs st;
  memset(&st, 0, sizeof(st));
  const paac_layout& L = ctx->layout;
  const int nt = L.num_tensors;
  *ba = params + L.offset[nt - 3];
  *bc = params + L.offset[nt - 1];
  *partial = ctx->ws[0].fc_slab;
  const int rc = (ctx->cfg.arch == PAAC_ARCH_NATURE)
                     ? forward_impl<NatureNet>(ctx, 0, params, states, batch, nullptr, nullptr, nullptr, ph, st, s, true)
                     : forward_impl<OtherNet>(ctx, 0, params, states, batch, nullptr, nullptr, nullptr, ph, st, s, true);
  *ntiles = ctx->heads_ntiles;       // (this launch's: whole or quarter tiles, fc_heads.h)
  return rc;
}

// Acting-shaped forward of the N bootstrap observations whose rows complete a training set the acting steps have kept
// (paac_keep_next_forward): conv tower + fc, rows kept at [train_row, train_row + batch); no heads -- the backward's first
// launch takes the bootstrap values from the kept fc activations like it does after a trunk-only training forward.
int launch_bootstrap_trunk(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, int train_row, hipStream_t s) {
  PhiloxArgs ph;
  memset(&ph, 0, sizeof(ph));
  SynthStepArgs st;
  memset(&st, 0, sizeof(st));
  ctx->keep_row = train_row;
  ctx->keep_h_only = 1;          // the backward covers the rollout rows only: of the bootstrap rows it reads the fc activations
  int rc;
  if (ctx->cfg.arch == PAAC_ARCH_NATURE)
    rc = forward_impl<NatureNet>(ctx, 0, params, states, batch, nullptr, nullptr, nullptr, ph, st, s, true);
  else
    rc = forward_impl<OtherNet>(ctx, 0, params, states, batch, nullptr, nullptr, nullptr, ph, st, s, true);
  ctx->keep_row = -1;
  ctx->keep_h_only = 0;
  if (rc) return rc;
  ctx->heads_pending_rows = train_row + batch;
  ctx->heads_pending_splits = 1;
  ctx->heads_pending_h = 1;
  return 0;
}

int launch_forward_sample(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, float* probs,
                          float* values, uint64_t seed, const uint64_t* step_base, uint64_t step_off,
                          uint32_t env_offset, int32_t* actions, const SynthStepArgs* step, hipStream_t s) {
  SynthStepArgs st;
  memset(&st, 0, sizeof(st));
  if (step) st = *step;
  PhiloxArgs ph;
  ph.enabled = 1;
  ph.seed = seed;
  ph.step_base = step_base;
  ph.step_off = step_off;
  ph.env_offset = env_offset;
  ph.actions = actions;
  if (ctx->cfg.arch == PAAC_ARCH_NATURE)
    return forward_impl<NatureNet>(ctx, 0, params, states, batch, nullptr, probs, values, ph, st, s);
  return forward_impl<OtherNet>(ctx, 0, params, states, batch, nullptr, probs, values, ph, st, s);
}

int launch_forward_sample_step(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, float* probs,
                               float* values, uint64_t seed, const uint64_t* step_base, uint64_t step_off,
                               uint32_t env_offset, int32_t* actions, uint64_t env_seed, uint32_t thresh,
                               uint8_t* stack_out, float* rewards, float* masks, float* ep_reward, int32_t* ep_len,
                               void* finished, hipStream_t s) {
  SynthStepArgs st;
  memset(&st, 0, sizeof(st));
  st.enabled = 1;
  st.seed = env_seed;
  st.thresh = thresh;
  st.stack_out = reinterpret_cast<uint32_t*>(stack_out);
  st.rewards = rewards;
  st.masks = masks;
  st.ep_reward = ep_reward;
  st.ep_len = ep_len;
  st.fin = reinterpret_cast<FinishedRing*>(finished);
  return launch_forward_sample(ctx, params, states, batch, probs, values, seed, step_base, step_off, env_offset, actions, &st,
                               s);
}

#ifdef PAAC_DMM_STAMPS
extern "C" void paac_debug_set_heads_stamps(unsigned long long* p) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_dev), &p, sizeof(p));
}
#endif

int fc_splits_max() { return FC_SPLITS_MAX; }

}  // namespace paac
