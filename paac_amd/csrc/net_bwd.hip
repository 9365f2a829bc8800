// Backward half of the actor-critic network (see net_fwd.hip): dgrad / wgrad launches, slab finalize, launch tuning.
#include "net_common.h"
#include "dgrad_tower.h"
#include "gemm3.h"

namespace paac {

// dgrad: A = FRAG_K patches of dY, B = FRAG_K taps of W^T.
template <int NDIM>
struct DgradN {
  static constexpr int TN = (NDIM % 64 == 0) ? 4 : (NDIM % 32 == 0) ? 2 : 1;
};
template <class G, int NDIM, int BCO, int EPI, int TM, int NWM, int WK, int PF, int XB = 0>
using DgradBody = Dmm<G, false, FRAG_K, FRAG_K, TM, DgradN<NDIM>::TN, NWM, 1, WK, BCO, EPI, false, PF, XB>;

template <int NDIM, int EPI>
static void resolve_dgrad(const GemmArgs& g, int zdim, Tune t, int& cfg, int& xcd) {
  constexpr int TN = DgradN<NDIM>::TN;
  cfg = t.cfg;
  xcd = t.xcd;
  if (cfg < 0) {
    const long tiles = (long)((g.M + 31) / 32) * ((g.N + 16 * TN - 1) / (16 * TN)) * zdim;
    cfg = (tiles <= 384) ? 1 : (tiles <= 1536) ? 2 : 3;
    xcd = (EPI == EPI_MASK_PARITY) ? 0 : -1;
  }
}

template <class G, int NDIM, int BCO, int EPI>
static void launch_dgrad(const GemmArgs& g, int zdim, Tune t, hipStream_t s) {
  int cfg, xcd;
  resolve_dgrad<NDIM, EPI>(g, zdim, t, cfg, xcd);
  if (cfg >= kSplitBf16) {
    switch (cfg - kSplitBf16) {
#define X(id, TM, NWM, WK, PF) \
  case id: launch_dmm<DgradBody<G, NDIM, BCO, EPI, TM, NWM, WK, split_pf(PF), 2>>(g, zdim, 1, xcd, s); return;
      PAAC_DGRAD_SPLIT_CFGS(X)
#undef X
      default: cfg -= kSplitBf16; break;
    }
  }
  switch (cfg) {
#define X(id, TM, NWM, WK, PF) \
  case id: launch_dmm<DgradBody<G, NDIM, BCO, EPI, TM, NWM, WK, PF>>(g, zdim, 1, xcd, s); break;
    PAAC_DGRAD_CFGS(X)
#undef X
    default: break;
  }
}

// wgrad: A = FRAG_MN patches^T (16*TM features per wave), B = FRAG_MN dY; split-K slabs + bias-gradient row.
template <int NDIM>
struct WgradN {
  static constexpr int VN = (NDIM % 64 == 0) ? 4 : (NDIM % 32 == 0) ? 2 : 1;
};
template <class G, bool U8, int NDIM, int TM, int WK, int PF, int XB = 0>
using WgradBody = Dmm<G, U8, FRAG_MN, FRAG_MN, TM, WgradN<NDIM>::VN, 1, 1, WK, 1, EPI_SLAB, true, PF, XB>;

template <bool U8, int NDIM>
static void resolve_wgrad(const GemmArgs& g, int max_split, Tune t, int& cfg, int& ks, int& xcd) {
  constexpr int VN = WgradN<NDIM>::VN;
  const int ngroups = (g.K + 15) / 16;
  cfg = t.cfg;
  ks = t.ksplit;
  xcd = t.xcd;
  if (cfg < 0) {
    const long tiles = (long)((g.M + 63) / 64) * ((g.N + 16 * VN - 1) / (16 * VN));
    if (ngroups >= 64 && max_split >= 8) {
      cfg = 0;
      ks = (pick_ksplit(tiles, 4, ngroups, max_split) + 7) / 8 * 8;   // multiple of 8: one K range per XCD
      xcd = 2;
    } else {
      cfg = 1;
      ks = 1;
      xcd = -1;
    }
    if (U8) cfg += kExactBf16;
  }
  if (ks < 1) ks = 1;
  if (ks > max_split) ks = max_split;
  const int plain = cfg % kExactBf16;
  if (U8 && (plain == 4 || plain == 5)) cfg -= plain;   // u8 patches are loaded as uchar4: 64 features per wave only
  if (U8 && cfg >= kSplitBf16) cfg = kExactBf16 + plain;              // u8 operand: exact path, not the split one
  if (!U8 && cfg >= kExactBf16 && cfg < kSplitBf16) cfg = plain;
}

template <class G, bool U8, int NDIM>
static int launch_wgrad(const GemmArgs& g, int max_split, Tune t, hipStream_t s) {
  int cfg, ks, xcd;
  resolve_wgrad<U8, NDIM>(g, max_split, t, cfg, ks, xcd);
  if constexpr (U8) {
    if (cfg >= kExactBf16) {
      switch (cfg - kExactBf16) {
#define X(id, TM, WK, PF)                                                                         \
  case id:                                                                                        \
    if constexpr (TM == 4) launch_dmm<WgradBody<G, U8, NDIM, TM, WK, PF, 1>>(g, ks, ks, xcd, s); \
    break;
        PAAC_WGRAD_CFGS(X)
#undef X
        default: break;
      }
      return ks;
    }
  }
  if constexpr (!U8) {
    if (cfg >= kSplitBf16) {
      switch (cfg - kSplitBf16) {
#define X(id, TM, WK, PF) \
  case id: launch_dmm<WgradBody<G, U8, NDIM, TM, WK, split_pf(PF), 2>>(g, ks, ks, xcd, s); return ks;
        PAAC_WGRAD_SPLIT_CFGS(X)
#undef X
        default: cfg -= kSplitBf16; break;
      }
    }
  }
  switch (cfg) {
#define X(id, TM, WK, PF)                                                                   \
  case id:                                                                                  \
    if constexpr (!U8 || TM == 4) launch_dmm<WgradBody<G, U8, NDIM, TM, WK, PF>>(g, ks, ks, xcd, s); \
    break;
    PAAC_WGRAD_CFGS(X)
#undef X
    default: break;
  }
  return ks;
}

// ---------------------------------------------------------------------------------------------
// Split-K slab reduction into the flat gradient (deterministic: fixed summation order).
// FIN_LANES adjacent lanes share one output float4: lane r sums slabs r, r + FIN_LANES, ... (up to 8 loads, all in
// flight at once: one memory round trip for up to 64 slabs), then the lanes' partial sums are combined by a fixed
// xor tree -- the order of the additions never changes from run to run.
constexpr int FIN_LANES = 8;
__global__ __launch_bounds__(256) void grad_finalize_kernel(const FinalizeArgs a) {
  const FinalizeSeg sg = a.seg[blockIdx.y];
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int r = gid % FIN_LANES;
  const int i = (gid / FIN_LANES) * 4;
  const bool live = i < sg.count;     // whole lane groups are live or not: the shuffles below stay inside a group
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int s0 = r; s0 < sg.splits; s0 += 8 * FIN_LANES) {
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int sp = s0 + u * FIN_LANES;
      t[u] = (live && sp < sg.splits) ? *reinterpret_cast<const f32x4*>(sg.src + (long)sp * sg.stride + i)
                                      : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
#pragma unroll
  for (int off = 1; off < FIN_LANES; off <<= 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] += __shfl_xor(v[c], off, 64);
  }
  if (live && r == 0) *reinterpret_cast<f32x4*>(sg.dst + i) = v;
}

// Data gradient of ONE VALID convolution of any kernel size / stride (a user architecture outside the reference trunks' layer
// shapes, networks.py:117-120): dX[b, iy, ix, c] = relu'(X) * sum over the outputs (oy, ox) the pixel fed -- iy = oy S + ky,
// ix = ox S + kx -- and their channels of dY[b, oy, ox, co] * W[ky, kx, c, co].  Plain fp32 FMAs in a fixed order, one thread
// per (pixel, input channel), 16-byte loads along the output channels: the correctness path for arbitrary geometry, not a
// tuned one (the family's layers keep their MFMA forms and the fused data-gradient tower).
template <class G, int COUT>
__global__ __launch_bounds__(256) void conv_dgrad_direct_kernel(const float* __restrict__ dY, const float* __restrict__ Wt,
                                                                const float* __restrict__ X, float* __restrict__ dX,
                                                                const int batch) {
  static_assert(COUT % 4 == 0, "output channels are read four at a time");
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)batch * G::IH * G::IW * G::C;
  if (gid >= total) return;
  const int c = (int)(gid % G::C);
  long pix = gid / G::C;
  const int ix = (int)(pix % G::IW);
  pix /= G::IW;
  const int iy = (int)(pix % G::IH);
  const int b = (int)(pix / G::IH);
  float acc = 0.f;
  if (X[gid] > 0.f) {
    for (int ky = iy % G::S; ky < G::KH && ky <= iy; ky += G::S) {
      const int oy = (iy - ky) / G::S;
      if (oy >= G::OH) continue;
      for (int kx = ix % G::S; kx < G::KW && kx <= ix; kx += G::S) {
        const int ox = (ix - kx) / G::S;
        if (ox >= G::OW) continue;
        const f32x4* dy = reinterpret_cast<const f32x4*>(dY + ((size_t)(b * G::OH + oy) * G::OW + ox) * COUT);
        const f32x4* w = reinterpret_cast<const f32x4*>(Wt + ((size_t)(ky * G::KW + kx) * G::C + c) * COUT);
#pragma unroll 4
        for (int q = 0; q < COUT / 4; ++q) {
          const f32x4 a = dy[q], v = w[q];
          acc = fmaf(a[0], v[0], acc);
          acc = fmaf(a[1], v[1], acc);
          acc = fmaf(a[2], v[2], acc);
          acc = fmaf(a[3], v[3], acc);
        }
      }
    }
  }
  dX[gid] = acc;
}
template <class G, int COUT>
static void launch_dgrad_direct(const float* dY, const float* Wt, const float* X, float* dX, int batch, hipStream_t s) {
  const long total = (long)batch * G::IH * G::IW * G::C;
  launch_k(conv_dgrad_direct_kernel<G, COUT>, dim3((unsigned)((total + 255) / 256)), dim3(256), s, PROF_WHOLE, dY, Wt, X, dX, batch);
}

int launch_pack_dgrad(paac_ctx* ctx, const float* params, hipStream_t s) {
  if (!ctx->tower_on) return 0;
  const paac_layout& L = ctx->layout;
  bf16x8* base = reinterpret_cast<bf16x8*>(ctx->tower_pack) + kTowerPackVecs;
  constexpr int threads = 18 * 4 * 64 + 4 * 8 * 2 * 64;
  launch_k(pack_dgrad_kernel, dim3((threads + 255) / 256), dim3(256), s, PROF_NONE, params + L.offset[2], params + L.offset[4],
           base, base + kDgradW3Vecs);
  return 0;
}

// phase: 0 = whole backward; 1 = heads + fc (gradients of fc_w .. critic_b, the contiguous tail of the flat
// buffer, 95 % of its bytes); 2 = conv layers (the head of the flat buffer) + slab finalize.  The split lets a
// data-parallel caller all-reduce the tail while phase 2 still computes.  3 = whole backward with the slab finalize
// left to the norm pass of the next paac_clip_rmsprop on `grad` (one launch less; the same sums in the same order).
template <class NT>
static int backward_impl(paac_ctx* ctx, const float* params, const uint8_t* states, const int32_t* actions,
                         const float* y, const float* adv, int batch, float beta, float* grad, float* loss_out,
                         int phase, const ReturnsArgs& rt, hipStream_t s) {
  const paac_layout& L = ctx->layout;
  Workspace& W = ctx->ws[1];
  const int cls = batch_class(batch);
  const int A = ctx->cfg.num_actions;
  const int i_w1 = 0, i_w2 = 2, i_w3 = 4;
  const int i_wf = (NT::NCONV == 3) ? 6 : 4;
  const int i_wa = i_wf + 2, i_wc = i_wf + 4;
  const float* wa = params + L.offset[i_wa];
  const float* wc = params + L.offset[i_wc];
  const float* wf = params + L.offset[i_wf];
  const float* w2 = params + L.offset[i_w2];
  const float* w3 = (NT::NCONV == 3) ? params + L.offset[i_w3] : nullptr;

  const bool do_fc = phase == 0 || phase == 1 || phase == 3, do_conv = phase == 0 || phase == 2 || phase == 3;
  // A trunk-only training forward is waiting for its heads: the whole backward of the three-conv network finishes them
  // inside the heads-gradient launch (heads_train_kernel; the reductions over rows ride in the dgrad tower launch);
  // every other route runs the heads launch that was left out first.
  bool fused_heads = false;
  if (ctx->heads_pending_rows > 0) {
    const bool boot_ok = !rt.boot_in_fwd || ctx->heads_pending_rows >= batch + rt.N;
    if (NT::NCONV == 3 && ctx->tower_on && (phase == 0 || phase == 3) && ctx->heads_pending_rows >= batch && boot_ok) {
      fused_heads = true;
    } else {
      const int rc = launch_deferred_heads(ctx, params, s);
      if (rc) return rc;
    }
  }
  ReturnsArgs rtl = rt;
  if (rtl.boot_in_fwd && !fused_heads) {       // the separate heads launch has produced the bootstrap rows' values
    rtl.v_boot = W.values + batch;
    rtl.boot_in_fwd = 0;
  }
  // (1) heads: dH, head weight/bias grads, loss scalars
  if (do_fc && fused_heads) {
    ProfScope ps(ctx, F_HEADS_BWD, batch, s);
    const int rows = ctx->heads_pending_rows;
    // (rows kept by the acting forwards hold finished fc activations: one "slab", zero bias -- fmaxf(h + 0, 0) == h)
    launch_heads_train<NT::H>(A, dim3(batch), s, (const float*)W.fc_slab, ctx->heads_pending_splits, (long)rows * NT::H,
                              ctx->heads_pending_h ? (const float*)ctx->zeros : params + L.offset[i_wf + 1], wa, params + L.offset[i_wa + 1], wc, params + L.offset[i_wc + 1], A,
                              batch, W.h, W.logits, W.probs, W.values, actions, y, adv, beta, ctx->dh, ctx->dl_buf, rtl);
    ctx->heads_pending_rows = 0;
    ctx->heads_pending_h = 0;
  } else if (do_fc) {
    ProfScope ps(ctx, F_HEADS_BWD, batch, s);
    launch_heads_bwd<NT::H>(A, dim3(batch + NT::H / 32 + 1), s, (const float*)W.probs, (const float*)W.values, actions, y,
                            adv, (const float*)W.h, wa, wc, A, batch, beta, ctx->dh, grad + L.offset[i_wa],
                            grad + L.offset[i_wa + 1], grad + L.offset[i_wc], grad + L.offset[i_wc + 1], loss_out, rtl);
  }
  const float* xf = (NT::NCONV == 3) ? W.act[2] : W.act[1];   // flattened last conv output
  float* dxf = (NT::NCONV == 3) ? ctx->dact[2] : ctx->dact[1];
  FinalizeArgs fin;
  memset(&fin, 0, sizeof(fin));
  float* slab = ctx->wslab;
  int conv1_splits_done = 0;      // > 0: conv1's weight gradient already ran, paired with conv2's
  bool fc_wgrad_held = false;
  GemmArgs held_gw;
  int held_ks = 1, held_xcd = -1;
  memset(&held_gw, 0, sizeof(held_gw));
  // conv wgrads write split-K slabs (+ bias row) into the workspace; grad_finalize_kernel sums them into the
  // flat gradient in a fixed order (deterministic, unlike float atomics).
  auto wgrad_out = [&](int i_w, int feats, int cout, float* base, int splits) {
    const long stride = (long)(feats + 1) * cout;
    fin.seg[fin.nseg++] = FinalizeSeg{base, grad + L.offset[i_w], feats * cout, splits, stride};
    fin.seg[fin.nseg++] = FinalizeSeg{base + (long)feats * cout, grad + L.offset[i_w + 1], cout, splits, stride};
  };
  // (A layer's weight and data gradient sharing one launch -- both wait only on dY -- was measured: the paired
  // kernel takes the SUM of the two times, their main loops already keep the MFMA pipes of the CUs they occupy busy.)
  // (2) fc wgrad (+ bias row): [FLAT+1][H] = fc_w then fc_b;  (3) fc dgrad, masked by relu'(last conv output)
  if (do_fc) {
    GemmArgs gw = make_args(xf, (size_t)batch * NT::FLAT * 4, ctx->dh, (size_t)batch * NT::H * 4, grad + L.offset[i_wf], nullptr, NT::FLAT, NT::H, batch, NT::H, NT::H);
    gw.slab_rows = NT::FLAT + 1;
    GemmArgs gd = make_args(ctx->dh, (size_t)batch * NT::H * 4, wf, (size_t)NT::FLAT * NT::H * 4, dxf, xf, batch, NT::FLAT, NT::H, 0, NT::FLAT);
    // Whole backward of the three-conv network: the fc weight gradient is held back to share a launch with conv3's (both
    // in their 2-wave fp32 configurations: 112 registers, 33 KB of LDS -- four workgroups per CU, so all 448 + 432 of them
    // are resident at once; each alone would prefer another configuration, the tuning table holds the pair-friendly ones
    // so that every route runs the same arithmetic)
    if constexpr (NT::NCONV == 3) {
      static const bool pair2 = env_int("PAAC_WGRAD_PAIR", 1) != 0;
      int wcfg, wks, wxcd;
      resolve_wgrad<false, NT::H>(gw, 1, ctx->tune[OP_FC_WGRAD][cls], wcfg, wks, wxcd);
      static const int gemm3_rows = env_int("PAAC_GEMM3_MIN_ROWS", 513);
      fc_wgrad_held = pair2 && do_conv && ctx->tower_on && wcfg == 1 && !(batch >= gemm3_rows && (batch % 32) == 0);
      held_gw = gw;
      held_ks = wks;
      held_xcd = wxcd;
    }
    static const int gemm3_min_rows = env_int("PAAC_GEMM3_MIN_ROWS", 513);
    if (!fc_wgrad_held && batch >= gemm3_min_rows && (batch % 32) == 0) {
      // dWf[FLAT, H] = X^T dH on the LDS-tiled split-bf16 GEMM (gemm3.h), 128 x 64 tiles, whole K (= rows) per workgroup:
      // written straight into the flat gradient; the row-block-0 workgroups also leave the bias gradient (column sums of dH)
      ProfScope ps(ctx, F_FC_WGRAD, batch, s);
      Gemm3Args a;
      memset(&a, 0, sizeof(a));
      a.A = xf; a.lda = NT::FLAT;            // given transposed: [K = rows, M = FLAT]
      a.B = ctx->dh; a.ldb = NT::H;          // [K = rows, N = H]
      a.out = grad + L.offset[i_wf]; a.ldo = NT::H;
      a.colsum_out = grad + L.offset[i_wf + 1];
      a.M = NT::FLAT; a.N = NT::H; a.K = batch;
      using D = Gemm3<false, false, EPI_SLAB, 4, 2, 2, 2>;
      a.MB = (NT::FLAT + D::TM - 1) / D::TM;
      a.NB = (NT::H + D::TN - 1) / D::TN;
      a.S = 1;
      a.stages_per_split = batch / 32;
      prof_mix(6);
      launch_k(gemm3_kernel<D>, dim3((unsigned)(a.MB * a.NB)), dim3(D::THREADS), s, PROF_WHOLE, a);
    } else if (!fc_wgrad_held) {
      ProfScope ps(ctx, F_FC_WGRAD, batch, s);
      launch_wgrad<typename NT::GFC, false, NT::H>(gw, 1, ctx->tune[OP_FC_WGRAD][cls], s);
    }
    ProfScope ps(ctx, F_FC_DGRAD, batch, s);
    if (batch >= gemm3_min_rows && (NT::H % 32) == 0) {
      // dX = dH * Wf^T on the LDS-tiled split-bf16 GEMM (gemm3.h); B = the fc weights as stored ([N = FLAT, K = H])
      Gemm3Args a;
      memset(&a, 0, sizeof(a));
      a.A = ctx->dh; a.lda = NT::H;
      a.B = wf; a.ldb = NT::H;
      a.out = dxf; a.ldo = NT::FLAT;
      a.mask = xf;
      a.M = batch; a.N = NT::FLAT; a.K = NT::H;
      using D = Gemm3<true, true, EPI_MASK>;
      a.MB = (batch + D::TM - 1) / D::TM;
      a.NB = (NT::FLAT + D::TN - 1) / D::TN;
      a.S = 1;
      a.stages_per_split = NT::H / 32;
      prof_mix(6);
      launch_k(gemm3_kernel<D>, dim3((unsigned)(a.MB * a.NB)), dim3(D::THREADS), s, PROF_WHOLE, a);
    } else {
      launch_dgrad<typename NT::GFCH, NT::FLAT, NT::H, EPI_MASK>(gd, 1, ctx->tune[OP_FC_DGRAD][cls], s);
    }
  }
  if (!do_conv) return 0;
  if constexpr (NT::NCONV == 3) {
    // (4) conv3 wgrad: dW3[576,64] = patches(a2)^T dY3;  (5) conv3 dgrad -> dact[1] masked by relu'(a2)
    const int feats = NT::G3::FEATS;
    constexpr int P2 = NT::G2::OPIX, P3 = NT::G3::OPIX;
    GemmArgs gw = make_args(W.act[1], (size_t)batch * P2 * NT::C2 * 4, ctx->dact[2], (size_t)batch * P3 * NT::C3 * 4, slab, nullptr, feats, NT::C3, batch * P3, NT::C3, NT::C3);
    gw.slab_rows = feats + 1;
    GemmArgs gd = make_args(ctx->dact[2], (size_t)batch * P3 * NT::C3 * 4, w3, (size_t)feats * NT::C3 * 4, ctx->dact[1], W.act[1], batch * P2, NT::C2, 9 * NT::C3, 0, NT::C2);
    // tap (kh, kw) of the full correlation reads the forward tap (2 - kh, 2 - kw)
    gd.tap_base[0] = 8 * NT::C2 * NT::C3;
    gd.tap_sh = -3 * NT::C2 * NT::C3;
    gd.tap_sw = -NT::C2 * NT::C3;
    int splits;
    int c3 = -1, k3 = 1, x3 = -1;
    resolve_wgrad<false, NT::C3>(gw, W_SPLITS_MAX, ctx->tune[OP_CONV3_WGRAD][cls], c3, k3, x3);
    if (fc_wgrad_held && c3 != 1) {        // not the pair's configuration: the held-back launch goes out on its own
      ProfScope ps(ctx, F_FC_WGRAD, batch, s);
      launch_wgrad<typename NT::GFC, false, NT::H>(held_gw, 1, ctx->tune[OP_FC_WGRAD][cls], s);
      fc_wgrad_held = false;
    }
    if (fc_wgrad_held) {
      using DF = WgradBody<typename NT::GFC, false, NT::H, 4, 2, 2>;
      using D3 = WgradBody<typename NT::G3, false, NT::C3, 4, 2, 2>;
      PairArgs pa;
      pa.g0 = held_gw;
      pa.g1 = gw;
      ProfScope ps(ctx, F_FC_CONV3_WGRAD, batch, s);
      pa.count0 = (int)prepare_dmm<DF>(pa.g0, held_ks, held_ks, held_xcd);
      const int n1 = (int)prepare_dmm<D3>(pa.g1, k3, k3, x3);
      pa.first1 = (pa.count0 + 7) / 8 * 8;
      launch_k(dmm_pair_kernel<DF, D3>, dim3((unsigned)(pa.first1 + n1)), dim3(128), s, PROF_WHOLE, pa);
      splits = k3;
    } else {
      ProfScope ps(ctx, F_CONV3_WGRAD, batch, s);
      splits = launch_wgrad<typename NT::G3, false, NT::C3>(gw, W_SPLITS_MAX, ctx->tune[OP_CONV3_WGRAD][cls], s);
    }
    if (ctx->tower_on) {
      // conv3 AND conv2 data gradients in one launch (dgrad_tower.h): da2 -> dact[1], da1 -> dact[0]
      ProfScope ps(ctx, F_DGRAD_TOWER, batch, s);
      prof_mix(6);     // conv3 and conv2 data gradients, both on the split-bf16 path
      DgradTowerArgs da;
      da.da3 = ctx->dact[2];
      da.act2 = W.act[1];
      da.act1 = W.act[0];
      const bf16x8* base = reinterpret_cast<const bf16x8*>(ctx->tower_pack) + kTowerPackVecs;
      da.w3d = base;
      da.w2d = base + kDgradW3Vecs;
      da.da2 = ctx->dact[1];
      da.da1 = ctx->dact[0];
      da.batch = batch;
      HeadsGradArgs hg;
      memset(&hg, 0, sizeof(hg));
      if (fused_heads) {
        hg.h = W.h;
        hg.dl_buf = ctx->dl_buf;
        hg.gWa = grad + L.offset[i_wa];
        hg.gba = grad + L.offset[i_wa + 1];
        hg.gWc = grad + L.offset[i_wc];
        hg.gbc = grad + L.offset[i_wc + 1];
        hg.loss_out = loss_out;
        hg.A = A;
        hg.B = batch;
        hg.blocks = NT::H / 32 + 1;
      }
      launch_k(dgrad_tower_kernel, dim3((unsigned)(batch + hg.blocks)), dim3(512), s, PROF_WHOLE, da, hg);
    } else if constexpr (NT::FAMILY) {
      ProfScope ps(ctx, F_CONV3_DGRAD, batch, s);
      launch_dgrad<typename NT::G3D, NT::C2, NT::C3, EPI_MASK>(gd, 1, ctx->tune[OP_CONV3_DGRAD][cls], s);
    } else {
      ProfScope ps(ctx, F_CONV3_DGRAD, batch, s);
      launch_dgrad_direct<typename NT::G3, NT::C3>(ctx->dact[2], w3, W.act[1], ctx->dact[1], batch, s);
    }
    wgrad_out(i_w3, feats, NT::C3, slab, splits);
    slab += (long)W_SPLITS_MAX * (feats + 1) * NT::C3;
  }
  // (6) conv2 wgrad;  (7) conv2 dgrad by output parity (4 classes in blockIdx.z) -> dact[0] masked by relu'(a1)
  {
    const int feats = NT::G2::FEATS;
    constexpr int P1 = NT::G1::OPIX, P2 = NT::G2::OPIX;
    GemmArgs gw = make_args(W.act[0], (size_t)batch * P1 * NT::C1 * 4, ctx->dact[1], (size_t)batch * P2 * NT::C2 * 4, slab, nullptr, feats, NT::C2, batch * P2, NT::C2, NT::C2);
    gw.slab_rows = feats + 1;
    GemmArgs gd = make_args(ctx->dact[1], (size_t)batch * P2 * NT::C2 * 4, w2, (size_t)feats * NT::C2 * 4, ctx->dact[0], W.act[0], batch * 100, NT::C1, 4 * NT::C2, 0, NT::C1);
    // output parity (py, px), tap (kh, kw) of the 2x2 dense correlation reads forward tap
    // (py + 2 (1 - kh), px + 2 (1 - kw)) of the 4x4 kernel
    for (int par = 0; par < 4; ++par) gd.tap_base[par] = (((par >> 1) + 2) * 4 + (par & 1) + 2) * NT::C1 * NT::C2;
    gd.tap_sh = -8 * NT::C1 * NT::C2;
    gd.tap_sw = -2 * NT::C1 * NT::C2;
    // conv2's data gradient first: conv1's weight gradient (which may share the launch below with conv2's) reads it
    if constexpr (!NT::FAMILY) {
      ProfScope ps(ctx, F_CONV2_DGRAD, batch, s);
      launch_dgrad_direct<typename NT::G2, NT::C2>(ctx->dact[1], w2, W.act[0], ctx->dact[0], batch, s);
    } else if (!(NT::NCONV == 3 && ctx->tower_on)) {     // Nature with the tower: done by dgrad_tower_kernel above
      ProfScope ps(ctx, F_CONV2_DGRAD, batch, s);
      launch_dgrad<typename NT::G2D, NT::C1, NT::C2, EPI_MASK_PARITY>(gd, 4, ctx->tune[OP_CONV2_DGRAD][cls], s);
    }
    int splits = 0;
    // The conv2 and conv1 weight gradients wait on the data-gradient tower only, and a workgroup of each fits a CU
    // together (two 4-wave bodies, 133 registers, 66.5 KB of LDS: 2 per CU = the 256 + 256 workgroups of the pair): one
    // launch.  conv1 runs its 4-wave configuration here (alone its 8-wave one is faster; in the pair it is not).
    if constexpr (NT::FAMILY) {       // (the stock NIPS geometry too: its heuristics land on the pair's configurations)
      static const bool pair_on = env_int("PAAC_WGRAD_PAIR", 1) != 0;
      const int f1 = NT::G1::FEATS;
      GemmArgs g1 = make_args(states, (size_t)batch * 28224, ctx->dact[0], (size_t)batch * P1 * NT::C1 * 4,
                              slab + (long)W_SPLITS_MAX * (feats + 1) * NT::C2, nullptr, f1, NT::C1, batch * P1, NT::C1, NT::C1);
      g1.slab_rows = f1 + 1;
      int c2, k2, x2, c1, k1, x1;
      resolve_wgrad<false, NT::C2>(gw, W_SPLITS_MAX, ctx->tune[OP_CONV2_WGRAD][cls], c2, k2, x2);
      resolve_wgrad<true, NT::C1>(g1, W_SPLITS_MAX, ctx->tune[OP_CONV1_WGRAD][cls], c1, k1, x1);
      if (pair_on && (ctx->tower_on || ctx->tower2_on) && c2 == 0 && c1 == kExactBf16 + 0) {
        using D2 = WgradBody<typename NT::G2, false, NT::C2, 4, 4, 2>;
        using D1 = WgradBody<typename NT::G1, true, NT::C1, 4, 4, 2, 1>;
        PairArgs pa;
        pa.g0 = gw;
        pa.g1 = g1;
        {
          ProfScope ps(ctx, F_CONV2_CONV1_WGRAD, batch, s);
          pa.count0 = (int)prepare_dmm<D2>(pa.g0, k2, k2, x2);
          const int n1 = (int)prepare_dmm<D1>(pa.g1, k1, k1, x1);
          pa.first1 = (pa.count0 + 7) / 8 * 8;
          launch_k(dmm_pair_kernel<D2, D1>, dim3((unsigned)(pa.first1 + n1)), dim3(256), s, PROF_WHOLE, pa);
        }
        splits = k2;
        conv1_splits_done = k1;
      }
    }
    if (splits == 0) {
      ProfScope ps(ctx, F_CONV2_WGRAD, batch, s);
      splits = launch_wgrad<typename NT::G2, false, NT::C2>(gw, W_SPLITS_MAX, ctx->tune[OP_CONV2_WGRAD][cls], s);
    }
    wgrad_out(i_w2, feats, NT::C2, slab, splits);
    slab += (long)W_SPLITS_MAX * (feats + 1) * NT::C2;
  }
  // (8) conv1 wgrad from the u8 frames
  {
    const int feats = NT::G1::FEATS;
    constexpr int P1 = NT::G1::OPIX;
    int splits = conv1_splits_done;
    if (splits == 0) {
      ProfScope ps(ctx, F_CONV1_WGRAD, batch, s);
      GemmArgs g = make_args(states, (size_t)batch * 28224, ctx->dact[0], (size_t)batch * P1 * NT::C1 * 4, slab, nullptr, feats, NT::C1, batch * P1, NT::C1, NT::C1);
      g.slab_rows = feats + 1;
      splits = launch_wgrad<typename NT::G1, true, NT::C1>(g, W_SPLITS_MAX, ctx->tune[OP_CONV1_WGRAD][cls], s);
    }
    wgrad_out(i_w1, feats, NT::C1, slab, splits);
  }
  ctx->pending_fin_grad = nullptr;
  if (phase == 3) {
    ctx->pending_fin = fin;
    ctx->pending_fin_grad = grad;
  } else {
    ProfScope ps(ctx, F_GRAD_FINALIZE, batch, s);
    int maxcount = 0;
    for (int i = 0; i < fin.nseg; ++i) maxcount = fin.seg[i].count > maxcount ? fin.seg[i].count : maxcount;
    launch_k(grad_finalize_kernel, dim3((maxcount / 4 * FIN_LANES + 255) / 256, fin.nseg), dim3(256), s, PROF_WHOLE, fin);
  }
  return 0;
}

int launch_backward(paac_ctx* ctx, const float* params, const uint8_t* states, const int32_t* actions, const float* y,
                    const float* adv, int batch, float beta, float* grad, float* loss_out, int phase, hipStream_t s,
                    const paac_returns* ret) {
  ReturnsArgs rt;
  memset(&rt, 0, sizeof(rt));
  if (ret) {
    rt.v_boot = ret->v_boot; rt.boot_in_fwd = ret->v_boot ? 0 : 1; rt.rewards = ret->rewards; rt.masks = ret->masks; rt.values_act = ret->values;
    rt.T = ret->T; rt.N = ret->N; rt.gamma = ret->gamma; rt.y_out = ret->y_out; rt.adv_out = ret->adv_out;
    rt.global_step = ret->global_step_dev; rt.step_inc = ret->increment; rt.lr0 = ret->initial_lr;
    rt.anneal = ret->lr_annealing_steps; rt.lr_out = ret->lr_out_dev; rt.tick = ret->tick_dev; rt.tick_inc = ret->tick_inc;
  }
  if (ctx->cfg.arch == PAAC_ARCH_NATURE)
    return backward_impl<NatureNet>(ctx, params, states, actions, y, adv, batch, beta, grad, loss_out, phase, rt, s);
  return backward_impl<OtherNet>(ctx, params, states, actions, y, adv, batch, beta, grad, loss_out, phase, rt, s);
}

int64_t wslab_floats_needed(int arch) {
  if (arch == PAAC_ARCH_NATURE)
    return (int64_t)W_SPLITS_MAX * ((NatureNet::G3::FEATS + 1) * NatureNet::C3 + (NatureNet::G2::FEATS + 1) * NatureNet::C2 +
                                    257 * NatureNet::C1);
  if (OtherNet::NCONV == 3)
    return (int64_t)W_SPLITS_MAX * ((OtherNet::G3::FEATS + 1) * OtherNet::C3 + (OtherNet::G2::FEATS + 1) * OtherNet::C2 +
                                    (OtherNet::G1::FEATS + 1) * OtherNet::C1);
  return (int64_t)W_SPLITS_MAX * ((OtherNet::G2::FEATS + 1) * OtherNet::C2 + (OtherNet::G1::FEATS + 1) * OtherNet::C1);
}

}  // namespace paac

namespace paac {
// Launch tuning measured on MI355X with tools/tune_gemm.py on the benchmark shapes (Nature, 32 envs x t_max 5:
// batch 32 acting, 160 training; profiles/r01_tune_gemm.txt).  The sweep is flat -- the size heuristics are within
// 1 us of the best everywhere except the entries below; anything else (other archs, larger batches) keeps the
// heuristics.
void default_tuning(paac_ctx* c) {
  if (c->cfg.arch != PAAC_ARCH_NATURE) return;
  if (c->max_batch <= 2048) {
    // class 2 (more than 512 rows): TUNE_N=256 tools/tune_gemm.py (256 envs x t_max 5: 1536-row training forward,
    // 1280-row backward).  At these sizes the K loops dominate and the split-bf16 path wins for most ops.  Contexts
    // sized for more than 2048 rows keep the size heuristics there (measured better at 2688 rows).
    c->tune[OP_CONV1_FWD][2] = Tune{kExactBf16 + 11, 0, -1};
    c->tune[OP_CONV2_FWD][2] = Tune{kSplitBf16 + 11, 0, -1};
    c->tune[OP_CONV3_FWD][2] = Tune{kSplitBf16 + 10, 0, -1};
    c->tune[OP_FC_FWD][2] = Tune{kSplitBf16 + 4, 4, -1};
    c->tune[OP_FC_WGRAD][2] = Tune{0, 1, 0};
    c->tune[OP_FC_DGRAD][2] = Tune{kSplitBf16 + 10, 0, -1};
    c->tune[OP_CONV3_WGRAD][2] = Tune{kSplitBf16 + 1, 48, 2};
    c->tune[OP_CONV3_DGRAD][2] = Tune{kSplitBf16 + 10, 0, -1};
    c->tune[OP_CONV2_WGRAD][2] = Tune{kSplitBf16 + 3, 32, 2};
    c->tune[OP_CONV2_DGRAD][2] = Tune{10, 0, 0};
    c->tune[OP_CONV1_WGRAD][2] = Tune{kExactBf16 + 2, 64, 2};
  } else {
    // contexts sized for more than 2048 rows (the 8 x 128-environment shards at t_max 20: 2560 / 2688 rows): the conv
    // weight gradients, TUNE_N=128 TUNE_T=20 TUNE_A=18 TUNE_OPS=6,8,10 tools/tune_gemm.py -- update 1135 -> 1040 us; the
    // other ops keep the size heuristics there (the fc layer runs on gemm3.h, the conv data path on the tower kernels)
    c->tune[OP_CONV3_WGRAD][2] = Tune{kSplitBf16 + 0, 48, 2};
    c->tune[OP_CONV2_WGRAD][2] = Tune{kSplitBf16 + 1, 64, 2};
    c->tune[OP_CONV1_WGRAD][2] = Tune{kExactBf16 + 2, 64, 2};
  }
  // classes 0 and 1: tools/tune_gemm.py on MI355X, 32 envs x t_max 5: acting batch 32 (class 0); training forward over
  // 192 rows and backward over 160 (class 1)
  c->tune[OP_CONV1_FWD][0] = Tune{kExactBf16 + 4, 0, -1};
  c->tune[OP_CONV1_FWD][1] = Tune{kExactBf16 + 10, 0, -1};
  c->tune[OP_CONV2_FWD][1] = Tune{kSplitBf16 + 12, 0, -1};
  c->tune[OP_CONV3_FWD][0] = Tune{kNarrow + 1, 0, -1};
  c->tune[OP_CONV3_FWD][1] = Tune{kSplitBf16 + 12, 0, -1};
  c->tune[OP_FC_FWD][0] = Tune{0, 8, 2};
  c->tune[OP_FC_FWD][1] = Tune{kSplitBf16 + 1, 8, 2};
  c->tune[OP_FC_WGRAD][1] = Tune{1, 1, 0};             // 2-wave body: shares a launch with conv3's (backward_impl)
  c->tune[OP_FC_DGRAD][1] = Tune{kSplitBf16 + 1, 0, -1};
  c->tune[OP_CONV3_WGRAD][1] = Tune{1, 64, 2};           // fp32 2-wave body (alone the split-bf16 one is 1.5 us faster);
                                                          // 64 slabs: 576 + 448 workgroups = the 1024 that are resident at once
  c->tune[OP_CONV3_DGRAD][1] = Tune{kSplitBf16 + 11, 0, -1};
  c->tune[OP_CONV2_WGRAD][1] = Tune{0, 32, 2};
  c->tune[OP_CONV2_DGRAD][1] = Tune{9, 0, -1};
  c->tune[OP_CONV1_WGRAD][1] = Tune{kExactBf16 + 0, 64, 2};   // 4-wave body: shares a launch with conv2's
}
}  // namespace paac

