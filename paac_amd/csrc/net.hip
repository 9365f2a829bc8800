// Actor-critic network of the PAAC hot path on gfx950: forward, loss, backward.
//   networks.py:100-169 (trunk), policy_v_network.py:6-57 (heads + loss), and the gradient graph
//   optimizer.compute_gradients(loss) builds from them (actor_learner.py:44).
// Layout contract: activations NHWC fp32, conv weights HWIO, fc weights [in,out], flatten in HWC order
// (networks.py:6-9) -- so every weight tensor is already the row-major [K,N] B-matrix of its GEMM.
#include "igemm.h"

namespace paac {

// ---------------------------------------------------------------------------------------------
// Compile-time network descriptions.
struct NatureNet {
  static constexpr int NCONV = 3, C1 = 32, C2 = 64, C3 = 64, H = 512, FLAT = 3136;
  using G1 = Geom<84, 84, 4, 20, 20, 4, 0, 0, 8, 8>;
  using G2 = Geom<20, 20, 32, 9, 9, 2, 0, 0, 4, 4>;
  using G3 = Geom<9, 9, 64, 7, 7, 1, 0, 0, 3, 3>;
  using GFC = Geom<1, 1, 3136, 1, 1, 1, 0, 0, 1, 1>;   // rows of the flattened last conv output
  using GFCH = Geom<1, 1, 512, 1, 1, 1, 0, 0, 1, 1>;   // rows of dH
  using G3D = Geom<7, 7, 64, 9, 9, 1, 2, 2, 3, 3>;     // conv3 dgrad: full correlation over dY3
  using G2D = Geom<9, 9, 64, 10, 10, 1, 1, 1, 2, 2>;   // conv2 dgrad, one output parity class
};
struct NipsNet {
  static constexpr int NCONV = 2, C1 = 16, C2 = 32, C3 = 32, H = 256, FLAT = 2592;
  using G1 = Geom<84, 84, 4, 20, 20, 4, 0, 0, 8, 8>;
  using G2 = Geom<20, 20, 16, 9, 9, 2, 0, 0, 4, 4>;
  using G3 = Geom<9, 9, 32, 7, 7, 1, 0, 0, 3, 3>;      // unused
  using GFC = Geom<1, 1, 2592, 1, 1, 1, 0, 0, 1, 1>;
  using GFCH = Geom<1, 1, 256, 1, 1, 1, 0, 0, 1, 1>;
  using G3D = Geom<7, 7, 32, 9, 9, 1, 2, 2, 3, 3>;     // unused
  using G2D = Geom<9, 9, 32, 10, 10, 1, 1, 1, 2, 2>;
};

constexpr int FC_SPLITS_MAX = 16;
constexpr int W_SPLITS_MAX = 64;
constexpr int MAXA = 32;

// Tile-config dispatch by the GEMM's compile-time N and the grid it would produce.
template <class G, bool U8, int AM, int BMo, int BCO, int EPI, bool BR, int NDIM>
static void launch_auto(const GemmArgs& a, int zdim, hipStream_t s) {
  if constexpr (NDIM % 32 != 0) {
    launch_igemm<G, U8, AM, BMo, BCO, EPI, BR, 64, 16, 4, 1, 1>(a, zdim, s);
  } else {
    constexpr int BNBIG = (NDIM % 64 == 0) ? 64 : 32;
    const long big_blocks = (long)((a.M + 63) / 64) * ((a.N + BNBIG - 1) / BNBIG) * zdim;
    if (big_blocks < 256) {
      launch_igemm<G, U8, AM, BMo, BCO, EPI, BR, 32, 32, 1, 1, 4>(a, zdim, s);
    } else if constexpr (NDIM % 64 == 0) {
      launch_igemm<G, U8, AM, BMo, BCO, EPI, BR, 64, 64, 2, 2, 1>(a, zdim, s);
    } else {
      launch_igemm<G, U8, AM, BMo, BCO, EPI, BR, 64, 32, 2, 1, 2>(a, zdim, s);
    }
  }
}

static GemmArgs make_args(const void* A, const float* B, float* out, const float* aux, int M, int N, int K, int a_rows,
                          int ldb, int ldo) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.out = out; g.aux = aux;
  g.M = M; g.N = N; g.K = K; g.a_rows = a_rows; g.ldb = ldb; g.ldo = ldo;
  g.chunks_per_split = (K + 31) / 32;
  g.slab_rows = M;
  return g;
}

// ---------------------------------------------------------------------------------------------
// Heads forward: h = relu(sum of fc split-K slabs + b); logits = h Wa + ba; pi = softmax; v = h Wc + bc.
// policy_v_network.py:24-26,37 / networks.py:84-89.  One 256-thread workgroup per batch row;
// wavefront shuffles for the A+1 dot-product reductions.
template <int H>
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ slab, int splits, long slab_stride,
                                                        const float* __restrict__ fc_b, const float* __restrict__ Wa,
                                                        const float* __restrict__ ba, const float* __restrict__ Wc,
                                                        const float* __restrict__ bc, int A, float* __restrict__ h_out,
                                                        float* __restrict__ logits_ws, float* __restrict__ probs_ws,
                                                        float* __restrict__ values_ws, float* __restrict__ logits_out,
                                                        float* __restrict__ probs_out, float* __restrict__ values_out) {
  const int i = blockIdx.x;
  const int tid = threadIdx.x;
  float part[MAXA + 1];
#pragma unroll
  for (int a = 0; a <= MAXA; ++a) part[a] = 0.f;
  for (int j = tid; j < H; j += 256) {
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += slab[sp * slab_stride + (long)i * H + j];
    s = fmaxf(s + fc_b[j], 0.f);
    h_out[(long)i * H + j] = s;
#pragma unroll
    for (int a = 0; a < MAXA; ++a)
      if (a < A) part[a] += s * Wa[j * A + a];
    part[MAXA] += s * Wc[j];
  }
  __shared__ float red[4][MAXA + 1];
  __shared__ float lg[MAXA + 1];
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int a = 0; a <= MAXA; ++a) {
    if (a < A || a == MAXA) {
      float v = part[a];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) red[wave][a] = v;
    }
  }
  __syncthreads();
  if (tid <= MAXA && (tid < A || tid == MAXA)) {
    const float b = (tid == MAXA) ? bc[0] : ba[tid];
    lg[tid] = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) + b;
  }
  __syncthreads();
  if (tid == 0) {
    float m = lg[0];
    for (int a = 1; a < A; ++a) m = fmaxf(m, lg[a]);
    float e[MAXA];
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < MAXA; ++a)
      if (a < A) {
        e[a] = expf(lg[a] - m);
        sum += e[a];
      }
#pragma unroll
    for (int a = 0; a < MAXA; ++a)
      if (a < A) {
        const float pa = e[a] / sum;
        probs_ws[(long)i * A + a] = pa;
        logits_ws[(long)i * A + a] = lg[a];
        if (probs_out) probs_out[(long)i * A + a] = pa;
        if (logits_out) logits_out[(long)i * A + a] = lg[a];
      }
    values_ws[i] = lg[MAXA];
    if (values_out) values_out[i] = lg[MAXA];
  }
}

// ---------------------------------------------------------------------------------------------
// Loss gradient wrt heads (policy_v_network.py:29-57; analytic form in DESIGN.md / SURVEY A.4):
//   s = 5/B; dv = s*0.5*(v - y); g_a = -(adv*1[a=act]/(pi_a+eps) - beta*(log(pi_a+eps) + pi_a/(pi_a+eps)))
//   dlogit_a = s*pi_a*(g_a - sum_j g_j pi_j)
__device__ __forceinline__ void head_grad_row(const float* __restrict__ pi, float v, int act, float y, float adv,
                                              float beta, float s, int A, float* dl /*[A]*/, float* dv, float* stats) {
  const float eps = 1e-30f;
  float g[MAXA];
  float dot = 0.f, ent = 0.f, logp = 0.f;
#pragma unroll
  for (int a = 0; a < MAXA; ++a)
    if (a < A) {
      const float p = pi[a];
      const float lp = logf(p + eps);
      const float inv = 1.0f / (p + eps);
      const float oh = (a == act) ? 1.f : 0.f;
      g[a] = -(adv * oh * inv - beta * (lp + p * inv));
      dot += g[a] * p;
      ent -= p * lp;
      logp += oh * lp;
    }
#pragma unroll
  for (int a = 0; a < MAXA; ++a)
    if (a < A) dl[a] = s * pi[a] * (g[a] - dot);
  *dv = s * 0.5f * (v - y);
  if (stats) {
    stats[0] = -(logp * adv + beta * ent);      // actor objective term
    stats[1] = 0.25f * (y - v) * (y - v);       // critic term
    stats[2] = ent;
  }
}

// One launch, three roles by blockIdx:
//   [0, B)            : row i -> dH[i,:] = (dlogits Wa^T + dv Wc^T) * 1[h > 0]
//   [B, B + H/256)    : head weight gradients dWa[j,:], dWc[j] for 256 values of j (loops over rows)
//   B + H/256         : head bias gradients + loss scalars
template <int H>
__global__ __launch_bounds__(256) void heads_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ values,
                                                        const int32_t* __restrict__ actions, const float* __restrict__ y,
                                                        const float* __restrict__ adv, const float* __restrict__ h,
                                                        const float* __restrict__ Wa, const float* __restrict__ Wc,
                                                        int A, int B, float beta, float* __restrict__ dH,
                                                        float* __restrict__ gWa, float* __restrict__ gba,
                                                        float* __restrict__ gWc, float* __restrict__ gbc,
                                                        float* __restrict__ loss_out) {
  const int tid = threadIdx.x;
  const float s = 5.0f / (float)B;
  __shared__ float sdl[64][MAXA + 1];
  if ((int)blockIdx.x < B) {
    const int i = blockIdx.x;
    if (tid == 0) {
      float dl[MAXA], dv;
      head_grad_row(probs + (long)i * A, values[i], actions[i], y[i], adv[i], beta, s, A, dl, &dv, nullptr);
#pragma unroll
      for (int a = 0; a < MAXA; ++a)
        if (a < A) sdl[0][a] = dl[a];
      sdl[0][MAXA] = dv;
    }
    __syncthreads();
    for (int j = tid; j < H; j += 256) {
      float acc = sdl[0][MAXA] * Wc[j];
      for (int a = 0; a < A; ++a) acc += sdl[0][a] * Wa[j * A + a];
      dH[(long)i * H + j] = h[(long)i * H + j] > 0.f ? acc : 0.f;
    }
    return;
  }
  const int role = blockIdx.x - B;
  if (role < H / 256) {
    const int j = role * 256 + tid;
    float acc[MAXA + 1];
#pragma unroll
    for (int a = 0; a <= MAXA; ++a) acc[a] = 0.f;
    for (int i0 = 0; i0 < B; i0 += 64) {
      __syncthreads();
      if (tid < 64 && i0 + tid < B) {
        const int i = i0 + tid;
        float dl[MAXA], dv;
        head_grad_row(probs + (long)i * A, values[i], actions[i], y[i], adv[i], beta, s, A, dl, &dv, nullptr);
#pragma unroll
        for (int a = 0; a < MAXA; ++a)
          if (a < A) sdl[tid][a] = dl[a];
        sdl[tid][MAXA] = dv;
      }
      __syncthreads();
      const int cnt = min(64, B - i0);
      for (int r = 0; r < cnt; ++r) {
        const float hv = h[(long)(i0 + r) * H + j];
#pragma unroll
        for (int a = 0; a < MAXA; ++a)
          if (a < A) acc[a] += hv * sdl[r][a];
        acc[MAXA] += hv * sdl[r][MAXA];
      }
    }
#pragma unroll
    for (int a = 0; a < MAXA; ++a)
      if (a < A) gWa[j * A + a] = acc[a];
    gWc[j] = acc[MAXA];
    return;
  }
  // bias gradients + loss scalars
  float accb[MAXA + 1];
  float st[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a <= MAXA; ++a) accb[a] = 0.f;
  for (int i = tid; i < B; i += 256) {
    float dl[MAXA], dv, stats[3];
    head_grad_row(probs + (long)i * A, values[i], actions[i], y[i], adv[i], beta, s, A, dl, &dv, stats);
#pragma unroll
    for (int a = 0; a < MAXA; ++a)
      if (a < A) accb[a] += dl[a];
    accb[MAXA] += dv;
    st[0] += stats[0]; st[1] += stats[1]; st[2] += stats[2];
  }
  __shared__ float red[4][MAXA + 4];
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int a = 0; a <= MAXA + 3; ++a) {
    float v = (a <= MAXA) ? accb[a] : st[a - MAXA - 1];
    if (a < A || a >= MAXA) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) red[wave][a] = v;
    }
  }
  __syncthreads();
  if (tid <= MAXA + 3 && (tid < A || tid >= MAXA)) {
    const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    if (tid < A) gba[tid] = v;
    else if (tid == MAXA) gbc[0] = v;
    else if (loss_out) red[0][tid] = v;
  }
  __syncthreads();
  if (tid == 0 && loss_out) {
    const float actor = red[0][MAXA + 1] / (float)B;
    const float critic = red[0][MAXA + 2] / (float)B;
    loss_out[0] = 5.0f * (actor + critic);
    loss_out[1] = actor;
    loss_out[2] = critic;
    loss_out[3] = red[0][MAXA + 3] / (float)B;
  }
}

// ---------------------------------------------------------------------------------------------
// Split-K slab reduction into the flat gradient (deterministic: fixed summation order).
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
__global__ __launch_bounds__(256) void grad_finalize_kernel(const FinalizeArgs a) {
  const FinalizeSeg sg = a.seg[blockIdx.y];
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= sg.count) return;
  float4 v = *reinterpret_cast<const float4*>(sg.src + i);
  for (int s = 1; s < sg.splits; ++s) {
    const float4 t = *reinterpret_cast<const float4*>(sg.src + s * sg.stride + i);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  *reinterpret_cast<float4*>(sg.dst + i) = v;
}

// ---------------------------------------------------------------------------------------------
static int tensor_index(const paac_ctx* ctx, int which /*0 conv1 .. fc, actor, critic*/, bool bias) {
  return which * 2 + (bias ? 1 : 0);
}

template <class NT>
static int forward_impl(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, float* logits,
                        float* probs, float* values, hipStream_t s) {
  const paac_layout& L = ctx->layout;
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

  {
    ProfScope ps(ctx, F_CONV1_FWD, s);
    GemmArgs g = make_args(states, w1, ctx->act[0], b1, batch * 400, NT::C1, 256, batch * 400, NT::C1, NT::C1);
    launch_auto<typename NT::G1, true, A_ROWS_M, B_KN, 1, EPI_BIAS_RELU, false, NT::C1>(g, 1, s);
  }
  {
    ProfScope ps(ctx, F_CONV2_FWD, s);
    GemmArgs g = make_args(ctx->act[0], w2, ctx->act[1], b2, batch * 81, NT::C2, 16 * NT::C1, batch * 81, NT::C2, NT::C2);
    launch_auto<typename NT::G2, false, A_ROWS_M, B_KN, 1, EPI_BIAS_RELU, false, NT::C2>(g, 1, s);
  }
  const float* last = ctx->act[1];
  if constexpr (NT::NCONV == 3) {
    ProfScope ps(ctx, F_CONV3_FWD, s);
    GemmArgs g = make_args(ctx->act[1], w3, ctx->act[2], b3, batch * 49, NT::C3, 9 * NT::C2, batch * 49, NT::C3, NT::C3);
    launch_auto<typename NT::G3, false, A_ROWS_M, B_KN, 1, EPI_BIAS_RELU, false, NT::C3>(g, 1, s);
    last = ctx->act[2];
  }
  int splits = 1;
  {
    ProfScope ps(ctx, F_FC_FWD, s);
    GemmArgs g = make_args(last, wf, ctx->fc_slab, nullptr, batch, NT::H, NT::FLAT, batch, NT::H, NT::H);
    const int chunks = (NT::FLAT + 31) / 32;
    if (batch < 512) {
      const int tiles = ((batch + 31) / 32) * (NT::H / 32);
      splits = (256 + tiles - 1) / tiles;
      if (splits > FC_SPLITS_MAX) splits = FC_SPLITS_MAX;
      if (splits < 1) splits = 1;
    }
    int cps = (chunks + splits - 1) / splits;
    splits = (chunks + cps - 1) / cps;
    g.chunks_per_split = cps;
    g.slab_rows = batch;
    launch_auto<typename NT::GFC, false, A_ROWS_M, B_KN, 1, EPI_SLAB, false, NT::H>(g, splits, s);
  }
  {
    ProfScope ps(ctx, F_HEADS_FWD, s);
    hipLaunchKernelGGL((heads_fwd_kernel<NT::H>), dim3(batch), dim3(256), 0, s, ctx->fc_slab, splits,
                       (long)batch * NT::H, bf, wa, ba, wc, bc, A, ctx->h, ctx->logits, ctx->probs, ctx->values, logits,
                       probs, values);
  }
  return 0;
}

static void split_plan(int tiles, int chunks, int max_splits, int* splits, int* cps) {
  int s = (320 + tiles - 1) / tiles;
  if (s > max_splits) s = max_splits;
  if (s > chunks) s = chunks;
  if (s < 1) s = 1;
  int c = (chunks + s - 1) / s;
  s = (chunks + c - 1) / c;
  *splits = s;
  *cps = c;
}

template <class NT>
static int backward_impl(paac_ctx* ctx, const float* params, const uint8_t* states, const int32_t* actions,
                         const float* y, const float* adv, int batch, float beta, float* grad, float* loss_out,
                         hipStream_t s) {
  const paac_layout& L = ctx->layout;
  const int A = ctx->cfg.num_actions;
  const int nt = L.num_tensors;
  // tensor indices
  const int i_w1 = 0, i_w2 = 2, i_w3 = 4;
  const int i_wf = (NT::NCONV == 3) ? 6 : 4;
  const int i_wa = i_wf + 2, i_wc = i_wf + 4;
  (void)nt;
  const float* wa = params + L.offset[i_wa];
  const float* wc = params + L.offset[i_wc];
  const float* wf = params + L.offset[i_wf];
  const float* w2 = params + L.offset[i_w2];
  const float* w3 = (NT::NCONV == 3) ? params + L.offset[i_w3] : nullptr;

  // (1) heads: dH, head weight/bias grads, loss scalars
  {
    ProfScope ps(ctx, F_HEADS_BWD, s);
    hipLaunchKernelGGL((heads_bwd_kernel<NT::H>), dim3(batch + NT::H / 256 + 1), dim3(256), 0, s, ctx->probs,
                       ctx->values, actions, y, adv, ctx->h, wa, wc, A, batch, beta, ctx->dh, grad + L.offset[i_wa],
                       grad + L.offset[i_wa + 1], grad + L.offset[i_wc], grad + L.offset[i_wc + 1], loss_out);
  }
  const float* xf = (NT::NCONV == 3) ? ctx->act[2] : ctx->act[1];   // flattened last conv output
  float* dxf = (NT::NCONV == 3) ? ctx->dact[2] : ctx->dact[1];
  // (2) fc wgrad (+ bias row) straight into the flat gradient: [FLAT+1][H] = fc_w then fc_b
  {
    ProfScope ps(ctx, F_FC_WGRAD, s);
    GemmArgs g = make_args(xf, ctx->dh, grad + L.offset[i_wf], nullptr, NT::FLAT, NT::H, batch, batch, NT::H, NT::H);
    g.slab_rows = NT::FLAT + 1;
    launch_auto<typename NT::GFC, false, A_ROWS_K, B_KN, 1, EPI_SLAB, true, NT::H>(g, 1, s);
  }
  // (3) fc dgrad, masked by relu'(last conv output)
  {
    ProfScope ps(ctx, F_FC_DGRAD, s);
    GemmArgs g = make_args(ctx->dh, wf, dxf, xf, batch, NT::FLAT, NT::H, batch, 0, NT::FLAT);
    g.tapoff[0][0] = 0;
    launch_auto<typename NT::GFCH, false, A_ROWS_M, B_NK_TAPS, NT::H, EPI_MASK, false, NT::FLAT>(g, 1, s);
  }
  FinalizeArgs fin;
  memset(&fin, 0, sizeof(fin));
  float* slab = ctx->wslab;
  auto add_segments = [&](int i_w, int feats, int cout, int splits, float* base) {
    if (splits <= 1) return;
    const long stride = (long)(feats + 1) * cout;
    fin.seg[fin.nseg++] = FinalizeSeg{base, grad + L.offset[i_w], feats * cout, splits, stride};
    fin.seg[fin.nseg++] = FinalizeSeg{base + (long)feats * cout, grad + L.offset[i_w + 1], cout, splits, stride};
  };
  if constexpr (NT::NCONV == 3) {
    // (4) conv3 wgrad: dW3[576,64] = patches(a2)^T dY3, split-K slabs
    int splits, cps;
    {
      ProfScope ps(ctx, F_CONV_WGRAD, s);
      const int feats = NT::G3::FEATS;
      GemmArgs g = make_args(ctx->act[1], ctx->dact[2], nullptr, nullptr, feats, NT::C3, batch * 49, batch * 49, NT::C3, NT::C3);
      split_plan(((feats + 63) / 64) * ((NT::C3 + 63) / 64), (batch * 49 + 31) / 32, W_SPLITS_MAX, &splits, &cps);
      g.chunks_per_split = cps;
      g.slab_rows = feats + 1;
      g.out = (splits > 1) ? slab : grad + L.offset[i_w3];
      launch_igemm<typename NT::G3, false, A_ROWS_K, B_KN, 1, EPI_SLAB, true, 64, 64, 2, 2, 1>(g, splits, s);
      add_segments(i_w3, feats, NT::C3, splits, slab);
      slab += (long)W_SPLITS_MAX * (feats + 1) * NT::C3;
    }
    // (5) conv3 dgrad -> dact[1] masked by relu'(a2)
    {
      ProfScope ps(ctx, F_CONV_DGRAD, s);
      GemmArgs g = make_args(ctx->dact[2], w3, ctx->dact[1], ctx->act[1], batch * 81, NT::C2, 9 * NT::C3, batch * 81, 0, NT::C2);
      for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) g.tapoff[0][kh * 3 + kw] = ((2 - kh) * 3 + (2 - kw)) * NT::C2 * NT::C3;
      launch_auto<typename NT::G3D, false, A_ROWS_M, B_NK_TAPS, NT::C3, EPI_MASK, false, NT::C2>(g, 1, s);
    }
  }
  // (6) conv2 wgrad
  {
    int splits, cps;
    ProfScope ps(ctx, F_CONV_WGRAD, s);
    const int feats = NT::G2::FEATS;
    GemmArgs g = make_args(ctx->act[0], ctx->dact[1], nullptr, nullptr, feats, NT::C2, batch * 81, batch * 81, NT::C2, NT::C2);
    constexpr int BNW = (NT::C2 % 64 == 0) ? 64 : 32;
    split_plan(((feats + 63) / 64) * ((NT::C2 + BNW - 1) / BNW), (batch * 81 + 31) / 32, W_SPLITS_MAX, &splits, &cps);
    g.chunks_per_split = cps;
    g.slab_rows = feats + 1;
    g.out = (splits > 1) ? slab : grad + L.offset[i_w2];
    if constexpr (NT::C2 % 64 == 0)
      launch_igemm<typename NT::G2, false, A_ROWS_K, B_KN, 1, EPI_SLAB, true, 64, 64, 2, 2, 1>(g, splits, s);
    else
      launch_igemm<typename NT::G2, false, A_ROWS_K, B_KN, 1, EPI_SLAB, true, 64, 32, 2, 1, 2>(g, splits, s);
    add_segments(i_w2, feats, NT::C2, splits, slab);
    slab += (long)W_SPLITS_MAX * (feats + 1) * NT::C2;
  }
  // (7) conv2 dgrad by output parity (4 classes in blockIdx.z) -> dact[0] masked by relu'(a1)
  {
    ProfScope ps(ctx, F_CONV_DGRAD, s);
    GemmArgs g = make_args(ctx->dact[1], w2, ctx->dact[0], ctx->act[0], batch * 100, NT::C1, 4 * NT::C2, batch * 100, 0, NT::C1);
    for (int par = 0; par < 4; ++par) {
      const int py = par >> 1, px = par & 1;
      for (int kh = 0; kh < 2; ++kh)
        for (int kw = 0; kw < 2; ++kw)
          g.tapoff[par][kh * 2 + kw] = ((py + 2 * (1 - kh)) * 4 + (px + 2 * (1 - kw))) * NT::C1 * NT::C2;
    }
    if constexpr (NT::C1 % 32 == 0)
      launch_igemm<typename NT::G2D, false, A_ROWS_M, B_NK_TAPS, NT::C2, EPI_MASK_PARITY, false, 64, 32, 2, 1, 2>(g, 4, s);
    else
      launch_igemm<typename NT::G2D, false, A_ROWS_M, B_NK_TAPS, NT::C2, EPI_MASK_PARITY, false, 64, 16, 4, 1, 1>(g, 4, s);
  }
  // (8) conv1 wgrad from the u8 frames
  {
    int splits, cps;
    ProfScope ps(ctx, F_CONV1_WGRAD, s);
    const int feats = 256;
    GemmArgs g = make_args(states, ctx->dact[0], nullptr, nullptr, feats, NT::C1, batch * 400, batch * 400, NT::C1, NT::C1);
    split_plan(4, (batch * 400 + 31) / 32, W_SPLITS_MAX, &splits, &cps);
    g.chunks_per_split = cps;
    g.slab_rows = feats + 1;
    g.out = (splits > 1) ? slab : grad + L.offset[i_w1];
    if constexpr (NT::C1 % 32 == 0)
      launch_igemm<typename NT::G1, true, A_ROWS_K, B_KN, 1, EPI_SLAB, true, 64, 32, 2, 1, 2>(g, splits, s);
    else
      launch_igemm<typename NT::G1, true, A_ROWS_K, B_KN, 1, EPI_SLAB, true, 64, 16, 4, 1, 1>(g, splits, s);
    add_segments(i_w1, feats, NT::C1, splits, slab);
  }
  if (fin.nseg > 0) {
    ProfScope ps(ctx, F_GRAD_FINALIZE, s);
    int maxcount = 0;
    for (int i = 0; i < fin.nseg; ++i) maxcount = fin.seg[i].count > maxcount ? fin.seg[i].count : maxcount;
    hipLaunchKernelGGL(grad_finalize_kernel, dim3((maxcount / 4 + 255) / 256, fin.nseg), dim3(256), 0, s, fin);
  }
  return 0;
}

int launch_forward(paac_ctx* ctx, const float* params, const uint8_t* states, int batch, bool, float* logits,
                   float* probs, float* values, hipStream_t s) {
  if (ctx->cfg.arch == PAAC_ARCH_NATURE) return forward_impl<NatureNet>(ctx, params, states, batch, logits, probs, values, s);
  return forward_impl<NipsNet>(ctx, params, states, batch, logits, probs, values, s);
}

int launch_backward(paac_ctx* ctx, const float* params, const uint8_t* states, const int32_t* actions, const float* y,
                    const float* adv, int batch, float beta, float* grad, float* loss_out, hipStream_t s) {
  if (ctx->cfg.arch == PAAC_ARCH_NATURE)
    return backward_impl<NatureNet>(ctx, params, states, actions, y, adv, batch, beta, grad, loss_out, s);
  return backward_impl<NipsNet>(ctx, params, states, actions, y, adv, batch, beta, grad, loss_out, s);
}

int64_t wslab_floats_needed(int arch) {
  if (arch == PAAC_ARCH_NATURE)
    return (int64_t)W_SPLITS_MAX * ((NatureNet::G3::FEATS + 1) * NatureNet::C3 + (NatureNet::G2::FEATS + 1) * NatureNet::C2 +
                                    257 * NatureNet::C1);
  return (int64_t)W_SPLITS_MAX * ((NipsNet::G2::FEATS + 1) * NipsNet::C2 + 257 * NipsNet::C1);
}
int fc_splits_max() { return FC_SPLITS_MAX; }

}  // namespace paac
