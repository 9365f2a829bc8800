// Shared by net_fwd.hip and net_bwd.hip (two translation units so that the template instantiations compile in
// parallel): network geometries, GEMM argument packing, launch configuration tables.
#pragma once
#include <stdlib.h>

#include "dmm.h"
#include "heads.h"

namespace paac {

// ---------------------------------------------------------------------------------------------
// Compile-time network descriptions.
struct NatureNet {
  static constexpr int NCONV = 3, C1 = 32, C2 = 64, C3 = 64, H = 512, FLAT = 3136;
  static constexpr bool FAMILY = true;      // conv 8x8 / 4 -> conv 4x4 / 2 [-> conv 3x3 / 1]: the shapes the fused kernels and
                                            // the MFMA data-gradient forms are written for
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
  static constexpr bool FAMILY = true;
  using G1 = Geom<84, 84, 4, 20, 20, 4, 0, 0, 8, 8>;
  using G2 = Geom<20, 20, 16, 9, 9, 2, 0, 0, 4, 4>;
  using G3 = Geom<9, 9, 32, 7, 7, 1, 0, 0, 3, 3>;      // unused
  using GFC = Geom<1, 1, 2592, 1, 1, 1, 0, 0, 1, 1>;
  using GFCH = Geom<1, 1, 256, 1, 1, 1, 0, 0, 1, 1>;
  using G3D = Geom<7, 7, 32, 9, 9, 1, 2, 2, 3, 3>;     // unused
  using G2D = Geom<9, 9, 32, 10, 10, 1, 1, 1, 2, 2>;
};

// A user architecture (reference networks.py:117-120, README "new architectures"): the same trunk family -- conv 8x8 / 4,
// conv 4x4 / 2 [, conv 3x3 / 1], fc -- with the user's filter counts and fc width, compiled into its own library
// (paac_amd/build.py: build_user_arch; -DPAAC_USER_ARCH -DPAAC_USER_NCONV=.. -DPAAC_USER_C1=.. ...).  It takes the place
// of the NIPS geometry in the two-way dispatch: such a library serves PAAC_ARCH_NATURE and PAAC_ARCH_USER.
#ifdef PAAC_USER_ARCH
// per-layer kernel size / stride: the family's unless the build says otherwise (paac_amd/build.py: build_user_arch)
#ifndef PAAC_USER_K1
#define PAAC_USER_K1 8
#define PAAC_USER_S1 4
#define PAAC_USER_K2 4
#define PAAC_USER_S2 2
#define PAAC_USER_K3 3
#define PAAC_USER_S3 1
#endif
struct UserNet {
  static constexpr int NCONV = PAAC_USER_NCONV, C1 = PAAC_USER_C1, C2 = PAAC_USER_C2;
  static constexpr int C3 = (PAAC_USER_NCONV == 3) ? PAAC_USER_C3 : PAAC_USER_C2, H = PAAC_USER_H;
  static constexpr int K1 = PAAC_USER_K1, S1 = PAAC_USER_S1, K2 = PAAC_USER_K2, S2 = PAAC_USER_S2;
  static constexpr int K3 = (PAAC_USER_NCONV == 3) ? PAAC_USER_K3 : 1, S3 = (PAAC_USER_NCONV == 3) ? PAAC_USER_S3 : 1;
  // VALID convolutions over the 84 x 84 x 4 input (networks.py:12-21)
  static constexpr int O1 = (84 - K1) / S1 + 1, O2 = (O1 - K2) / S2 + 1, O3 = (PAAC_USER_NCONV == 3) ? (O2 - K3) / S3 + 1 : O2;
  static constexpr int FLAT = (PAAC_USER_NCONV == 3) ? O3 * O3 * C3 : O2 * O2 * C2;
  static constexpr bool FAMILY = K1 == 8 && S1 == 4 && K2 == 4 && S2 == 2 && (PAAC_USER_NCONV == 2 || (K3 == 3 && S3 == 1));
  static_assert(NCONV == 2 || NCONV == 3, "two or three conv layers");
  static_assert(C1 % 16 == 0 && C2 % 16 == 0 && C3 % 16 == 0 && C1 >= 16 && C2 >= 16 && C3 >= 16, "filter counts: multiples of 16");
  static_assert(H % 256 == 0 && H >= 256, "fc width: a multiple of 256");
  static_assert((K1 * 4) % 16 == 0, "conv1: kernel width x 4 input channels must be a multiple of 16 (one MFMA K group never "
                                    "straddles two kernel rows): sizes 4, 8, 12, 16");
  static_assert(O1 >= 1 && O2 >= 1 && O3 >= 1 && K1 >= 1 && K2 >= 1 && K3 >= 1 && S1 >= 1 && S2 >= 1 && S3 >= 1, "layer shapes");
  using G1 = Geom<84, 84, 4, O1, O1, S1, 0, 0, K1, K1>;
  using G2 = Geom<O1, O1, C1, O2, O2, S2, 0, 0, K2, K2>;
  using G3 = Geom<O2, O2, C2, O3, O3, S3, 0, 0, K3, K3>;
  using GFC = Geom<1, 1, FLAT, 1, 1, 1, 0, 0, 1, 1>;
  using GFCH = Geom<1, 1, H, 1, 1, 1, 0, 0, 1, 1>;
  // the family's MFMA data-gradient geometries (conv3: full correlation; conv2: one output-parity class); other layer
  // shapes take the direct data-gradient kernel (net_bwd.hip: conv_dgrad_direct_kernel)
  using G3D = Geom<7, 7, C3, 9, 9, 1, 2, 2, 3, 3>;
  using G2D = Geom<9, 9, C2, 10, 10, 1, 1, 1, 2, 2>;
};
using OtherNet = UserNet;
#else
using OtherNet = NipsNet;
#endif

constexpr int W_SPLITS_MAX = 64;

#ifdef PAAC_DMM_STAMPS
extern unsigned long long* g_stamps;   // diagnostic build: the `which`-th dmm launch after the call is stamped
extern int g_stamp_which, g_stamp_calls;
#endif

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

static GemmArgs make_args(const void* A, size_t a_bytes, const float* B, size_t b_bytes, float* out, const float* aux,
                          int M, int N, int K, int ldb, int ldo) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.out = out; g.aux = aux;
  g.a_bytes = (unsigned)(a_bytes < 0x7FFFFFF0ull ? a_bytes : 0x7FFFFFF0ull);
  g.b_bytes = (unsigned)(b_bytes < 0x7FFFFFF0ull ? b_bytes : 0x7FFFFFF0ull);
  g.M = M; g.N = N; g.K = K; g.ldb = ldb; g.ldo = ldo;
  g.slab_rows = M;
#ifdef PAAC_DMM_STAMPS
  g.stamps = (g_stamp_calls++ == g_stamp_which) ? g_stamps : nullptr;
#endif
  return g;
}

// blockIdx.z split of K so that the launch has about `target_waves` waves.
static int pick_ksplit(long tiles, int wk, int ngroups, int max_split, int target_waves = 1024) {
  long s = (target_waves + tiles * wk - 1) / (tiles * wk);
  if (s > max_split) s = max_split;
  if (s > ngroups / wk) s = ngroups / wk;
  if (s < 1) s = 1;
  const int per = (int)((ngroups + s * wk - 1) / (s * wk));     // groups per (z, wk) part
  s = (ngroups + (long)per * wk - 1) / ((long)per * wk);         // drop empty tail slabs
  return (int)s;
}

// ---------------------------------------------------------------------------------------------
// Launch configurations.  Each GEMM family has a small table of (tiles per wave, waves per workgroup, K split over
// waves, prefetch depth) instantiations; a Tune record (per op and batch class, set from measured sweeps --
// tools/tune_gemm.py -- or left at cfg = -1 for the size heuristic) picks one, plus the blockIdx.z K split and the
// XCD-tied grid dimension.
//                      id TM NWM WK PF
#define PAAC_FWD_CFGS(X) X(0, 1, 1, 8, 5) X(1, 1, 1, 4, 5) X(2, 2, 1, 8, 3) X(3, 2, 1, 4, 3) X(4, 2, 2, 2, 2) \
                         X(5, 2, 4, 1, 2) X(6, 1, 2, 4, 4) X(7, 2, 2, 4, 2) X(8, 2, 2, 1, 3) X(9, 1, 4, 1, 4) \
                         X(10, 4, 1, 2, 2) X(11, 4, 2, 1, 2) X(12, 2, 1, 2, 3)
#define PAAC_DGRAD_CFGS(X) X(0, 1, 1, 8, 4) X(1, 2, 1, 4, 4) X(2, 2, 2, 2, 3) X(3, 2, 4, 1, 2) X(4, 1, 1, 4, 4) \
                           X(5, 2, 1, 8, 3) X(6, 1, 2, 4, 4) X(7, 2, 2, 1, 3) X(8, 1, 4, 1, 4) X(9, 4, 1, 2, 2)  \
                           X(10, 4, 2, 1, 2) X(11, 2, 1, 2, 3)
//                        id TM WK PF
#define PAAC_WGRAD_CFGS(X) X(0, 4, 4, 2) X(1, 4, 2, 2) X(2, 4, 8, 2) X(3, 4, 4, 3) X(4, 2, 4, 3) X(5, 2, 8, 3) X(6, 4, 1, 3) \
                           X(7, 4, 1, 4) X(8, 4, 2, 3)
// the entries also instantiated on the split-bf16 path (ids + kSplitBf16); an id outside falls back to its fp32 form
#define PAAC_FWD_SPLIT_CFGS(X) X(1, 1, 1, 4, 5) X(4, 2, 2, 2, 2) X(7, 2, 2, 4, 2) X(10, 4, 1, 2, 2) X(11, 4, 2, 1, 2) \
                               X(12, 2, 1, 2, 3)
#define PAAC_DGRAD_SPLIT_CFGS(X) X(1, 2, 1, 4, 4) X(5, 2, 1, 8, 3) X(9, 4, 1, 2, 2) X(10, 4, 2, 1, 2) X(11, 2, 1, 2, 3)
#define PAAC_WGRAD_SPLIT_CFGS(X) X(0, 4, 4, 2) X(1, 4, 2, 2) X(3, 4, 4, 3)
constexpr int kFwdCfgs = 13, kDgradCfgs = 12, kWgradCfgs = 9;
constexpr int kExactBf16 = 100;   // cfg ids from here on: the same table entry on the exact-bf16 path (u8 operand only)
constexpr int kNarrow = 300;       // ... with half-width N tiles (32 columns per wave): more workgroups for the small acting batch
#define PAAC_FWD_NARROW_CFGS(X) X(0, 1, 1, 8, 5) X(1, 1, 1, 4, 5) X(2, 2, 1, 8, 3) X(3, 2, 1, 4, 3) X(6, 1, 2, 4, 4)
constexpr int kSplitBf16 = 200;   // ... on the six-product split-bf16 path (fp32 operands; dmm.h: XB = 2)
constexpr int split_pf(int pf) { return pf > 2 ? 2 : pf; }   // a stage is two K groups there: shallower ring


}  // namespace paac
