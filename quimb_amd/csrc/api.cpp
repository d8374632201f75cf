// api.cpp -- host side of the C-ABI (include/quimb_amd.h): plan validation,
// tile / vector-width / split-K selection, bundle construction for the tiled
// permute, argument packing.  No device code here; kernels live in gett.hip and
// elementwise.hip.  Nothing in this file allocates device memory or synchronises.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/quimb_amd.h"
#include "ew_args.h"
#include "gett_args.h"
#include "program.h"

static const int kEsize[4] = {4, 8, 8, 16};

extern "C" int qamd_abi_version(void) { return QAMD_ABI_VERSION; }
extern "C" const char* qamd_build_info(void) {
  return "quimb_amd gfx950 (CDNA4) HIP backend; MFMA f32/f64 16x16x4 GETT; ABI 1";
}

// ---------------------------------------------------------------------------
// pairwise plan
// ---------------------------------------------------------------------------
static bool prod_ok(const int64_t* d, int n, int64_t& out) {
  int64_t p = 1;
  for (int i = 0; i < n; ++i) {
    if (d[i] <= 0) return false;
    p *= d[i];
    if (p >= (1ll << 31)) return false;
  }
  out = p;
  return true;
}

struct PairDims { int64_t B, M, N, K; };

static int pair_dims(const qamd_pair_plan* p, PairDims& d) {
  if (!p) return QAMD_EINVAL;
  if (p->nb < 0 || p->nm < 0 || p->nn < 0 || p->nk < 0) return QAMD_EINVAL;
  if (p->nb > QAMD_MAX_GROUPS || p->nm > QAMD_MAX_GROUPS || p->nn > QAMD_MAX_GROUPS || p->nk > QAMD_MAX_GROUPS)
    return QAMD_EINVAL;
  if (!prod_ok(p->dim_b, p->nb, d.B) || !prod_ok(p->dim_m, p->nm, d.M) || !prod_ok(p->dim_n, p->nn, d.N) ||
      !prod_ok(p->dim_k, p->nk, d.K))
    return QAMD_EINVAL;
  return QAMD_OK;
}

static int pick_vec(int64_t inner_dim, int64_t align_bytes, int esize, const std::vector<int64_t>& others) {
  for (int v = 4; v >= 2; v >>= 1) {
    if (inner_dim % v) continue;
    if (align_bytes % std::min<int64_t>((int64_t)v * esize, 16)) continue;   // no access is wider than 16 bytes
    bool ok = true;
    for (int64_t s : others)
      if (s % v) { ok = false; break; }
    if (ok) return v;
  }
  return 1;
}

static const int kNumCU = 256;
static const int kNumTileCfg = 8;
static const int kFastCfg = 6;    // gettf.hip: full 128x128x16 tiles, compile-time loader
static const int kFastCfgN = 7;   // gettf.hip: full 128x64x16 tiles
static const int kTileBM[kNumTileCfg] = {128, 64, 256, 256, 128, 128, 128, 128};
static const int kTileBN[kNumTileCfg] = {128, 64, 48, 16, 32, 128, 128, 64};
static const int kTileBK[kNumTileCfg] = {16, 16, 16, 16, 16, 32, 16, 16};

// full-tile fast path: every tile whole, 4-element vector loads on both operands, enough
// tiles to fill the chip without split-K
static bool fast_tile_ok(const qamd_pair_plan* p, const PairDims& d, int cfg) {
  const int bn = kTileBN[cfg];
  return d.M % 128 == 0 && d.N % bn == 0 && d.K % 16 == 0 && p->vec_a >= 2 && p->vec_b >= 2;
}
static const int kBK = 32;  // k-offset tables are padded to the largest k-tile

// size of the trailing block of N groups that is contiguous (stride-1 run) in C
static int64_t n_inner_block(const qamd_pair_plan* p) {
  int64_t d = 1;
  for (int g = p->nn - 1; g >= 0; --g) {
    if (p->sc_n[g] != d) break;
    d *= p->dim_n[g];
  }
  return d;
}

static int stream_lds_bytes(int64_t K, int64_t N, int es) {
  int64_t kpad = ((K + 3) / 4) * 4;
  int64_t npad = ((N + 15) / 16) * 16;
  int64_t ldw = npad + ((48 - npad % 32) % 32);
  return (int)((2 * npad + kpad) * 8 + kpad * ldw * es);
}

static int64_t c_extent(const qamd_pair_plan* p);

// "few rows x one long vector" (dotm.hip): M*N <= 32 with one of them 1, K one stride-1 group in
// both operands.  Fills the row offsets; returns false if the shape does not qualify.
static bool dot_rows(const qamd_pair_plan* p, const PairDims& d, DotArgs& a) {
  if (p->nb != 0 || p->nk != 1 || p->sa_k[0] != 1 || p->sb_k[0] != 1) return false;
  const bool rows_a = d.N == 1 && d.M <= 32, rows_b = d.M == 1 && d.N <= 32 && !rows_a;
  if (!rows_a && !rows_b) return false;
  const int ng = rows_a ? p->nm : p->nn;
  const int64_t* dims = rows_a ? p->dim_m : p->dim_n;
  const int64_t* str = rows_a ? p->sa_m : p->sb_n;
  a.S = (int32_t)(rows_a ? d.M : d.N);
  a.K = d.K;
  for (int s = 0; s < a.S; ++s) {
    int64_t idx = s, off = 0;
    for (int g = ng - 1; g >= 0; --g) { off += (idx % dims[g]) * str[g]; idx /= dims[g]; }
    a.row_off[s] = off;
  }
  return true;
}

// ---- gemmk.hip eligibility and tile choice --------------------------------------------------------
// Both operands carry their free bundle innermost in 4-element vectors, ONE K group (K % 8 == 0),
// non-negative strides, free-bundle offsets of the whole operand below 4 GiB.  The operands swap roles when
// C's stride-1 index lives in the M bundle (lanes of an MFMA result run along the B operand's index).
static bool gemmk_operand_ok(int ng, const int64_t* dims, const int64_t* strides, int64_t sk) {
  if (ng < 1 || strides[ng - 1] != 1 || dims[ng - 1] % 4 || sk <= 0) return false;
  // the kernel addresses a lane's 16-byte piece as (wave-uniform 64-bit base that advances along k) + (32-bit byte
  // offset of its free-bundle index, absolute within the operand, + up to 15 k rows): everything must stay below 4 GiB
  int64_t span = 0;
  for (int g = 0; g < ng; ++g) {
    if (strides[g] < 0) return false;
    if (g < ng - 1 && strides[g] % 4) return false;
    span += (dims[g] - 1) * strides[g];
  }
  return (span + 16 * sk + 4) * 4 < (1ll << 32);
}

static double gemmk_model(int64_t M, int64_t N, int64_t B, int ta, int tb, int64_t* tiles_out) {
  const int64_t tiles = ((M + 64 * ta - 1) / (64 * ta)) * ((N + 64 * tb - 1) / (64 * tb)) * B;
  // rounds of one tile per CU: two co-resident workgroups of a small tile time-share the CU's matrix pipes, so
  // occupancy smooths the tail but does not add throughput (measured, profiles/r03_gemm_pow6.txt: 2048 x 2560 x 512
  // runs 82 TFLOP/s on 220 tiles of 192 x 128 against 75 on 160 tiles of 128 x 256)
  const int64_t rounds = (tiles + kNumCU - 1) / kNumCU;
  static const double eff[17] = {0, 0, 0, 0, 0.86, 0, 0.90, 0, 0.93, 0.95, 0, 0, 0.96, 0, 0, 0, 0.97};
  if (tiles_out) *tiles_out = tiles;
  return (double)rounds * ta * tb / eff[ta * tb];
}

// ``pin``: 0 = the model's tile, else 10 ta + tb (a caller's explicit pin: qamd_pair_plan.kernel = -5 with tile_cfg = 16 ta + tb
// on input; the grid-size floor is waived for a pinned tile)
static bool gemmk_config(const qamd_pair_plan* p, const PairDims& d, int64_t align_a, int64_t align_b, int64_t align_c,
                         int pin, int& ta, int& tb) {
  if (p->dtype != QAMD_F32 || p->nk != 1 || p->a_kcontig || p->b_kcontig || p->vec_a < 4 || p->vec_b < 4) return false;
  if (d.K % 8 || d.K < 64 || d.M < 128 || d.N < 128 || d.M % 4 || d.N % 4) return false;
  if ((align_a % 16) || (align_b % 16) || (align_c % 4)) return false;
  if (!gemmk_operand_ok(p->nm, p->dim_m, p->sa_m, p->sa_k[0]) ||
      !gemmk_operand_ok(p->nn, p->dim_n, p->sb_n, p->sb_k[0]))
    return false;
  for (int i = 0; i < p->nb; ++i)
    if (p->sa_b[i] % 4 || p->sb_b[i] % 4) return false;
  const bool swap = !p->c_ncontig;
  const int64_t M = swap ? d.N : d.M, N = swap ? d.M : d.N;
  double best = 0;
  ta = tb = 0;
  for (int a = 2; a <= 4; ++a)
    for (int b = 2; b <= 4; ++b) {
      if (pin && pin != 10 * a + b) continue;
      int64_t tiles = 0;
      const double t = gemmk_model(M, N, d.B, a, b, &tiles);
      if (!ta || t < best) { best = t; ta = a; tb = b; }
    }
  if (!ta) return false;
  int64_t tiles = 0;
  gemmk_model(M, N, d.B, ta, tb, &tiles);
  // under-filled grids keep the split-K kernels
  return tiles >= 96 || pin != 0;
}

// kernel 7 (gemmh.hip, defined beside its launcher below)
static bool gemmh_config(const qamd_pair_plan* p, const PairDims& d, int pin, bool all_layouts, int& ta, int& tb);
static int64_t gemmh_workspace_bytes(const qamd_pair_plan* p, const PairDims& d);

// ---- gemmd.hip eligibility and tile choice (fp64, LDS-DMA ring, either operand layout) -----------------------------
// An operand is FREE-contiguous (its innermost free group is stride-1: granules of two consecutive free elements) or
// K-contiguous (its innermost K group is stride-1: granules of two consecutive k).  Either way the granule group has
// an even size and every other stride of the operand is even (16-byte granules stay aligned); K % 16 == 0.
static bool gemmd_operand_ok(bool kcontig, int nfree, const int64_t* dfree, const int64_t* sfree, int nk,
                             const int64_t* dk, const int64_t* sk, int nb, const int64_t* sb) {
  if (nfree < 1 || nk < 1) return false;
  const int gi = kcontig ? nk - 1 : nfree - 1;
  const int64_t* gd = kcontig ? dk : dfree;
  const int64_t* gs = kcontig ? sk : sfree;
  if (gs[gi] != 1 || gd[gi] % 2) return false;
  for (int g = 0; g < nfree; ++g)
    if (sfree[g] < 0 || (!(g == nfree - 1 && !kcontig) && sfree[g] % 2)) return false;
  for (int g = 0; g < nk; ++g)
    if (sk[g] < 0 || (!(g == nk - 1 && kcontig) && sk[g] % 2)) return false;
  for (int g = 0; g < nb; ++g)
    if (sb[g] % 2) return false;
  return true;
}

// cost of a (32 ta) x (64 tb) tiling with ``split`` k-slabs: rounds of one workgroup per CU x MFMAs per k-step of a
// wave, plus what the slab reduction moves; narrow wave tiles read more LDS per MFMA, which fp64's 64-cycle MFMAs
// mostly forgive
static double gemmd_model(int64_t M, int64_t N, int64_t K, int64_t B, int ta, int tb, int split) {
  const int64_t tiles = ((M + 32 * ta - 1) / (32 * ta)) * ((N + 64 * tb - 1) / (64 * tb)) * B * split;
  const int64_t rounds = (tiles + kNumCU - 1) / kNumCU;
  const int w = ta * tb;
  const double eff = w >= 10 ? 0.92 : (w >= 8 ? 0.9 : (w >= 6 ? 0.87 : (w >= 4 ? 0.82 : 0.75)));
  const double ksteps = (double)((K / 16 + split - 1) / split) + 3.0;          // + prologue / epilogue of a tile
  double t = (double)rounds * w * ksteps / eff;
  if (split > 1) t += 0.04 * (double)(split + 1) * (double)(M * N * B) / (256.0 * 64.0);   // slab write + reduce, in MFMA units
  return t;
}

// A k-contiguous operand is gathered row by row, so the ORDER of its free groups is the kernel's to choose: put the group
// that is most contiguous in C innermost.  The lanes of the result tile then run along C's stride-1 index and a store
// instruction writes 128-byte runs (the chi = 512 matvec's first product writes its result as [S1, p, S2, B, a] with the
// free bundle of L = (a, p): 42 MB of 8-byte stores 4 KB apart before this, 157 us; 16 consecutive a per store after).
static void gemmd_order_for_stores(qamd_pair_plan* p) {
  auto to_back = [](int n, int64_t* dim, int64_t* s_op, int64_t* s_c) {
    int best = n - 1;
    for (int g = 0; g < n; ++g)
      if (s_c[g] > 0 && (s_c[best] <= 0 || s_c[g] < s_c[best])) best = g;
    for (int g = best; g + 1 < n; ++g) {
      std::swap(dim[g], dim[g + 1]);
      std::swap(s_op[g], s_op[g + 1]);
      std::swap(s_c[g], s_c[g + 1]);
    }
  };
  if (p->a_kcontig && p->nm > 1) to_back(p->nm, p->dim_m, p->sa_m, p->sc_m);
  if (p->b_kcontig && p->nn > 1) to_back(p->nn, p->dim_n, p->sb_n, p->sc_n);
  const bool n1 = p->nn > 0 && p->sc_n[p->nn - 1] == 1;
  const bool m1 = p->nm > 0 && p->sc_m[p->nm - 1] == 1;
  p->c_ncontig = (n1 || !m1) ? 1 : 0;
}

// ``pin``: 0 = the model's tile, else 10 ta + tb (qamd_pair_plan.kernel = -6 with tile_cfg = 16 ta + tb on input) -- and take
// every shape the kernel can run
static bool gemmd_config(const qamd_pair_plan* p, const PairDims& d, int64_t align_a, int64_t align_b, int pin, int& ta,
                         int& tb, int& split) {
  if (p->dtype != QAMD_F64) return false;
  const bool te = pin != 0;
  if (d.K % 16 || d.K < (te ? 16 : 64) || d.M < (te ? 2 : 64) || d.N < (te ? 2 : 64) || d.M % 2 || d.N % 2) return false;
  if (p->nk < 1 || p->dim_k[p->nk - 1] % 16) return false;     // a k-tile (16 rows) stays inside the innermost K group
  if ((align_a % 16) || (align_b % 16)) return false;
  if (!gemmd_operand_ok(p->a_kcontig, p->nm, p->dim_m, p->sa_m, p->nk, p->dim_k, p->sa_k, p->nb, p->sa_b) ||
      !gemmd_operand_ok(p->b_kcontig, p->nn, p->dim_n, p->sb_n, p->nk, p->dim_k, p->sb_k, p->nb, p->sb_b))
    return false;
  const bool compact = c_extent(p) == d.B * d.M * d.N;   // the slab reduction needs a compact C
  double best = 0;
  ta = tb = 0;
  split = 1;
  for (int a = 2; a <= 5; ++a)
    for (int b = 1; b <= 2; ++b) {
      if (pin && pin != 10 * a + b) continue;
      for (int s = 1; s <= 16; s *= 2) {
        if (s > 1 && (!compact || d.K / 16 / s < 8)) break;
        const double t = gemmd_model(d.M, d.N, d.K, d.B, a, b, s);
        if (!ta || t < best) { best = t; ta = a; tb = b; split = s; }
      }
    }
  if (!ta) return false;
  const int64_t tiles = ((d.M + 32 * ta - 1) / (32 * ta)) * ((d.N + 64 * tb - 1) / (64 * tb)) * d.B * split;
  return tiles >= 64 || te;       // tiny grids keep the generic kernels
}

// Z-mode streaming with "breaks": the innermost M group (length l_in, C stride d_in) is not a whole number of chunks of
// ``ch`` = 16 ev rows, but (i) 4 l_in >= ch, so a chunk meets a handful of its pieces at most, (ii) A is contiguous across
// a suffix of the M groups whose product R is a multiple of ch (chunks then never straddle a run of A), (iii) every piece
// end falls on a 16-byte vector boundary of C (chunk starts are multiples of gcd(ch, l_in) rows).  Returns R (elements) or 0.
// break mode of the streaming kernel's Z stores: entries (8 bytes each, in LDS) of the per-workgroup table of group starts --
// the pieces of the innermost M group a workgroup's rows (4 waves x chunks_per_wave chunks of 16 V rows) can touch, + what
// the last chunk looks up.  ONE formula for the plan-time LDS check (finalize) and the launch (fill_stream_args): the table
// grows with M (chunks_per_wave), so a fixed allowance would let a very large M with a short group pass finalize and fail
// at launch.
static uint32_t stream_chunks_per_wave(int64_t M, int V) {
  const uint64_t chunks = (uint64_t)((M + 16 * V - 1) / (16 * V));
  const uint32_t target_waves = 256 * 4 * 3;
  const uint64_t cpw = (chunks + target_waves - 1) / target_waves;
  return (uint32_t)(cpw < 1 ? 1 : cpw);
}
static uint32_t stream_zb_groups(int64_t M, int V, int64_t l_in) {
  return (uint32_t)((uint64_t)stream_chunks_per_wave(M, V) * 4 * 16 * V / (uint64_t)l_in + 3 + 16 * V / l_in + 2);
}

static int64_t z_break_run(const qamd_pair_plan* p, int ev, int64_t d_in) {
  const int64_t ch = 16 * ev;
  if (p->nm < 2 || p->sa_m[p->nm - 1] != 1) return 0;
  const int64_t l_in = p->dim_m[p->nm - 1];
  int64_t g = ch, r = l_in;
  while (r) { const int64_t t = g % r; g = r; r = t; }          // gcd(ch, l_in)
  if (l_in % ch == 0 || 4 * l_in < ch || (l_in * d_in) % ev || (g * d_in) % ev) return 0;
  int64_t run = l_in;
  for (int g = p->nm - 2; g >= 0; --g) {
    if (p->sa_m[g] != run) break;          // A no longer contiguous across this group
    run *= p->dim_m[g];
    if (run % ch == 0) return run;
  }
  return 0;
}

extern "C" int qamd_pair_plan_finalize(qamd_pair_plan* p, int64_t align_a, int64_t align_b, int64_t align_c) {
  PairDims d;
  int rc = pair_dims(p, d);
  if (rc) return rc;
  // the caller's explicit pins (the ONLY way to steer the choice: nothing below reads the environment):
  //   kernel  0 auto | -1 tiled GETT | -2 auto without the MFMA GEMM kernels | -5 / -6 gemmk / gemmd with the tile named by
  //   tile_cfg = 16 ta + tb (where the kernel can run the shape at all; otherwise the automatic choice) | -7 split products on
  //   the f16 matrix pipe (gemmh.hip) for the fp32 k-outer joins large enough to pay for the split pass, auto elsewhere | -8 the
  //   same for every fp32 GEMM-shaped pair without a batch bundle, whatever its operand layout
  const int pin_kernel = p->kernel;
  int pin_gemm = 0;
  if (pin_kernel == -5 || pin_kernel == -6 || pin_kernel == -7 || pin_kernel == -8) {
    if (p->tile_cfg > 0) pin_gemm = 10 * (p->tile_cfg / 16) + p->tile_cfg % 16;
    p->tile_cfg = -1;
    p->kernel = 0;
  } else if (pin_kernel == -2) {
    p->kernel = 0;
  } else if (pin_kernel < -1 || pin_kernel > 0) {
    p->kernel = 0;      // (a finalized plan handed in again: its output codes are not pins)
  }
  if (p->dtype != QAMD_F32 && p->dtype != QAMD_F64) return QAMD_EUNSUPPORTED;
  const int es = kEsize[p->dtype];

  // ---- operand A: which bundle owns the stride-1 index --------------------
  {
    bool m1 = p->nm > 0 && p->sa_m[p->nm - 1] == 1;
    bool k1 = p->nk > 0 && p->sa_k[p->nk - 1] == 1;
    p->a_kcontig = (!m1 && k1) ? 1 : 0;
    p->vec_a = 1;
    if (m1 || k1) {
      std::vector<int64_t> others;
      for (int i = 0; i < p->nb; ++i) others.push_back(p->sa_b[i]);
      for (int i = 0; i < p->nm; ++i)
        if (!(m1 && !p->a_kcontig && i == p->nm - 1)) others.push_back(p->sa_m[i]);
      for (int i = 0; i < p->nk; ++i)
        if (!(p->a_kcontig && i == p->nk - 1)) others.push_back(p->sa_k[i]);
      int64_t inner = p->a_kcontig ? p->dim_k[p->nk - 1] : p->dim_m[p->nm - 1];
      p->vec_a = pick_vec(inner, align_a, es, others);
    }
  }
  {
    bool n1 = p->nn > 0 && p->sb_n[p->nn - 1] == 1;
    bool k1 = p->nk > 0 && p->sb_k[p->nk - 1] == 1;
    p->b_kcontig = (!n1 && k1) ? 1 : 0;
    p->vec_b = 1;
    if (n1 || k1) {
      std::vector<int64_t> others;
      for (int i = 0; i < p->nb; ++i) others.push_back(p->sb_b[i]);
      for (int i = 0; i < p->nn; ++i)
        if (!(n1 && !p->b_kcontig && i == p->nn - 1)) others.push_back(p->sb_n[i]);
      for (int i = 0; i < p->nk; ++i)
        if (!(p->b_kcontig && i == p->nk - 1)) others.push_back(p->sb_k[i]);
      int64_t inner = p->b_kcontig ? p->dim_k[p->nk - 1] : p->dim_n[p->nn - 1];
      p->vec_b = pick_vec(inner, align_b, es, others);
    }
  }
  {
    bool n1 = p->nn > 0 && p->sc_n[p->nn - 1] == 1;
    bool m1 = p->nm > 0 && p->sc_m[p->nm - 1] == 1;
    p->c_ncontig = (n1 || !m1) ? 1 : 0;
  }

  // ---- streaming kernel eligibility (big tensor x small tensor) ---------------
  // kernel 1 (X): A and C both have their stride-1 index in the same innermost M group.
  // kernel 2 (Z): A as above, C = [.., m_in, n_in] with n_in stride-1 directly inside
  //               the innermost M group (death-ordered executor layouts).
  // Both: the whole small operand fits in LDS, no batch bundle.
  {
    const bool base_ok = p->nb == 0 && p->nm >= 1 && p->sa_m[p->nm - 1] == 1 && !p->a_kcontig &&
                         d.N <= 64 && d.M >= 4096 && d.K <= 4096;
    const int ev = 16 / es;  // elements per 16-byte vector
    int kern = 0, vc = 1;
    const int64_t d_in = n_inner_block(p);
    // the innermost M group is a whole number of chunks -- or (round 5) it is not, but it is at least a chunk long, A runs
    // on contiguously across it and 16-byte stores never cross its end (the last site of row 4 of a boundary sweep: runs of
    // 216 open-leg values, until now the generic tiled kernel at 0.13 of the HBM roof): a chunk then breaks at most once in C
    const int64_t l_in = p->nm >= 1 ? p->dim_m[p->nm - 1] : 0;
    const bool z_chunks = l_in % (16 * ev) == 0 || z_break_run(p, ev, d_in) > 0;
    if (base_ok && p->nn >= 1 && d_in > 1 && p->sc_m[p->nm - 1] == d_in && p->vec_a >= ev &&
        align_c % 16 == 0 && z_chunks) {
      bool ok = true;
      for (int i = 0; i + 1 < p->nm; ++i) ok = ok && (p->sc_m[i] % ev == 0);
      for (int i = 0; i < p->nn; ++i) ok = ok && (p->sc_n[i] < d_in || p->sc_n[i] % ev == 0);
      // (+ the table of group starts in break mode, sized exactly as the launch will size it)
      const int64_t zb_bytes = l_in % (16 * ev) ? 8ll * stream_zb_groups(d.M, ev, l_in) : 0;
      ok = ok && (stream_lds_bytes(d.K, d.N, es) + 4 * (d.N + d_in) * 16 * ev * es + zb_bytes <= 80 * 1024);
      if (ok) { kern = 2; vc = ev; }
    }
    if (!kern && base_ok && p->sc_m[p->nm - 1] == 1 && stream_lds_bytes(d.K, d.N, es) <= 64 * 1024) {
      std::vector<int64_t> others;
      for (int i = 0; i + 1 < p->nm; ++i) others.push_back(p->sc_m[i]);
      for (int i = 0; i < p->nn; ++i) others.push_back(p->sc_n[i]);
      vc = pick_vec(p->dim_m[p->nm - 1], align_c, es, others);
      vc = std::min(vc, (int)p->vec_a);
      if (vc > ev) vc = ev;
      kern = 1;
    }
    p->vec_c = vc;
    // kernel 4: reduction-shaped (norms, projections onto a few vectors): streaming multi-dot
    if (!kern && d.K >= 32768 && align_a % 16 == 0 && align_b % 16 == 0 && c_extent(p) == d.M * d.N) {
      DotArgs da;
      if (dot_rows(p, d, da)) {
        const int ev = 16 / es;
        bool ok = true;
        for (int s = 0; s < da.S; ++s) ok = ok && (da.row_off[s] % ev == 0);
        if (ok) kern = 4;
      }
    }
    if (p->kernel == -1) p->kernel = 0;  // caller forces the tiled kernel
    else p->kernel = kern;
    // kernel 5 (gemmk.hip): GEMM-shaped, both operands "k-outer" (free bundle stride-1), fp32
    if (p->kernel == 0 && kern == 0 && p->tile_cfg < 0 && pin_kernel != -2) {
      int ta = 0, tb = 0;
      // kernel 7 (gemmh.hip): opt-in, the same shapes as kernel 5 with B == 1
      // (-7 without a pinned tile: only where the DEFAULT choice would have been the chain kernel gemmk -- the claim of that mode is
      // "never less accurate than the default")
      int ka = 0, kb = 0;
      const bool chain_default = pin_kernel != -7 || pin_gemm != 0 || gemmk_config(p, d, align_a, align_b, align_c, 0, ka, kb);
      if ((pin_kernel == -7 || pin_kernel == -8) && chain_default && gemmh_config(p, d, pin_gemm, pin_kernel == -8, ta, tb)) {
        p->kernel = 7;
        p->tile_cfg = 16 * ta + tb;
        p->split_k = 1;
        return QAMD_OK;
      }
      if (pin_kernel != -6 && gemmk_config(p, d, align_a, align_b, align_c, pin_kernel == -5 ? pin_gemm : 0, ta, tb)) {
        p->kernel = 5;
        p->tile_cfg = 16 * ta + tb;
        p->split_k = 1;
        return QAMD_OK;
      }
      // kernel 6 (gemmd.hip): fp64, GEMM-shaped, either operand layout
      int split = 1;
      if (pin_kernel != -5 && gemmd_config(p, d, align_a, align_b, pin_kernel == -6 ? pin_gemm : 0, ta, tb, split)) {
        gemmd_order_for_stores(p);
        p->kernel = 6;
        p->tile_cfg = 16 * ta + tb;
        if (p->split_k < 1 || c_extent(p) != d.B * d.M * d.N) p->split_k = split;     // (a caller's split is kept)
        return QAMD_OK;
      }
    }
  }
  if (p->kernel == 4) {
    // one slab of partial sums per workgroup; every workgroup streams >= 2 x 256 16-byte vectors per row
    const int64_t nvec = d.K / (16 / es);
    p->split_k = (int32_t)std::max<int64_t>(1, std::min<int64_t>(1024, nvec / (256 * 2)));
    p->tile_cfg = 1;
    return QAMD_OK;
  }

  // ---- tile shape -----------------------------------------------------------
  if (p->tile_cfg < 0 || p->tile_cfg >= kNumTileCfg) {
    int cfg;
    if (d.N <= 16) cfg = 3;
    else if (d.N <= 32) cfg = 4;
    else if (d.N <= 48) cfg = 2;
    else {
      // measured (profiles/r01_microbench.txt): the 64x64 tile (more workgroups in flight)
      // beats 128x128x16 and 128x128x32 at every size from 1024^3 to 8192^3, f32 and f64;
      // the compile-time 128x128 kernel (gettf.hip) beats both wherever its preconditions hold
      const int want = -1;     // (a caller pins a shape with tile_cfg = 6 / 7 on input, or 1 to keep the 64 x 64 tile)
      cfg = 1;
      {
        // 128x128 once it fills the chip twice over, else 128x64 (more workgroups, split-K on top if needed)
        const bool big = (d.M / 128) * (d.N / 128) * d.B >= 2 * kNumCU;
        if ((want == kFastCfg || (want < 0 && big)) && fast_tile_ok(p, d, kFastCfg)) cfg = kFastCfg;
        else if ((want == kFastCfgN || want < 0) && fast_tile_ok(p, d, kFastCfgN)) cfg = kFastCfgN;
      }
    }
    p->tile_cfg = cfg;
  }
  if ((p->tile_cfg == kFastCfg || p->tile_cfg == kFastCfgN) && !fast_tile_ok(p, d, p->tile_cfg))
    return QAMD_EUNSUPPORTED;
  // ---- split-K for launches that cannot fill the chip ------------------------
  if (p->split_k < 1) {
    int bm = kTileBM[p->tile_cfg], bn = kTileBN[p->tile_cfg];
    int64_t tiles = ((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn) * d.B;
    const int bk = kTileBK[p->tile_cfg];
    int64_t ksteps = (d.K + bk - 1) / bk;
    int64_t s = 1;
    // aim at ~4 workgroups per CU (several are resident per CU at these register counts; a
    // grid of one workgroup per CU leaves 3/4 of the wave slots empty), >= 8 k-tiles per split
    // (the slab reduction moves (2s + 1) x |C|: only worth it when K is long or the chip is under-filled)
    if (tiles < 3 * kNumCU && (ksteps >= 64 || (tiles < kNumCU && ksteps >= 16))) {
      s = (4 * kNumCU + tiles - 1) / tiles;
      s = std::min<int64_t>(s, ksteps / 8);
      s = std::min<int64_t>(s, 1024);
      s = std::max<int64_t>(s, 1);
    }
    if (c_extent(p) != d.B * d.M * d.N) s = 1;   // the slab reduction needs a compact C
    p->split_k = (int32_t)s;
  }
  return QAMD_OK;
}

static int64_t kpad_of(int64_t K) { return ((K + kBK - 1) / kBK) * kBK; }

extern "C" int64_t qamd_pair_ktab_len(const qamd_pair_plan* p) {
  PairDims d;
  if (pair_dims(p, d)) return -1;
  return 2 * kpad_of(d.K);
}

extern "C" int qamd_pair_build_ktab(const qamd_pair_plan* p, void* ktab, void* stream) {
  PairDims d;
  int rc = pair_dims(p, d);
  if (rc) return rc;
  KtabArgs a;
  memset(&a, 0, sizeof(a));
  a.nk = p->nk;
  a.K = (uint32_t)d.K;
  a.Kpad = (uint32_t)kpad_of(d.K);
  for (int i = 0; i < p->nk; ++i) {
    a.dim_k[i] = (uint32_t)p->dim_k[i];
    a.sa_k[i] = p->sa_k[i];
    a.sb_k[i] = p->sb_k[i];
  }
  return qamd_build_ktab_launch(ktab, &a, stream);
}

static int64_t c_extent(const qamd_pair_plan* p) {
  int64_t e = 1;
  for (int i = 0; i < p->nb; ++i) e += (p->dim_b[i] - 1) * p->sc_b[i];
  for (int i = 0; i < p->nm; ++i) e += (p->dim_m[i] - 1) * p->sc_m[i];
  for (int i = 0; i < p->nn; ++i) e += (p->dim_n[i] - 1) * p->sc_n[i];
  return e;
}

extern "C" int64_t qamd_pair_workspace_bytes(const qamd_pair_plan* p) {
  PairDims d;
  if (pair_dims(p, d)) return -1;
  if (p->kernel == 7) return gemmh_workspace_bytes(p, d);
  if (p->split_k <= 1) return 0;
  return (int64_t)p->split_k * d.B * d.M * d.N * kEsize[p->dtype];
}

extern "C" int qamd_contract_pair(const qamd_pair_plan* p, const void* A, const void* B, void* C,
                                  const void* ktab, void* ws, int64_t ws_bytes, void* stream) {
  return qamd_contract_pair_ex(p, A, B, C, ktab, ws, ws_bytes, nullptr, stream);
}

// Does the launch qualify for the straight-line sweep kernel?  (aligned chunks, full
// 16-byte vectors, K <= 36, 32-bit in-chunk byte offsets, a usable chunk-range divisor)
static bool sweep_config(const qamd_pair_plan* p, const StreamArgs& s, int V, int& PS, uint32_t& cpb) {
  const int es = kEsize[p->dtype];
  const int ev = 16 / es;
  int64_t kmax = 0;
  for (int i = 0; i < p->nk; ++i) kmax += (p->dim_k[i] - 1) * p->sa_k[i];
  if (!(s.aligned && !s.c_break && V == ev && s.KS <= 9 && (kmax + 16 * ev) * es < (1ll << 32)))
    return false;
  static const int kPS[6] = {1, 2, 3, 4, 6, 9};
  PS = 9;
  for (int i = 0; i < 6; ++i) if (kPS[i] >= (int)s.KS) { PS = kPS[i]; break; }
  // chunks per workgroup: a divisor of the innermost group's chunk count, near the target
  const uint32_t ic = s.inner_chunks;
  // aim at ~12 workgroups per CU so that the dispatcher can balance the chip (a grid of
  // only 1-2 workgroups per CU leaves SIMDs with a single wave and CUs with uneven work)
  const uint32_t target = std::max<uint32_t>(8, (s.chunks + 256 * 12 - 1) / (256 * 12));
  uint32_t best = 0;
  for (uint32_t dlo = 1; (uint64_t)dlo * dlo <= ic; ++dlo) {
    if (ic % dlo) continue;
    uint32_t cand[2] = {dlo, ic / dlo};
    for (uint32_t c : cand)
      if (c <= target && c > best) best = c;
  }
  cpb = best;
  return best >= 4;
}

static void fill_stream_args(const qamd_pair_plan* p, const PairDims& d, StreamArgs& s) {
  memset(&s, 0, sizeof(s));
  s.nm = p->nm; s.nn = p->nn;
  for (int i = 0; i < p->nm; ++i) { s.dim_m[i] = (uint32_t)p->dim_m[i]; s.sa_m[i] = p->sa_m[i]; s.sc_m[i] = p->sc_m[i]; }
  for (int i = 0; i < p->nn; ++i) { s.dim_n[i] = (uint32_t)p->dim_n[i]; s.sb_n[i] = p->sb_n[i]; s.sc_n[i] = p->sc_n[i]; }
  s.M = (uint32_t)d.M; s.N = (uint32_t)d.N; s.K = (uint32_t)d.K;
  s.KS = (uint32_t)((d.K + 3) / 4);
  s.Kpad = 4 * s.KS;
  s.KpadTab = (uint32_t)kpad_of(d.K);
  s.NT = (uint32_t)((d.N + 15) / 16);
  const int V = p->vec_c;
  s.zmode = (p->kernel == 2) ? 1 : 0;
  s.d_in = s.zmode ? (uint32_t)n_inner_block(p) : 1;
  s.sc_m_in = p->sc_m[p->nm - 1];
  s.aligned = (p->dim_m[p->nm - 1] % (16 * V) == 0) ? 1 : 0;
  s.inner_chunks = s.aligned ? (uint32_t)(p->dim_m[p->nm - 1] / (16 * V)) : 1;
  s.l_in = (uint32_t)p->dim_m[p->nm - 1];
  if (s.zmode && !s.aligned) {
    // (finalize admitted this plan to the Z path through z_break_run: V = 16 / itemsize here)
    const int64_t run = z_break_run(p, V, (int64_t)s.d_in);
    if (run > 0) {
      s.c_break = 1;
      s.aligned = 1;                                        // loads: chunks are whole pieces of A's contiguous run
      s.inner_chunks = (uint32_t)(run / (16 * V));
    }
  }
  s.chunks = (uint32_t)((d.M + 16 * V - 1) / (16 * V));
  s.chunks_per_wave = stream_chunks_per_wave(d.M, V);
  uint32_t waves = (s.chunks + s.chunks_per_wave - 1) / s.chunks_per_wave;
  s.grid = (waves + 3) / 4;
  if (s.c_break) s.zb_groups = stream_zb_groups(d.M, V, s.l_in);      // (the size finalize counted into its LDS check)
}

static int launch_stream(const qamd_pair_plan* p, const PairDims& d, const void* A, const void* B, void* C,
                         const void* ktab, const qamd_epilogue* ep, void* stream) {
  StreamArgs s;
  fill_stream_args(p, d, s);
  const int V = p->vec_c;
  // ---- straight-line "sweep" kernel when the launch qualifies ------------------
  int PS = 0;
  uint32_t cpb = 0;
  if (sweep_config(p, s, V, PS, cpb)) {
    StreamArgs w = s;
    w.Kpad = 4 * PS;
    w.chunks_per_wave = cpb;  // chunks per WORKGROUP for the sweep kernel
    w.grid = s.chunks / cpb;
    if (p->dtype == QAMD_F32)
      return qamd_sweep_launch_f32(PS, &w, A, B, C, ktab, ep ? ep->scale_a : nullptr,
                                   ep ? ep->scale_b : nullptr, ep ? ep->absmax_out : nullptr, stream);
    return qamd_sweep_launch_f64(PS, &w, A, B, C, ktab, ep ? ep->scale_a : nullptr,
                                 ep ? ep->scale_b : nullptr, ep ? ep->absmax_out : nullptr, stream);
  }
  return qamd_stream_launch(p->dtype, V, &s, A, B, C, ktab, ep ? ep->scale_a : nullptr,
                            ep ? ep->scale_b : nullptr, ep ? ep->absmax_out : nullptr, stream);
}

extern "C" int qamd_gemmk_dot_launch(int ta, int tb, const GettArgs* a, const void* A, const void* B, const void* T,
                                     void* partial, void* stream);
extern "C" int qamd_gemmk_dot_finish(void* out, const void* partial, int n, const void* scale_a, const void* scale_b,
                                     const void* scale_t, void* absmax_out, void* stream);

// the kernel-role view of a GEMM-shaped plan (gemmk.hip / gemmh.hip): "m" = rows of the MFMA result, "n" = its lanes (C's
// contiguous side); the operands swap roles when C's stride-1 index lives in the M bundle
static void gemm_role_args(const qamd_pair_plan* p, const PairDims& d, int ta, int tb, const void* C, GettArgs& a) {
  const bool swap = !p->c_ncontig;
  memset(&a, 0, sizeof(a));
  a.nb = p->nb;
  for (int i = 0; i < p->nb; ++i) {
    a.dim_b[i] = (uint32_t)p->dim_b[i];
    a.sa_b[i] = swap ? p->sb_b[i] : p->sa_b[i];
    a.sb_b[i] = swap ? p->sa_b[i] : p->sb_b[i];
    a.sc_b[i] = p->sc_b[i];
  }
  const int nm = swap ? p->nn : p->nm, nn = swap ? p->nm : p->nn;
  const int64_t* dm = swap ? p->dim_n : p->dim_m;  const int64_t* dn = swap ? p->dim_m : p->dim_n;
  const int64_t* sam = swap ? p->sb_n : p->sa_m;   const int64_t* sbn = swap ? p->sa_m : p->sb_n;
  const int64_t* scm = swap ? p->sc_n : p->sc_m;   const int64_t* scn = swap ? p->sc_m : p->sc_n;
  a.nm = nm; a.nn = nn; a.nk = 1;
  for (int i = 0; i < nm; ++i) { a.dim_m[i] = (uint32_t)dm[i]; a.sa_m[i] = sam[i]; a.sc_m[i] = scm[i]; }
  for (int i = 0; i < nn; ++i) { a.dim_n[i] = (uint32_t)dn[i]; a.sb_n[i] = sbn[i]; a.sc_n[i] = scn[i]; }
  a.B = (uint32_t)d.B; a.M = (uint32_t)(swap ? d.N : d.M); a.N = (uint32_t)(swap ? d.M : d.N); a.K = (uint32_t)d.K;
  a.sa_k0 = swap ? p->sb_k[0] : p->sa_k[0];
  a.sb_k0 = swap ? p->sa_k[0] : p->sb_k[0];
  a.tiles_m = (uint32_t)((a.M + 64 * ta - 1) / (64 * ta));
  a.tiles_n = (uint32_t)((a.N + 64 * tb - 1) / (64 * tb));
  a.split_k = 1;
  // vector stores along n: the innermost n group is stride-1 in C, a multiple of the vector, everything else aligned
  {
    int v = 1;
    if (nn >= 1 && scn[nn - 1] == 1) {
      std::vector<int64_t> others;
      for (int i = 0; i < p->nb; ++i) others.push_back(p->sc_b[i]);
      for (int i = 0; i < nm; ++i) others.push_back(scm[i]);
      for (int i = 0; i + 1 < nn; ++i) others.push_back(scn[i]);
      v = pick_vec(dn[nn - 1], (int64_t)((uintptr_t)C & 15) ? 4 : 16, 4, others);
    }
    a.vec_c = v;
  }
}

// dot_T != NULL: the result is not stored but multiplied with the tensor at dot_T (C's layout) and summed -- one double
// per workgroup into dot_partial (qamd_contract_pair_dot); C is then only consulted for its alignment class (= dot_T's)
static int launch_gemmk(const qamd_pair_plan* p, const PairDims& d, const void* A, const void* B, void* C,
                        const qamd_epilogue* ep, void* stream, const void* dot_T = nullptr, void* dot_partial = nullptr,
                        int* tiles_out = nullptr) {
  if (dot_T) C = const_cast<void*>(dot_T);
  const int ta = p->tile_cfg / 16, tb = p->tile_cfg % 16;
  if (ta < 2 || ta > 4 || tb < 2 || tb > 4 || p->dtype != QAMD_F32 || p->nk != 1 || !A || !B || !C) return QAMD_EINVAL;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return QAMD_EINVAL;
  const bool swap = !p->c_ncontig;
  GettArgs a;
  gemm_role_args(p, d, ta, tb, C, a);
  if (tiles_out) *tiles_out = (int)(a.tiles_m * a.tiles_n * a.B);
  if (dot_T) return dot_partial ? qamd_gemmk_dot_launch(ta, tb, &a, swap ? B : A, swap ? A : B, dot_T, dot_partial, stream) : 0;
  const void* sa = ep ? (swap ? ep->scale_b : ep->scale_a) : nullptr;
  const void* sb = ep ? (swap ? ep->scale_a : ep->scale_b) : nullptr;
  return qamd_gemmk_launch(ta, tb, &a, swap ? B : A, swap ? A : B, C, sa, sb, ep ? ep->absmax_out : nullptr, stream);
}

// ---- kernel 7 (gemmh.hip): the same joins as split products on the f16 matrix pipe (opt-in: plan.kernel = -7 on input) --------
// workspace: [0, 16) the two operands' (scale, 1 / scale) pairs, [256, 512) / [512, 768) absmax slots for operands that
// bring none, [1024, ..) the kernel-role "m" operand's column means + their partial sums (doubles), the "n" operand's, then
// the "m" operand's split images and the "n" operand's (every block a multiple of 256 bytes)
static const int kGemmhTiles[6][2] = {{4, 4}, {3, 4}, {4, 3}, {3, 3}, {2, 4}, {4, 2}};
static int64_t gemmh_kpad(int64_t K) { return (K + 31) / 32 * 32; }

static bool gemmh_config(const qamd_pair_plan* p, const PairDims& d, int pin, bool all_layouts, int& ta, int& tb) {
  if (p->dtype != QAMD_F32 || p->nk < 1 || p->nb != 0 || d.B != 1 || p->nm < 1 || p->nn < 1) return false;
  // kernel = -7: the k-outer joins only -- the pairs whose DEFAULT kernel (gemmk.hip) accumulates as one fp32 chain over k, which
  // the split products never round worse than.  kernel = -8: any operand layout (the split pass gathers with the operands' own
  // strides and the product kernel only ever sees the images): k-contiguous operands, K in several groups, complex pairs'
  // real expansions -- pairs whose default kernels (gettf.hip: k-tiles, split-K) accumulate in BLOCKS, which on incoherent
  // operands (random signs / phases) is more accurate than any single chain, this one included.
  if (!all_layouts) {
    if (p->nk != 1 || p->a_kcontig || p->b_kcontig || p->sa_m[p->nm - 1] != 1 || p->sb_n[p->nn - 1] != 1) return false;
    if (p->sa_k[0] <= 0 || p->sb_k[0] <= 0) return false;
  }
  // (worth two extra passes over the operands only where the product dominates them; a pinned tile waives the floors down
  // to what the kernel needs: two 32-k stages)
  if (d.M >= (1ll << 31) || d.N >= (1ll << 31) || d.K >= (1ll << 31) || d.K < 33) return false;
  if (!pin && (d.K < 256 || d.M < 256 || d.N < 256)) return false;
  const bool swap = !p->c_ncontig;
  const int64_t M = swap ? d.N : d.M, N = swap ? d.M : d.N;
  double best = 0;
  ta = tb = 0;
  for (const auto& t : kGemmhTiles) {
    if (pin && pin != 10 * t[0] + t[1]) continue;
    const int64_t tiles = ((M + 64 * t[0] - 1) / (64 * t[0])) * ((N + 64 * t[1] - 1) / (64 * t[1]));
    // rounds of one tile per CU (one workgroup per CU: 130 KB of LDS); the larger tile re-reads less
    // (even tiles run the eight-wave two-group kernel, gemmh8_kernel: measured 5-7 % faster per tile than the four-wave one)
    const bool w8 = t[0] % 2 == 0 && t[1] % 2 == 0;
    const double cost = (double)((tiles + kNumCU - 1) / kNumCU) * t[0] * t[1] * (1.0 + 0.02 * (16 - t[0] * t[1]) / 16.0) * (w8 ? 0.93 : 1.0);
    if (!ta || cost < best) { best = cost; ta = t[0]; tb = t[1]; }
  }
  return ta != 0;
}

static int64_t gemmh_workspace_bytes(const qamd_pair_plan* p, const PairDims& d) {
  const int ta = p->tile_cfg / 16, tb = p->tile_cfg % 16;
  const bool swap = !p->c_ncontig;
  const int64_t M = swap ? d.N : d.M, N = swap ? d.M : d.N;
  const int64_t mpad = (M + 64 * ta - 1) / (64 * ta) * (64 * ta), npad = (N + 64 * tb - 1) / (64 * tb) * (64 * tb);
  return 1024 + qamd_gemmh_mean_bytes(mpad) + qamd_gemmh_mean_bytes(npad) + qamd_gemmh_image_bytes(mpad, gemmh_kpad(d.K)) +
         qamd_gemmh_image_bytes(npad, gemmh_kpad(d.K));
}

static int launch_gemmh(const qamd_pair_plan* p, const PairDims& d, const void* A, const void* B, void* C, void* ws,
                        int64_t ws_bytes, const qamd_epilogue* ep, void* stream, const void* dot_T = nullptr,
                        void* dot_partial = nullptr, int* tiles_out = nullptr) {
  if (dot_T) C = const_cast<void*>(dot_T);
  const int ta = p->tile_cfg / 16, tb = p->tile_cfg % 16;
  if (ta < 2 || ta > 4 || tb < 2 || tb > 4 || p->dtype != QAMD_F32 || p->nk < 1 || p->nb != 0 || !A || !B || !C) return QAMD_EINVAL;
  if (!ws || ws_bytes < gemmh_workspace_bytes(p, d) || ((uintptr_t)ws & 15)) return QAMD_EWORKSPACE;
  const bool swap = !p->c_ncontig;
  GettArgs a;
  gemm_role_args(p, d, ta, tb, C, a);
  a.Kloop = (uint32_t)gemmh_kpad(d.K);
  if (tiles_out) *tiles_out = (int)(a.tiles_m * a.tiles_n);
  SplitArgs sm, sn;
  memset(&sm, 0, sizeof(sm));
  memset(&sn, 0, sizeof(sn));
  sm.ng = a.nm; sn.ng = a.nn;
  for (int i = 0; i < a.nm; ++i) { sm.dim[i] = a.dim_m[i]; sm.stride[i] = a.sa_m[i]; }
  for (int i = 0; i < a.nn; ++i) { sn.dim[i] = a.dim_n[i]; sn.stride[i] = a.sb_n[i]; }
  sm.sk = a.sa_k0; sn.sk = a.sb_k0;
  // an innermost K group of TWO (the re / im component of a complex pair's real expansion: ops._complex_spec): columns that
  // alternate between two populations with k -- one centring constant per parity of k
  const int period = (p->nk >= 2 && p->dim_k[p->nk - 1] == 2) ? 2 : 1;
  sm.period = sn.period = period;
  a.pad_ = period;
  sm.nk = sn.nk = p->nk;
  for (int i = 0; i < p->nk; ++i) {
    sm.dim_k[i] = sn.dim_k[i] = (uint32_t)p->dim_k[i];
    sm.stride_k[i] = swap ? p->sb_k[i] : p->sa_k[i];
    sn.stride_k[i] = swap ? p->sa_k[i] : p->sb_k[i];
  }
  sm.X = a.M; sm.Xpad = a.tiles_m * 64 * ta; sn.X = a.N; sn.Xpad = a.tiles_n * 64 * tb;
  sm.K = sn.K = a.K; sm.KG = sn.KG = a.Kloop / 8;
  char* w = (char*)ws;
  float* hdr_m = (float*)w;
  float* hdr_n = hdr_m + 2;
  char* mean_m = w + 1024;
  char* mean_n = mean_m + qamd_gemmh_mean_bytes(sm.Xpad);
  char* img_m = mean_n + qamd_gemmh_mean_bytes(sn.Xpad);
  char* img_n = img_m + qamd_gemmh_image_bytes(sm.Xpad, a.Kloop);
  const void* Am = swap ? B : A;
  const void* Bn = swap ? A : B;
  const void* slots_m = ep ? (swap ? ep->scale_b : ep->scale_a) : nullptr;
  const void* slots_n = ep ? (swap ? ep->scale_a : ep->scale_b) : nullptr;
  int rc;
  // an operand without exponent slots (a plain tensordot): its absmax is taken here, but the result is then NOT divided by it
  const void* split_m = slots_m;
  const void* split_n = slots_n;
  if (!split_m) { if ((rc = qamd_gemmh_absmax_launch(&sm, Am, w + 256, stream))) return rc; split_m = w + 256; }
  if (!split_n) { if ((rc = qamd_gemmh_absmax_launch(&sn, Bn, w + 512, stream))) return rc; split_n = w + 512; }
  // (both operands centred: the part of the product carried by the columns' means is added exactly in the epilogue)
  if ((rc = qamd_gemmh_split_launch(&sm, Am, split_m, hdr_m, img_m, mean_m, stream))) return rc;
  if ((rc = qamd_gemmh_split_launch(&sn, Bn, split_n, hdr_n, img_n, mean_n, stream))) return rc;
  if (dot_T)
    return dot_partial ? qamd_gemmh_dot_launch(ta, tb, &a, img_m, img_n, dot_T, hdr_m, hdr_n, mean_m, mean_n, dot_partial, stream) : 0;
  return qamd_gemmh_launch(ta, tb, &a, img_m, img_n, C, slots_m, slots_n, hdr_m, hdr_n, mean_m, mean_n,
                           ep ? ep->absmax_out : nullptr, stream);
}

extern "C" int qamd_gemmd_launch(int ta, int tb, const GettArgs* a, int swap, const void* A, const void* B, void* C,
                                 const void* ktab, const void* scale_a, const void* scale_b, void* absmax_out,
                                 void* stream);

static int launch_gemmd(const qamd_pair_plan* p, const PairDims& d, const void* A, const void* B, void* C,
                        const void* ktab, void* ws, int64_t ws_bytes, const qamd_epilogue* ep, void* stream) {
  const int ta = p->tile_cfg / 16, tb = p->tile_cfg % 16;
  if (ta < 2 || ta > 5 || tb < 1 || tb > 2 || p->dtype != QAMD_F64 || !A || !B || !C || !ktab || p->split_k < 1)
    return QAMD_EINVAL;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || d.K % 16) return QAMD_EINVAL;
  GettArgs a;
  memset(&a, 0, sizeof(a));
  a.nb = p->nb; a.nm = p->nm; a.nn = p->nn; a.nk = p->nk;
  for (int i = 0; i < p->nb; ++i) {
    a.dim_b[i] = (uint32_t)p->dim_b[i];
    a.sa_b[i] = p->sa_b[i]; a.sb_b[i] = p->sb_b[i]; a.sc_b[i] = p->sc_b[i];
  }
  for (int i = 0; i < p->nm; ++i) { a.dim_m[i] = (uint32_t)p->dim_m[i]; a.sa_m[i] = p->sa_m[i]; a.sc_m[i] = p->sc_m[i]; }
  for (int i = 0; i < p->nn; ++i) { a.dim_n[i] = (uint32_t)p->dim_n[i]; a.sb_n[i] = p->sb_n[i]; a.sc_n[i] = p->sc_n[i]; }
  for (int i = 0; i < p->nk; ++i) { a.dim_k[i] = (uint32_t)p->dim_k[i]; a.sa_k[i] = p->sa_k[i]; a.sb_k[i] = p->sb_k[i]; }
  a.B = (uint32_t)d.B; a.M = (uint32_t)d.M; a.N = (uint32_t)d.N; a.K = (uint32_t)d.K;
  a.Kpad = (uint32_t)kpad_of(d.K);
  a.Kloop = a.K;
  const int64_t ksteps = d.K / 16;
  int split = (int)std::min<int64_t>(p->split_k, ksteps);
  const int64_t steps_per = (ksteps + split - 1) / split;
  split = (int)((ksteps + steps_per - 1) / steps_per);
  a.Kc = (uint32_t)(steps_per * 16);
  a.split_k = (uint32_t)split;
  a.tiles_m = (uint32_t)((d.M + 32 * ta - 1) / (32 * ta));
  a.tiles_n = (uint32_t)((d.N + 64 * tb - 1) / (64 * tb));
  a.a_kcontig = p->a_kcontig; a.b_kcontig = p->b_kcontig;
  const int swap = p->c_ncontig ? 0 : 1;
  const void* sa = ep ? ep->scale_a : nullptr;
  const void* sb = ep ? ep->scale_b : nullptr;
  void* amax = ep ? ep->absmax_out : nullptr;
  if (split == 1) {
    a.slab_stride = 0;
    return qamd_gemmd_launch(ta, tb, &a, swap, A, B, C, ktab, sa, sb, amax, stream);
  }
  const int64_t csz = d.B * d.M * d.N;
  if (c_extent(p) != csz) return QAMD_EUNSUPPORTED;  // split-K needs a compact C
  if (!ws || ws_bytes < (int64_t)split * csz * 8) return QAMD_EWORKSPACE;
  a.slab_stride = csz;
  int rc = qamd_gemmd_launch(ta, tb, &a, swap, A, B, ws, ktab, nullptr, nullptr, nullptr, stream);
  if (rc) return rc;
  return qamd_splitk_reduce_launch(p->dtype, C, ws, csz, split, sa, sb, amax, stream);
}

// ---- a join consumed by one inner product (the closing step of a two-sided / four-quadrant contraction) -------------
static int64_t dot_partial_bytes(const qamd_pair_plan* p, const PairDims& d) {
  const int ta = p->tile_cfg / 16, tb = p->tile_cfg % 16;
  if (ta < 2 || ta > 4 || tb < 2 || tb > 4) return 0;
  const bool swap = !p->c_ncontig;
  const int64_t M = swap ? d.N : d.M, N = swap ? d.M : d.N;
  return (int64_t)sizeof(double) * d.B * ((M + 64 * ta - 1) / (64 * ta)) * ((N + 64 * tb - 1) / (64 * tb));
}

// kernel 7: the partial sums (rounded up to 256 bytes) are followed by the split product's own workspace
extern "C" int64_t qamd_pair_dot_workspace_bytes(const qamd_pair_plan* p) {
  if (!p || (p->kernel != 5 && p->kernel != 7)) return 0;
  PairDims d;
  if (pair_dims(p, d)) return 0;
  const int64_t part = dot_partial_bytes(p, d);
  if (part <= 0 || p->kernel == 5) return part;
  return (part + 255) / 256 * 256 + gemmh_workspace_bytes(p, d);
}

extern "C" int qamd_contract_pair_dot(const qamd_pair_plan* p, const void* A, const void* B, const void* T, void* out_dev,
                                      void* ws, int64_t ws_bytes, const qamd_epilogue* ep, const void* scale_t,
                                      void* stream) {
  if (!p || !A || !B || !T || !out_dev) return QAMD_EINVAL;
  if ((p->kernel != 5 && p->kernel != 7) || p->dtype != QAMD_F32) return QAMD_EUNSUPPORTED;
  if (qamdp_recording()) return qamdp_rec_pair_dot(p, A, B, T, out_dev, ws, ws_bytes, ep, scale_t);
  PairDims d;
  int rc = pair_dims(p, d);
  if (rc) return rc;
  const int64_t need = qamd_pair_dot_workspace_bytes(p);
  if (need <= 0) return QAMD_EINVAL;
  if (!ws || ws_bytes < need) return QAMD_EWORKSPACE;
  int tiles = 0;
  if (p->kernel == 7) {
    const int64_t part = (dot_partial_bytes(p, d) + 255) / 256 * 256;
    rc = launch_gemmh(p, d, A, B, nullptr, (char*)ws + part, ws_bytes - part, ep, stream, T, ws, &tiles);
  } else {
    rc = launch_gemmk(p, d, A, B, nullptr, nullptr, stream, T, ws, &tiles);
  }
  if (rc) return rc;
  return qamd_gemmk_dot_finish(out_dev, ws, tiles, ep ? ep->scale_a : nullptr, ep ? ep->scale_b : nullptr, scale_t,
                               ep ? ep->absmax_out : nullptr, stream);
}

extern "C" int qamd_contract_pair_ex(const qamd_pair_plan* p, const void* A, const void* B, void* C,
                                     const void* ktab, void* ws, int64_t ws_bytes, const qamd_epilogue* ep,
                                     void* stream) {
  if (qamdp_recording()) return qamdp_rec_pair(p, A, B, C, ktab, ws, ws_bytes, ep);
  PairDims d;
  int rc = pair_dims(p, d);
  if (rc) return rc;
  if (p->dtype != QAMD_F32 && p->dtype != QAMD_F64) return QAMD_EUNSUPPORTED;
  if (p->kernel == 5) return launch_gemmk(p, d, A, B, C, ep, stream);
  if (p->kernel == 7) return launch_gemmh(p, d, A, B, C, ws, ws_bytes, ep, stream);
  if (p->kernel == 6) return launch_gemmd(p, d, A, B, C, ktab, ws, ws_bytes, ep, stream);
  if (p->tile_cfg < 0 || p->tile_cfg >= kNumTileCfg || p->split_k < 1) return QAMD_EINVAL;
  if (!A || !B || !C || !ktab) return QAMD_EINVAL;
  if (p->kernel == 1 || p->kernel == 2) return launch_stream(p, d, A, B, C, ktab, ep, stream);
  if (p->kernel == 4) {
    DotArgs da;
    memset(&da, 0, sizeof(da));
    if (!dot_rows(p, d, da) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return QAMD_EINVAL;
    da.grid = (uint32_t)p->split_k;
    const int64_t csz = d.M * d.N;
    if (!ws || ws_bytes < (int64_t)p->split_k * csz * kEsize[p->dtype]) return QAMD_EWORKSPACE;
    const bool rows_a = d.N == 1 && d.M <= 32;
    return qamd_dotm_launch(p->dtype, &da, rows_a ? A : B, rows_a ? B : A, ws, C, ep ? ep->scale_a : nullptr,
                            ep ? ep->scale_b : nullptr, ep ? ep->absmax_out : nullptr, stream);
  }
  const int bm = kTileBM[p->tile_cfg], bn = kTileBN[p->tile_cfg];
  const void* sa = ep ? ep->scale_a : nullptr;
  const void* sb = ep ? ep->scale_b : nullptr;
  void* amax = ep ? ep->absmax_out : nullptr;

  GettArgs a;
  memset(&a, 0, sizeof(a));
  a.nb = p->nb; a.nm = p->nm; a.nn = p->nn; a.nk = p->nk;
  for (int i = 0; i < p->nb; ++i) {
    a.dim_b[i] = (uint32_t)p->dim_b[i];
    a.sa_b[i] = p->sa_b[i]; a.sb_b[i] = p->sb_b[i]; a.sc_b[i] = p->sc_b[i];
  }
  for (int i = 0; i < p->nm; ++i) {
    a.dim_m[i] = (uint32_t)p->dim_m[i];
    a.sa_m[i] = p->sa_m[i]; a.sc_m[i] = p->sc_m[i];
  }
  for (int i = 0; i < p->nn; ++i) {
    a.dim_n[i] = (uint32_t)p->dim_n[i];
    a.sb_n[i] = p->sb_n[i]; a.sc_n[i] = p->sc_n[i];
  }
  a.B = (uint32_t)d.B; a.M = (uint32_t)d.M; a.N = (uint32_t)d.N; a.K = (uint32_t)d.K;
  a.Kpad = (uint32_t)kpad_of(d.K);
  int split = p->split_k;
  const int bk = kTileBK[p->tile_cfg];
  a.Kloop = (uint32_t)(((d.K + bk - 1) / bk) * bk);
  int64_t ksteps = (d.K + bk - 1) / bk;
  if (split > ksteps) split = (int)std::max<int64_t>(ksteps, 1);
  int64_t steps_per = (ksteps + split - 1) / split;
  split = (int)((ksteps + steps_per - 1) / steps_per);
  if (split < 1) split = 1;
  a.Kc = (uint32_t)(steps_per * bk);
  a.split_k = (uint32_t)split;
  a.tiles_m = (uint32_t)((d.M + bm - 1) / bm);
  a.tiles_n = (uint32_t)((d.N + bn - 1) / bn);
  a.vec_a = p->vec_a; a.vec_b = p->vec_b;
  a.a_kcontig = p->a_kcontig; a.b_kcontig = p->b_kcontig;
  const int swap = p->c_ncontig ? 0 : 1;

  if (p->tile_cfg == kFastCfg || p->tile_cfg == kFastCfgN) {
    const uintptr_t es_ = (uintptr_t)kEsize[p->dtype];
    const uintptr_t ma = std::min<uintptr_t>(16, p->vec_a * es_) - 1, mb = std::min<uintptr_t>(16, p->vec_b * es_) - 1;
    if (!fast_tile_ok(p, d, p->tile_cfg) || ((uintptr_t)A & ma) || ((uintptr_t)B & mb)) return QAMD_EINVAL;
    if (split == 1) {
      a.slab_stride = 0;
      return qamd_gettf_launch(p->dtype, bn, &a, swap, A, B, C, ktab, sa, sb, amax, stream);
    }
    const int64_t csz = d.B * d.M * d.N;
    if (c_extent(p) != csz) return QAMD_EUNSUPPORTED;  // split-K needs a compact C
    if (!ws || ws_bytes < (int64_t)split * csz * kEsize[p->dtype]) return QAMD_EWORKSPACE;
    a.slab_stride = csz;
    rc = qamd_gettf_launch(p->dtype, bn, &a, swap, A, B, ws, ktab, nullptr, nullptr, nullptr, stream);
    if (rc) return rc;
    return qamd_splitk_reduce_launch(p->dtype, C, ws, csz, split, sa, sb, amax, stream);
  }
  if (split == 1) {
    a.slab_stride = 0;
    return qamd_gett_launch(p->dtype, p->tile_cfg, &a, swap, A, B, C, ktab, sa, sb, amax, stream);
  }
  const int64_t csize = d.B * d.M * d.N;
  if (c_extent(p) != csize) return QAMD_EUNSUPPORTED;  // split-K needs a compact C
  if (!ws || ws_bytes < (int64_t)split * csize * kEsize[p->dtype]) return QAMD_EWORKSPACE;
  a.slab_stride = csize;
  rc = qamd_gett_launch(p->dtype, p->tile_cfg, &a, swap, A, B, ws, ktab, nullptr, nullptr, nullptr, stream);
  if (rc) return rc;
  return qamd_splitk_reduce_launch(p->dtype, C, ws, csize, split, sa, sb, amax, stream);
}

// ---------------------------------------------------------------------------
// permute
// ---------------------------------------------------------------------------
struct Dim { int64_t n, ss, sd; };

static void fuse_dims(std::vector<Dim>& d) {
  // d in dst order (sd decreasing, contiguous): merge i,i+1 when src agrees
  std::vector<Dim> out;
  for (const Dim& x : d) {
    if (x.n == 1) continue;
    if (!out.empty() && out.back().ss == x.ss * x.n && out.back().sd == x.sd * x.n) {
      out.back().n *= x.n;
      out.back().ss = x.ss;
      out.back().sd = x.sd;
    } else {
      out.push_back(x);
    }
  }
  d.swap(out);
}

static int fill_bundle(const std::vector<Dim>& v, uint32_t* dim, int64_t* ss, int64_t* sd, int32_t& n, uint32_t& total) {
  if ((int)v.size() > QAMD_PG) return QAMD_EUNSUPPORTED;
  int64_t p = 1;
  n = (int32_t)v.size();
  for (int i = 0; i < n; ++i) {
    dim[i] = (uint32_t)v[i].n;
    ss[i] = v[i].ss;
    sd[i] = v[i].sd;
    p *= v[i].n;
    if (p >= (1ll << 31)) return QAMD_EUNSUPPORTED;
  }
  total = (uint32_t)p;
  return 0;
}

// Plan for permute_stream_kernel (elementwise.hip): the tile is a set of (parts of) dims holding a contiguous
// run of >= 32 elements on the source side AND on the destination side where the shape allows, <= 4096
// elements in all; a dim that is too long is split into (outer, inner) and the outer part joins Z, the loop
// of a workgroup; offsets fit 32 bits.
static int64_t largest_divisor_le(int64_t n, int64_t limit) {
  for (int64_t t = std::min(n, limit); t >= 2; --t)
    if (n % t == 0) return t;
  return 1;
}

static bool plan_permute_stream(const std::vector<Dim>& d0, int64_t src_offset, int64_t cap, PermArgs& a) {
  std::vector<Dim> pool = d0, tile;
  int64_t span_s = 0, span_d = 0;
  for (const Dim& x : pool) { span_s += (x.n - 1) * std::llabs(x.ss); span_d += (x.n - 1) * x.sd; }
  if (span_s >= (1ll << 31) || span_d >= (1ll << 31)) return false;
  for (const Dim& x : pool) if (x.ss <= 0) return false;      // reversed / broadcast views: the general kernel
  int64_t vol = 1;
  // move (a part of) pool[idx], at most ``room`` elements, into the tile; the outer remainder stays in the pool
  auto take = [&](size_t idx, int64_t room) -> bool {
    Dim x = pool[idx];
    if (x.n <= room) { tile.push_back(x); vol *= x.n; pool.erase(pool.begin() + idx); return true; }
    const int64_t t = largest_divisor_le(x.n, room);
    if (t < 2) return false;
    tile.push_back(Dim{t, x.ss, x.sd});
    vol *= t;
    pool[idx] = Dim{x.n / t, x.ss * t, x.sd * t};
    return true;
  };
  // length of the contiguous run the tile holds on one side: groups chained stride -> stride * n from stride 1
  auto run_len = [&](bool by_src) -> int64_t {
    int64_t run = 1;
    for (bool grown = true; grown;) {
      grown = false;
      for (const Dim& x : tile)
        if ((by_src ? x.ss : x.sd) == run && x.n > 1) { run *= x.n; grown = true; break; }
    }
    return run;
  };
  // the pool dim that continues that run, or -1
  auto next_of = [&](bool by_src, int64_t run) -> int {
    for (size_t i = 0; i < pool.size(); ++i)
      if ((by_src ? pool[i].ss : pool[i].sd) == run) return (int)i;
    return -1;
  };
  auto grow = [&](bool by_src, int64_t want, int64_t cap) {
    for (;;) {
      const int64_t run = run_len(by_src);
      if (run >= want || vol >= cap) return;
      const int idx = next_of(by_src, run);
      if (idx < 0 || !take((size_t)idx, std::min<int64_t>(cap / vol, std::max<int64_t>(2, 4 * want / run)))) return;
    }
  };
  // ``cap``: elements per tile (4096; 2048 for 16-byte elements, whose padded tile must stay inside 64 KB of LDS)
  grow(true, 32, cap);             // a source run of >= 32 elements ...
  grow(false, 32, cap);            // ... and a destination run of >= 32
  for (int round = 0; round < 8 && vol < cap / 2; ++round) {    // then lengthen both, in turn
    const int64_t before = vol;
    grow(true, 2 * run_len(true), cap);
    if (vol < cap / 2) grow(false, 2 * run_len(false), cap);
    if (vol == before) break;
  }
  if (vol < 512 || run_len(true) < 8) return false;
  std::stable_sort(tile.begin(), tile.end(), [](const Dim& p, const Dim& q) { return p.ss > q.ss; });   // outermost first
  memset(&a, 0, sizeof(a));
  if (fill_bundle(tile, a.dim_x, a.ss_x, a.sd_x, a.nx, a.X)) return false;
  if (fill_bundle(pool, a.dim_z, a.ss_z, a.sd_z, a.nz, a.Z)) return false;
  std::vector<int> ord(a.nx);
  for (int i = 0; i < a.nx; ++i) ord[i] = i;
  std::stable_sort(ord.begin(), ord.end(), [&](int p, int q) { return a.sd_x[p] < a.sd_x[q]; });
  a.direct = 1;
  for (int i = 0; i < a.nx; ++i) {
    a.xorder[i] = ord[i];
    if (ord[i] != a.nx - 1 - i) a.direct = 0;       // same order on both sides inside the tile: no LDS stage
  }
  a.Y = 1; a.TX = (int32_t)a.X; a.TY = 1; a.tiles_x = a.tiles_y = 1;
  a.src_offset = src_offset;
  const uint64_t zsplit = std::max<uint64_t>(1, std::min<uint64_t>(a.Z, 4096));   // ~16 workgroups per CU
  a.zchunk = (uint32_t)((a.Z + zsplit - 1) / zsplit);
  return true;
}

extern "C" int qamd_permute(void* dst, const void* src, int32_t ndim, const int64_t* shape,
                            const int64_t* src_strides, int64_t src_offset, int32_t dtype, void* stream) {
  if (qamdp_recording()) return qamdp_rec_permute(dst, src, ndim, shape, src_strides, src_offset, dtype);
  if (ndim < 0 || ndim > QAMD_MAX_NDIM || dtype < 0 || dtype > 3) return QAMD_EINVAL;
  if (!dst || !src) return QAMD_EINVAL;
  std::vector<Dim> d(ndim);
  int64_t sd = 1;
  for (int i = ndim - 1; i >= 0; --i) {
    if (shape[i] < 0) return QAMD_EINVAL;
    if (shape[i] == 0) return QAMD_OK;
    d[i] = Dim{shape[i], src_strides[i], sd};
    sd *= shape[i];
  }
  fuse_dims(d);
  if (d.empty()) d.push_back(Dim{1, 1, 1});
  const int n = (int)d.size();
  {
    // the z-looping, offset-caching kernel first; the tile-per-workgroup one takes what it cannot address
    PermArgs ps;
    if (plan_permute_stream(d, src_offset, kEsize[dtype] == 16 ? 2048 : 4096, ps)) {
      int rc = qamd_permute_stream_launch(kEsize[dtype], dst, src, &ps, stream);
      if (rc != -2) return rc;
    }
  }

  // src-fastest dim
  int fi = n - 1;
  for (int i = n - 1; i >= 0; --i)
    if (std::llabs(d[i].ss) < std::llabs(d[fi].ss)) fi = i;

  std::vector<Dim> X, Y, Z;
  std::vector<char> used(n, 0);
  PermArgs a;
  memset(&a, 0, sizeof(a));
  if (fi == n - 1) {
    // direct: trailing dims are fast on both sides
    int64_t px = 1;
    int i = n - 1;
    for (; i >= 0 && px < 256; --i) { X.insert(X.begin(), d[i]); used[i] = 1; px *= d[i].n; }
    int64_t py = 1;
    for (; i >= 0 && py < 16; --i) { Y.insert(Y.begin(), d[i]); used[i] = 1; py *= d[i].n; }
    a.direct = 1;
    a.TX = (int32_t)std::min<int64_t>(px, 256);
    a.TY = (int32_t)std::max<int64_t>(1, std::min<int64_t>(py, 4096 / a.TX));
  } else {
    int64_t py = 1;
    for (int i = n - 1; i >= 0 && py < 64 && i != fi; --i) { Y.insert(Y.begin(), d[i]); used[i] = 1; py *= d[i].n; }
    // remaining dims ordered by |src stride| ascending -> X, innermost = smallest stride
    std::vector<int> rest;
    for (int i = 0; i < n; ++i) if (!used[i]) rest.push_back(i);
    std::sort(rest.begin(), rest.end(), [&](int p, int q) { return std::llabs(d[p].ss) < std::llabs(d[q].ss); });
    int64_t px = 1;
    for (int idx : rest) {
      if (px >= 64) break;
      X.insert(X.begin(), d[idx]);
      used[idx] = 1;
      px *= d[idx].n;
    }
    a.direct = 0;
    a.TX = (int32_t)std::min<int64_t>(px, 64);
    a.TY = (int32_t)std::min<int64_t>(py, 64);
  }
  for (int i = 0; i < n; ++i) if (!used[i]) Z.push_back(d[i]);

  int rc;
  if ((rc = fill_bundle(X, a.dim_x, a.ss_x, a.sd_x, a.nx, a.X))) return rc;
  if ((rc = fill_bundle(Y, a.dim_y, a.ss_y, a.sd_y, a.ny, a.Y))) return rc;
  if ((rc = fill_bundle(Z, a.dim_z, a.ss_z, a.sd_z, a.nz, a.Z))) return rc;
  a.tiles_x = (a.X + a.TX - 1) / a.TX;
  a.tiles_y = (a.Y + a.TY - 1) / a.TY;
  a.src_offset = src_offset;
  return qamd_permute_launch(kEsize[dtype], dst, src, &a, stream);
}

// ---------------------------------------------------------------------------
// strided reduce / binary
// ---------------------------------------------------------------------------
extern "C" int qamd_reduce_sum(void* out, const void* x, int32_t ndk, const int64_t* shape_keep,
                               const int64_t* strides_keep, int32_t ndr, const int64_t* shape_red,
                               const int64_t* strides_red, int32_t dtype, void* stream) {
  if (ndk < 0 || ndr < 0 || ndk > QAMD_PG || ndr > QAMD_PG || dtype < 0 || dtype > 3) return QAMD_EINVAL;
  if (qamdp_recording()) return qamdp_rec_reduce(out, x, ndk, shape_keep, strides_keep, ndr, shape_red, strides_red, dtype);
  ReduceArgs a;
  memset(&a, 0, sizeof(a));
  int64_t nk = 1, nr = 1;
  for (int i = 0; i < ndk; ++i) {
    if (shape_keep[i] <= 0) return shape_keep[i] == 0 ? QAMD_OK : QAMD_EINVAL;
    a.dim_keep[i] = (uint32_t)shape_keep[i];
    a.s_keep[i] = strides_keep[i];
    nk *= shape_keep[i];
  }
  for (int i = 0; i < ndr; ++i) {
    if (shape_red[i] <= 0) return QAMD_EINVAL;
    a.dim_red[i] = (uint32_t)shape_red[i];
    a.s_red[i] = strides_red[i];
    nr *= shape_red[i];
  }
  if (nk >= (1ll << 31) || nr >= (1ll << 31)) return QAMD_EUNSUPPORTED;
  a.nd_keep = ndk; a.nd_red = ndr;
  a.n_keep = (uint32_t)nk; a.n_red = (uint32_t)nr;
  a.wave_per_out = (nr >= 256 || nk < 4096) ? 1 : 0;
  return qamd_reduce_sum_launch(dtype, out, x, &a, stream);
}

extern "C" int qamd_binary(void* out, const void* x, const int64_t* xs, const void* y, const int64_t* ys,
                           int32_t ndim, const int64_t* shape, int32_t op, int32_t dtype, void* stream) {
  if (ndim < 0 || ndim > QAMD_MAX_NDIM || dtype < 0 || dtype > 3 || op < 0 || op > 3) return QAMD_EINVAL;   // 0 add, 1 mul, 2 sub, 3 true division
  if (qamdp_recording()) return qamdp_rec_binary(out, x, xs, y, ys, ndim, shape, op, dtype);
  // fuse adjacent dims where both operands allow it
  struct D3 { int64_t n, sa, sb; };
  std::vector<D3> v;
  int64_t total = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] < 0) return QAMD_EINVAL;
    if (shape[i] == 0) return QAMD_OK;
    total *= shape[i];
    if (shape[i] == 1) continue;
    D3 cur{shape[i], xs[i], ys[i]};
    if (!v.empty() && v.back().sa == cur.sa * cur.n && v.back().sb == cur.sb * cur.n) {
      v.back().n *= cur.n; v.back().sa = cur.sa; v.back().sb = cur.sb;
    } else v.push_back(cur);
  }
  if ((int)v.size() > QAMD_PG) return QAMD_EUNSUPPORTED;
  BinaryArgs a;
  memset(&a, 0, sizeof(a));
  a.nd = (int32_t)v.size();
  a.op = op;
  a.n = total;
  for (int i = 0; i < a.nd; ++i) { a.dim[i] = v[i].n; a.sa[i] = v[i].sa; a.sb[i] = v[i].sb; }
  return qamd_binary_launch(dtype, out, x, y, &a, stream);
}

extern "C" int qamd_pair_describe(const qamd_pair_plan* p, char* buf, int32_t buflen) {
  PairDims d;
  int rc = pair_dims(p, d);
  if (rc || !buf || buflen <= 0) return QAMD_EINVAL;
  const char* T = p->dtype == QAMD_F32 ? "float" : (p->dtype == QAMD_F64 ? "double" : "?");
  if (p->kernel == 4) {
    snprintf(buf, buflen, "dotm_kernel<%s, %d>", T, (int)std::max(d.M, d.N));
    return QAMD_OK;
  }
  if (p->kernel == 5) {
    const int ta = p->tile_cfg / 16, tb = p->tile_cfg % 16;
    const bool two = ta * tb == 12 || ta * tb == 9;   // (gemmk.hip, QAMD_GEMMK_CASES: two-stage ring, two workgroups per CU)
    snprintf(buf, buflen, "gemmk_kernel<%d, %d, %d, %d>", ta, tb, two ? 2 : 3, (two || ta * tb <= 6) ? 2 : 1);
    return QAMD_OK;
  }
  if (p->kernel == 7) {
    const int ta = p->tile_cfg / 16, tb = p->tile_cfg % 16;
    snprintf(buf, buflen, (ta % 2 == 0 && tb % 2 == 0) ? "gemmh8_kernel<%d, %d> f16x3" : "gemmh_kernel<%d, %d> f16x3", ta, tb);
    return QAMD_OK;
  }
  if (p->kernel == 6) {
    snprintf(buf, buflen, "gemmd_kernel<%d, %d, %s, %s, %s> split_k=%d", p->tile_cfg / 16, p->tile_cfg % 16,
             p->a_kcontig ? "true" : "false", p->b_kcontig ? "true" : "false", p->c_ncontig ? "false" : "true", p->split_k);
    return QAMD_OK;
  }
  if (p->kernel == 1 || p->kernel == 2) {
    StreamArgs s;
    fill_stream_args(p, d, s);
    int PS = 0;
    uint32_t cpb = 0;
    if (sweep_config(p, s, p->vec_c, PS, cpb))
      snprintf(buf, buflen, "sweep_kernel<%s, %u, %d, %s>", T, s.NT, PS, s.zmode ? "true" : "false");
    else
      snprintf(buf, buflen, "stream_kernel<%s, %d, %u, 8, %s>", T, p->vec_c, s.NT, s.zmode ? "true" : "false");
  } else if (p->tile_cfg == kFastCfg || p->tile_cfg == kFastCfgN) {
    snprintf(buf, buflen, "gettf_kernel<%s, %d, %d, %d, %s, %s, %s>", T, p->tile_cfg == kFastCfg ? 4 : 2,
             p->vec_a >= 4 ? 4 : 2, p->vec_b >= 4 ? 4 : 2, p->a_kcontig ? "true" : "false",
             p->b_kcontig ? "true" : "false", p->c_ncontig ? "false" : "true");
  } else {
    static const char* cfg[kNumTileCfg] = {"2, 2, 4, 4, 16", "2, 2, 2, 2, 16", "4, 1, 4, 3, 16", "4, 1, 4, 1, 16",
                                           "4, 1, 2, 2, 16", "2, 2, 4, 4, 32", "fast", "fast"};
    snprintf(buf, buflen, "gett_kernel<%s, %s, %s> split_k=%d", T,
             (p->tile_cfg >= 0 && p->tile_cfg < kNumTileCfg) ? cfg[p->tile_cfg] : "?", p->c_ncontig ? "false" : "true",
             p->split_k);
  }
  return QAMD_OK;
}

// ---------------------------------------------------------------------------
// fused pair of streaming contractions
// ---------------------------------------------------------------------------
// the register-resident variant (chain2r.hip) serves fp32 D <= 6 at the 16-m chunk; the LDS-tile kernel (chain2.hip)
// everything else the fused pair covers: fp64, D = 7, results that are not 16-byte aligned.  QAMD_CHAIN2_FORCE_LDS in
// the plan's flags keeps the LDS-tile kernel for a shape the register kernels would take (tests, measurements)
static bool chain2_uses_registers(const qamd_chain2_plan* p, const void* C) {
  if (p->flags & QAMD_CHAIN2_FORCE_LDS) return false;
  if (!(p->flags & QAMD_CHAIN2_C_ALIGNED16) || ((uintptr_t)C & 15)) return false;   // 16-byte stores
  return qamd_chain2r_supported(p->dtype, p->D) && qamd_chain2_chunk(p->dtype, p->D) == 16;
}

// the 4x4x1 multi-block variant (chain2q.hip): fp32, D in {4, 6}, innermost m group a whole number of 64-m chunks,
// and at least one chunk for every wave of the persistent grid (flags: QAMD_CHAIN2_FORCE_REG never, QAMD_CHAIN2_FORCE_QUAD
// at any size).  The bar was four chunks per wave until round 4: the range-sliced last rows of a rank of 4 / 8
// (M = 139968 / 69984: 2187 / 1093 chunks) sat on chain2r below it, and chain2q runs them faster
static bool chain2_uses_quad(const qamd_chain2_plan* p, const void* C) {
  if (p->flags & QAMD_CHAIN2_FORCE_REG) return false;
  if (!chain2_uses_registers(p, C) || !qamd_chain2q_supported(p->dtype, p->D)) return false;
  if (p->nm < 1 || p->dim_m[p->nm - 1] % 64) return false;
  int64_t M = 1;
  for (int i = 0; i < p->nm; ++i) M *= p->dim_m[i];
  return (p->flags & QAMD_CHAIN2_FORCE_QUAD) || M / 64 >= 1024;
}

// chunks per workgroup of the fused-pair kernels: the largest divisor of the innermost m group's chunk count
// that still leaves ~12 workgroups per CU (``sc``: the retired super-chunk variant, always false)
static bool chain2_geometry(const qamd_chain2_plan* p, int ch, bool registers, uint32_t& chunks, uint32_t& cpb, bool& sc) {
  int64_t M = 1;
  for (int i = 0; i < p->nm; ++i) M *= p->dim_m[i];
  const int64_t inner = p->dim_m[p->nm - 1];
  chunks = (uint32_t)(M / ch);
  const uint32_t ic = (uint32_t)(inner / ch);
  const uint32_t target = std::max<uint32_t>(4, (chunks + 256 * 12 - 1) / (256 * 12));
  uint32_t best = 0, best_even = 0;
  for (uint32_t dlo = 1; (uint64_t)dlo * dlo <= ic; ++dlo) {
    if (ic % dlo) continue;
    uint32_t cand[2] = {dlo, ic / dlo};
    for (uint32_t c : cand) {
      if (c <= target && c > best) best = c;
      if (c % 2 == 0 && c <= 2 * target && c > best_even) best_even = c;
    }
  }
  if (best < 1) return false;
  cpb = best;
  sc = false;
  (void)best_even;
  (void)registers;
  return true;
}

extern "C" int qamd_chain2_describe(const qamd_chain2_plan* p, char* buf, int32_t buflen) {
  if (!p || !buf || buflen <= 0) return QAMD_EINVAL;
  const int ch = qamd_chain2_chunk(p->dtype, p->D);
  if (!ch) return QAMD_EUNSUPPORTED;
  const bool variant = (p->flags & (QAMD_CHAIN2_K1_SINGLE | QAMD_CHAIN2_NO_N2OUT)) != 0;
  if (variant && !chain2_uses_registers(p, nullptr)) return QAMD_EUNSUPPORTED;
  if (chain2_uses_quad(p, nullptr)) {
    snprintf(buf, buflen, "chain2q_kernel<%d, %d, %d>", p->D, (p->flags & QAMD_CHAIN2_K1_SINGLE) ? 1 : 2,
             (p->flags & QAMD_CHAIN2_NO_N2OUT) ? 0 : 1);
    return QAMD_OK;
  }
  if (chain2_uses_registers(p, nullptr))
  {
    uint32_t chunks = 0, cpb = 0;
    bool sc = false;
    if (p->nm < 1 || !chain2_geometry(p, ch, true, chunks, cpb, sc)) return QAMD_EUNSUPPORTED;
    snprintf(buf, buflen, "chain2r_kernel<%d, %d, %d, %s>", p->D, (p->flags & QAMD_CHAIN2_K1_SINGLE) ? 1 : 2,
             (p->flags & QAMD_CHAIN2_NO_N2OUT) ? 0 : 1, sc ? "true" : "false");
  }
  else
    snprintf(buf, buflen, "chain2_kernel<%s, %d, %d>", p->dtype == QAMD_F32 ? "float" : "double", p->D, ch / 16);
  return QAMD_OK;
}

extern "C" int qamd_contract_chain2(const qamd_chain2_plan* p, const void* A, const void* W1p, const void* W2p,
                                    void* C, const void* offK1_dev, const void* offCo_dev, const void* scale_a,
                                    const void* scale_1, const void* scale_2, void* absmax_out, void* stream) {
  if (qamdp_recording()) return qamdp_rec_chain2(p, A, W1p, W2p, C, offK1_dev, offCo_dev, scale_a, scale_1, scale_2, absmax_out);
  if (!p || !A || !W1p || !W2p || !C || !offK1_dev || !offCo_dev) return QAMD_EINVAL;
  if (p->nm < 1 || p->nm > QAMD_MAX_GROUPS) return QAMD_EINVAL;
  const int ch = qamd_chain2_chunk(p->dtype, p->D);
  if (!ch) return QAMD_EUNSUPPORTED;
  const int64_t DD = (int64_t)p->D * p->D;
  int64_t M = 1;
  for (int i = 0; i < p->nm; ++i) {
    if (p->dim_m[i] <= 0) return QAMD_EINVAL;
    M *= p->dim_m[i];
    if (M >= (1ll << 31)) return QAMD_EUNSUPPORTED;
  }
  const int64_t inner = p->dim_m[p->nm - 1];
  if (p->sa_m[p->nm - 1] != 1 || p->sc_m[p->nm - 1] != DD || inner % ch) return QAMD_EUNSUPPORTED;
  Chain2Args a;
  memset(&a, 0, sizeof(a));
  a.nm = p->nm;
  for (int i = 0; i < p->nm; ++i) { a.dim_m[i] = (uint32_t)p->dim_m[i]; a.sa_m[i] = p->sa_m[i]; a.sc_m[i] = p->sc_m[i]; }
  a.sa_v = p->sa_v;
  {
    bool sc = false;
    if (!chain2_geometry(p, ch, chain2_uses_registers(p, C), a.chunks, a.chunks_per_block, sc)) return QAMD_EUNSUPPORTED;
    a.sc = sc ? 1 : 0;
  }
  a.grid = a.chunks / a.chunks_per_block;
  const int k1_single = (p->flags & QAMD_CHAIN2_K1_SINGLE) ? 1 : 0, no_n2out = (p->flags & QAMD_CHAIN2_NO_N2OUT) ? 1 : 0;
  if (k1_single && no_n2out) return QAMD_EUNSUPPORTED;
  if (p->flags & QAMD_CHAIN2_W_STRIDED) {
    if (!chain2_uses_registers(p, C)) return QAMD_EUNSUPPORTED;   // the LDS-tile kernel wants packed W
    for (int i = 0; i < 4; ++i) { a.w1s[i] = p->w1_strides[i]; a.w2s[i] = p->w2_strides[i]; }
  } else {   // dense packed copies: [k1][x][y] and [y][v][n2_out][n2_in]
    const int64_t N2 = (no_n2out ? 1 : p->D) * (int64_t)p->D;
    a.w1s[0] = k1_single ? DD : (int64_t)p->D * DD; a.w1s[1] = DD; a.w1s[2] = p->D; a.w1s[3] = 1;
    a.w2s[0] = (int64_t)p->D * N2; a.w2s[1] = N2; a.w2s[2] = p->D; a.w2s[3] = 1;
  }
  if (chain2_uses_quad(p, C)) {
    a.chunks = (uint32_t)(M / 64);
    a.chunks_per_block = 0;
    a.grid = std::min<uint32_t>((a.chunks + 3) / 4, 256);   // persistent: one workgroup (4 waves, 1 per SIMD) per CU
    return qamd_chain2q_launch(p->D, k1_single, no_n2out, &a, A, W1p, W2p, C, offK1_dev, offCo_dev, scale_a, scale_1,
                               scale_2, absmax_out, stream);
  }
  if (chain2_uses_registers(p, C))
    return qamd_chain2r_launch(p->D, k1_single, no_n2out, &a, A, W1p, W2p, C, offK1_dev, offCo_dev, scale_a, scale_1,
                               scale_2, absmax_out, stream);
  if (k1_single || no_n2out) return QAMD_EUNSUPPORTED;   // the row-start / row-end shapes exist in chain2r only
  return qamd_chain2_launch(p->dtype, p->D, &a, A, W1p, W2p, C, offK1_dev, offCo_dev, scale_a, scale_1, scale_2,
                            absmax_out, stream);
}

// ---------------------------------------------------------------------------
// one row of a 2D boundary sweep in one launch (rowpass.hip)
// ---------------------------------------------------------------------------
extern "C" int qamd_rowpass_supported(int32_t dtype, int32_t D, int32_t nsites) {
  return dtype == QAMD_F32 && D == 6 && nsites == 5;
}

extern "C" int qamd_contract_rowpass(const qamd_rowpass_plan* p, const void* A, const void* const* W, void* C,
                                     const void* scale_a, const void* const* scale_w, void* absmax_out, void* stream) {
  if (!p || !W) return QAMD_EINVAL;
  if (qamdp_recording()) return qamdp_rec_rowpass(p, A, W, C, scale_a, scale_w, absmax_out);
  if (!C || (!A && p->nS >= 0)) return QAMD_EINVAL;
  if (!qamd_rowpass_supported(p->dtype, p->D, p->nsites) || p->nS < -1 || p->nS > 4) return QAMD_EUNSUPPORTED;
  RowArgs a;
  memset(&a, 0, sizeof(a));
  bool full = true;
  for (int i = 0; i < 5; ++i) {
    const int32_t e = p->ed[i] ? p->ed[i] : p->D;
    if (e < 1 || e > p->D) return QAMD_EINVAL;
    a.ed[i] = (uint32_t)e;
    full = full && e == p->D;
  }
  {
    const int32_t e = p->eh ? p->eh : p->D;
    if (e < 1 || e > p->D) return QAMD_EINVAL;
    a.eh = (uint32_t)e;
    full = full && e == p->D;
  }
  int64_t items = a.ed[0];
  for (int i = 0; i < 5; ++i) {
    if (!W[i]) return QAMD_EINVAL;
    a.sv[i] = p->sv[i];
    a.sd[i] = p->sd[i];
    for (int j = 0; j < 4; ++j) a.ws[i][j] = p->w_strides[i][j];
  }
  a.sh = p->sh;
  a.nS = p->nS;
  int64_t a_span = 1;       // elements the boundary tensor spans: rowq addresses it with 32-bit per-lane offsets
  for (int i = 0; i < 5; ++i) a_span += (p->D - 1) * (p->sv[i] < 0 ? -p->sv[i] : p->sv[i]);
  for (int g = 0; g < p->nS; ++g) {
    if (p->dim_s[g] <= 0 || p->dim_s[g] >= (1ll << 31)) return QAMD_EINVAL;
    a.dimS[g] = (uint32_t)p->dim_s[g];
    a.sSa[g] = p->sa_s[g];
    a.sSc[g] = p->sc_s[g];
    items *= p->dim_s[g];
    if (items >= (1ll << 31)) return QAMD_EUNSUPPORTED;
  }
  a.items = p->nS < 0 ? 0 : (uint32_t)items;
  int kernel = p->kernel;
  if (p->nS < 0) kernel = 1;                               // the first row: rowfirst_kernel lives in rowpass.hip
  else if (kernel == 0) kernel = 2;
  if (kernel == 1 && !full) return QAMD_EUNSUPPORTED;
  if (kernel >= 2 && kernel <= 5) {
    for (int i = 0; i < 5; ++i)
      if (p->sv[i] < 0) return QAMD_EUNSUPPORTED;
    if (a_span >= (1ll << 31)) return QAMD_EUNSUPPORTED;
    if (kernel == 3) a.pad2_ = 512;      // the per-stream item queue instead of equal static shares
    if (kernel == 4) a.pad2_ = 1024;     // wave priorities by (workgroup / 8) % 3 (experiments)
    if (kernel == 5) a.pad2_ = 128;      // wave priorities by (workgroup / 256) % 3 (experiments)
  } else if (kernel != 1) {
    return QAMD_EUNSUPPORTED;
  }
  const int rc = kernel >= 2 ? qamd_rowq_launch(&a, A, W, C, scale_a, scale_w, absmax_out, stream)
                             : qamd_rowpass_launch(&a, A, W, C, scale_a, scale_w, absmax_out, stream);
  return rc == 0 ? QAMD_OK : (rc == -2 ? QAMD_EUNSUPPORTED : QAMD_ELAUNCH);
}

// ---------------------------------------------------------------------------
// device-resident tree of small contractions (microtree.hip)
// ---------------------------------------------------------------------------
extern "C" int qamd_microtree_launch(int dtype, const qamd_micro_step* steps_dev, int nsteps, const int32_t* etab,
                                     const int32_t* ktab, const void* const* inputs_dev, int ninputs, void* arena_dev,
                                     int64_t arena_elems, void* out_dev, int64_t out_elems, int ninst, int wide,
                                     void* stream);

extern "C" int qamd_microtree_run_ex(int32_t dtype, const qamd_micro_step* steps_dev, int32_t nsteps, const int32_t* etab_dev,
                                     const int32_t* ktab_dev, const void* const* inputs_dev, int32_t ninputs, void* arena_dev,
                                     int64_t arena_elems, void* out_dev, int64_t out_elems, int32_t ninst, int32_t flags,
                                     void* stream) {
  if (!steps_dev || !etab_dev || !ktab_dev || !inputs_dev || !out_dev || nsteps <= 0 || ninputs <= 0 || ninst <= 0)
    return QAMD_EINVAL;
  if (dtype < 0 || dtype > 3 || (flags & ~QAMD_MICRO_WIDE)) return QAMD_EUNSUPPORTED;
  const int wide = (flags & QAMD_MICRO_WIDE) && (dtype == QAMD_F32 || dtype == QAMD_C64);
  if (!arena_dev && arena_elems * (int64_t)kEsize[dtype] * (wide ? 2 : 1) > QAMD_MICRO_LDS_ARENA_BYTES) return QAMD_EINVAL;
  int rc = qamd_microtree_launch(dtype, steps_dev, nsteps, etab_dev, ktab_dev, inputs_dev, ninputs, arena_dev, arena_elems,
                                 out_dev, out_elems, ninst, wide, stream);
  return rc == 0 ? QAMD_OK : (rc == -2 ? QAMD_EUNSUPPORTED : QAMD_ELAUNCH);
}

extern "C" int qamd_microtree_run(int32_t dtype, const qamd_micro_step* steps_dev, int32_t nsteps, const int32_t* etab_dev,
                                  const int32_t* ktab_dev, const void* const* inputs_dev, int32_t ninputs, void* arena_dev,
                                  int64_t arena_elems, void* out_dev, int64_t out_elems, int32_t ninst, void* stream) {
  return qamd_microtree_run_ex(dtype, steps_dev, nsteps, etab_dev, ktab_dev, inputs_dev, ninputs, arena_dev, arena_elems,
                               out_dev, out_elems, ninst, 0, stream);
}
