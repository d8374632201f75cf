// Kernel-argument blocks shared by gett.hip (device) and api.cpp (host planner).
#pragma once
#include <stdint.h>

#define QAMD_G 8
#define QAMD_SLOTS 64  // absmax slots per tensor (sharded atomicMax targets)

struct GettArgs {
  int32_t nb, nm, nn, nk;
  uint32_t dim_b[QAMD_G], dim_m[QAMD_G], dim_n[QAMD_G];
  int64_t sa_b[QAMD_G], sb_b[QAMD_G], sc_b[QAMD_G];
  int64_t sa_m[QAMD_G], sc_m[QAMD_G];
  int64_t sb_n[QAMD_G], sc_n[QAMD_G];
  uint32_t B, M, N, K;
  uint32_t Kpad;  // stride between the A and B halves of the k-offset table
  uint32_t Kloop; // K rounded up to this kernel's k-tile
  uint32_t Kc;    // k range per split (multiple of the k-tile)
  uint32_t tiles_m, tiles_n, split_k;
  int32_t vec_a, vec_b, a_kcontig, b_kcontig;
  int64_t slab_stride;  // elements between split-K slabs
  int64_t sa_k0, sb_k0; // gemmk.hip: strides of the single K group
  int32_t vec_c;        // gemmk.hip: elements of C that are contiguous and aligned along n (1, 2, 4)
  int32_t pad_;
  // gemmd.hip: the K groups themselves (k offsets are computed in registers, not read from the k-offset table: a
  // table read in the request path would tie the request counter to ordinary loads)
  uint32_t dim_k[QAMD_G];
  int64_t sa_k[QAMD_G], sb_k[QAMD_G];
};

// "big tensor x small tensor" streaming kernel (stream.hip)
struct StreamArgs {
  int32_t nm, nn;
  uint32_t dim_m[QAMD_G], dim_n[QAMD_G];
  int64_t sa_m[QAMD_G], sc_m[QAMD_G];
  int64_t sb_n[QAMD_G], sc_n[QAMD_G];
  uint32_t M, N, K;
  uint32_t KS;       // k-steps of 4
  uint32_t Kpad;     // 4 * KS
  uint32_t KpadTab;  // stride between the A and B halves of the k-offset table
  uint32_t NT;       // n-tiles of 16
  uint32_t chunks, chunks_per_wave, grid;
  uint32_t aligned;       // innermost M group is a whole number of chunks (wave-uniform offsets)
  uint32_t inner_chunks;  // chunks per innermost M group (aligned mode)
  uint32_t zmode;         // 1: C[.., m, n_in] with n_in stride-1 -> LDS-transposed stores
  uint32_t d_in;          // zmode: size of the innermost N group
  int64_t sc_m_in;        // C stride of the innermost M group
  uint32_t c_break;       // zmode: the innermost M group of C (l_in >= a chunk, not a multiple of it) may end inside a chunk;
  uint32_t l_in;          //        ``aligned`` / ``inner_chunks`` then describe the (longer) contiguous run of A
  uint32_t zb_groups;     // c_break: entries of a workgroup's table of group starts (0 otherwise)
  uint32_t pad2_;
};

// fused pair of streaming contractions (chain2.hip)
struct Chain2Args {
  int32_t nm;
  uint32_t dim_m[QAMD_G];
  int64_t sa_m[QAMD_G], sc_m[QAMD_G];
  int64_t sa_v;  // A stride of the carried index v
  uint32_t chunks, chunks_per_block, grid;
  int64_t w1s[4];   // W1 element strides of (k1 outer, k1 inner, x, y); a single-index k1 uses w1s[0] only
  int64_t w2s[4];   // W2 element strides of (y, v, n2_out, n2_in)
  uint32_t sc;      // chain2r: waves work on pairs of adjacent chunks (chunks_per_block is even)
  uint32_t ablate;  // debug/ablation bits (QAMD_CHAIN2_ABLATE): 1 no stores, 2 no stage-1 scatter, 4 no loads, 8 no stage 2
};

// few rows x one long vector (dotm.hip)
struct DotArgs {
  int32_t S;            // rows (<= 32)
  int64_t K;            // length of the contraction, stride 1 in both operands
  int64_t row_off[32];  // element offset of every row
  uint32_t grid;        // workgroups = slabs of partial sums
};

// one row of a 2D boundary sweep in one launch (rowpass.hip): five sites, bond dimension 6
struct RowArgs {
  int64_t sv[5];        // strides (elements) of the five up legs v1..v5 in the boundary tensor
  int64_t sd[5], sh;    // strides of the new down legs d1..d5 and of the row's new open leg h in the result
  int32_t nS, pad_;     // spectator groups (every other index of the boundary tensor), outermost first
  uint32_t dimS[4];
  int64_t sSa[4], sSc[4];
  int64_t ws[5][4];     // site tensor strides of (up, left, down, right); site 0 has no left leg, site 4's right leg is h
  uint32_t items, pad2_;   // (number of S values) x (extent of d1)
  uint32_t ed[5], eh;      // rowq.hip: extents (<= 6) of the new down legs d1..d5 and of the open leg h
};

// one operand of a split-product join (gemmh.hip): the fp32 source's free bundle and k stride, the image's padded extents
struct SplitArgs {
  int32_t ng, pad_;
  uint32_t dim[QAMD_G];
  int64_t stride[QAMD_G];
  int64_t sk;            // element stride of k in the source (nk <= 1)
  uint32_t X, Xpad;      // free extent, and padded to whole workgroup tiles
  uint32_t K, KG;        // contraction extent, and k-groups of 8 in the image (K rounded up to 32, / 8)
  int32_t nk, period;    // nk > 1: the contraction bundle in several groups (outermost first), k -> offset by decomposition;
                         // period 2: one centring constant per PARITY of k (an innermost K group of two: re / im interleaved); else 1
  uint32_t dim_k[QAMD_G];
  int64_t stride_k[QAMD_G];
};

struct KtabArgs {
  int32_t nk;
  uint32_t K, Kpad;
  uint32_t dim_k[QAMD_G];
  int64_t sa_k[QAMD_G], sb_k[QAMD_G];
};

#ifdef __cplusplus
extern "C" {
#endif
int qamd_gett_launch(int dtype, int cfg, const GettArgs* a, int swap, const void* A, const void* B,
                     void* C, const void* ktab, const void* scale_a, const void* scale_b, void* absmax_out,
                     void* stream);
int qamd_gettf_launch(int dtype, int bn, const GettArgs* a, int swap, const void* A, const void* B, void* C,
                      const void* ktab, const void* scale_a, const void* scale_b, void* absmax_out, void* stream);
int qamd_gemmk_launch(int ta, int tb, const GettArgs* a, const void* A, const void* B, void* C,
                      const void* scale_a, const void* scale_b, void* absmax_out, void* stream);
int64_t qamd_gemmh_image_bytes(int64_t xpad, int64_t kpad);
int64_t qamd_gemmh_mean_bytes(int64_t xpad);
int qamd_gemmh_absmax_launch(const SplitArgs* a, const void* X, void* slots, void* stream);
int qamd_gemmh_split_launch(const SplitArgs* a, const void* X, const void* slots, void* hdr, void* P, void* mean, void* stream);
int qamd_gemmh_launch(int ta, int tb, const GettArgs* a, const void* PA, const void* PB, void* C, const void* scale_a,
                      const void* scale_b, const void* hdrA, const void* hdrB, const void* meanA, const void* meanB,
                      void* absmax_out, void* stream);
int qamd_gemmh_dot_launch(int ta, int tb, const GettArgs* a, const void* PA, const void* PB, const void* T,
                          const void* hdrA, const void* hdrB, const void* meanA, const void* meanB, void* partial, void* stream);
int qamd_dotm_launch(int dtype, const DotArgs* a, const void* R, const void* v, void* slab, void* C,
                     const void* scale_a, const void* scale_b, void* absmax_out, void* stream);
int qamd_stream_launch(int dtype, int V, const StreamArgs* a, const void* A, const void* B, void* C,
                       const void* ktab, const void* scale_a, const void* scale_b, void* absmax_out,
                       void* stream);
int qamd_rowpass_launch(const RowArgs* a, const void* A, const void* const* W, void* C, const void* scale_a,
                        const void* const* scale_w, void* absmax_out, void* stream);
int qamd_rowq_launch(const RowArgs* a, const void* A, const void* const* W, void* C, const void* scale_a,
                     const void* const* scale_w, void* absmax_out, void* stream);
int qamd_sweep_launch_f32(int PS, const StreamArgs* a, const void* A, const void* B, void* C, const void* ktab,
                          const void* scale_a, const void* scale_b, void* absmax_out, void* stream);
int qamd_sweep_launch_f64(int PS, const StreamArgs* a, const void* A, const void* B, void* C, const void* ktab,
                          const void* scale_a, const void* scale_b, void* absmax_out, void* stream);
int qamd_chain2_launch(int dtype, int D, const Chain2Args* a, const void* A, const void* W1p, const void* W2p,
                       void* C, const void* offK1, const void* offCo, const void* scale_a, const void* scale_1,
                       const void* scale_2, void* absmax_out, void* stream);
int qamd_chain2r_supported(int dtype, int D);
int qamd_chain2r_launch(int D, int k1_single, int no_n2out, const Chain2Args* a, const void* A, const void* W1p,
                        const void* W2p, void* C, const void* offK1, const void* offCo, const void* scale_a,
                        const void* scale_1, const void* scale_2, void* absmax_out, void* stream);
int qamd_chain2q_supported(int dtype, int D);
int qamd_chain2q_launch(int D, int k1_single, int no_n2out, const Chain2Args* a, const void* A, const void* W1p,
                        const void* W2p, void* C, const void* offK1, const void* offCo, const void* scale_a,
                        const void* scale_1, const void* scale_2, void* absmax_out, void* stream);
void qamd_gett_tile_dims(int cfg, int* bm, int* bn, int* bk);
int qamd_splitk_reduce_launch(int dtype, void* C, const void* ws, int64_t n, int split_k,
                              const void* scale_a, const void* scale_b, void* absmax_out, void* stream);
int qamd_build_ktab_launch(void* ktab, const KtabArgs* a, void* stream);
#ifdef __cplusplus
}
#endif
