/*
 * quimb_amd.h -- C-ABI of libquimb_amd.so, the MI355X (gfx950) contraction
 * backend that sits underneath quimb's array-backend boundary.
 *
 * Every entry point replaces an L0 array-library call that the reference's
 * hot path bottoms out in.  Citations are to
 * /root/reference (quimb) -- the arithmetic itself lives in numpy/cotengra,
 * which quimb reaches through these call sites:
 *
 *   qamd_contract_pair   <- do("tensordot", ...) quimb/tensor/tensor_core.py:3793
 *                           ctg.array_contract   quimb/tensor/contraction.py:285
 *                           (cotengra lowers every pairwise step to
 *                            permute/reshape + (batched) matmul; here the
 *                            permutes are folded into the operand addressing)
 *   qamd_permute         <- do("transpose")/do("reshape") in the composed
 *                           `fuse`  quimb/tensor/array_ops.py:148-182,
 *                           Tensor.transpose tensor_core.py:2743,
 *                           Tensor.isel (take / getitem) tensor_core.py:2260-2348
 *   qamd_strip_exponent  <- strip_exponent=True contract semantics,
 *                           tensor_core.py:330-340, tests/test_tensor/test_contract.py:8-19
 *   qamd_reduce_sum      <- single-operand einsum terms ("ab->a") that
 *                           cotengra preprocesses before a pairwise step
 *   qamd_binary          <- do("multiply") for pure hyper-index steps,
 *                           slice accumulation (sum over sliced indices)
 *   qamd_scale / qamd_conj / qamd_cast
 *                        <- Tensor.__mul__/__truediv__ (tensor_core.py:3771,3813),
 *                           Tensor.conj, Tensor.astype (tensor_core.py:2716)
 *
 * Conventions: plain pointers and sizes only, no C++/torch types.  All
 * extents/strides are int64 in units of ELEMENTS.  Every call is
 * stream-ordered on the hipStream_t passed (as void*), never synchronises,
 * never allocates device memory: the caller owns every buffer.  Return value:
 * 0 on success, negative QAMD_E* code otherwise (no exceptions cross the ABI).
 */
#ifndef QUIMB_AMD_H
#define QUIMB_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QAMD_ABI_VERSION 1
#define QAMD_MAX_GROUPS 8   /* fused index groups per bundle            */
#define QAMD_MAX_NDIM 32    /* max rank accepted by permute/binary/reduce */

typedef enum {
  QAMD_F32 = 0,
  QAMD_F64 = 1,
  QAMD_C64 = 2,
  QAMD_C128 = 3
} qamd_dtype;

enum {
  QAMD_OK = 0,
  QAMD_EINVAL = -1,    /* malformed plan / argument                    */
  QAMD_EUNSUPPORTED = -2,
  QAMD_EWORKSPACE = -3, /* workspace too small                          */
  QAMD_ELAUNCH = -4    /* hipGetLastError() != success after launch     */
};

/*
 * Pairwise contraction plan ("GETT" form: GEMM with tensor addressing).
 *
 *   C[b, m, n] = sum_k A[b, m, k] * B[b, k, n]
 *
 * b, m, n, k are *bundles* of fused index groups.  A linear bundle index is
 * decomposed mixed-radix over dim_x[0..nx-1] (group nx-1 fastest) and dotted
 * with the per-operand strides to give an element offset.  No operand is ever
 * physically permuted.
 */
typedef struct {
  int32_t dtype;     /* qamd_dtype                                       */
  int32_t nb, nm, nn, nk;
  int32_t conj_a, conj_b; /* complex only                                */
  int32_t kernel;    /* filled by finalize: 0 tiled GETT, 1 streaming (big x small), 2 streaming with LDS-transposed stores, 4 few-rows x long-vector reduction (split_k slabs of workspace), 5 k-outer MFMA GETT (gemmk.hip: GEMM-shaped joins, fp32, both free bundles stride-1; tile_cfg = 16 ta + tb names the (64 ta) x (64 tb) workgroup tile; pinned on input with kernel = -5 and tile_cfg = 16 ta + tb), 6 fp64 MFMA GETT on an LDS-DMA ring (gemmd.hip: GEMM-shaped fp64 contractions with either operand free- or k-contiguous; tile_cfg = 16 ta + tb names the (32 ta) x (64 tb) workgroup tile, split_k the number of k slabs; pinned on input with kernel = -6 and tile_cfg = 16 ta + tb), 7 fp32 GEMM-shaped pairs (no batch bundle, K, M, N >= 256) as SPLIT PRODUCTS on the f16 matrix pipe (gemmh.hip, OPT-IN, never chosen automatically: kernel = -7 on input asks for it exactly where the automatic choice would have been kernel 5, kernel = -8 on every operand layout -- the operands are re-laid-out by a split pass; complex pairs' real expansions included --: every fp32 operand is scaled by a power of two and written as the sum of two fp16 numbers -- 2^-24 relative --, the three products a1 b1 + a1 b2 + a2 b1 are exact and accumulate in fp32; the workspace holds the operands' split images: qamd_pair_workspace_bytes).  ON INPUT: 0 automatic, -1 force 0, -2 automatic without kernels 5 / 6, -5 / -6 as above, -7 / -8 kernel 7 where it applies and the automatic choice elsewhere -- the library reads no environment variable, these fields are the only way to steer the choice */
  int64_t dim_b[QAMD_MAX_GROUPS], sa_b[QAMD_MAX_GROUPS], sb_b[QAMD_MAX_GROUPS], sc_b[QAMD_MAX_GROUPS];
  int64_t dim_m[QAMD_MAX_GROUPS], sa_m[QAMD_MAX_GROUPS], sc_m[QAMD_MAX_GROUPS];
  int64_t dim_n[QAMD_MAX_GROUPS], sb_n[QAMD_MAX_GROUPS], sc_n[QAMD_MAX_GROUPS];
  int64_t dim_k[QAMD_MAX_GROUPS], sa_k[QAMD_MAX_GROUPS], sb_k[QAMD_MAX_GROUPS];
  /* tuning hints filled by qamd_pair_plan_finalize (may be overridden)   */
  int32_t tile_cfg;  /* index into the kernel table, -1 = auto           */
  int32_t split_k;   /* >=1                                              */
  int32_t vec_a, vec_b; /* 1,2,4: elements per contiguous global load    */
  int32_t a_kcontig, b_kcontig; /* 1: stride-1 index lives in the K bundle */
  int32_t c_ncontig; /* 1: C's stride-1 index lives in the N bundle      */
  int32_t vec_c;     /* streaming kernel: elements per contiguous C store */
} qamd_pair_plan;

/*
 * Fused exponent stripping (strip_exponent=True, tensor_core.py:330-340).
 * Every tensor that takes part owns QAMD_ABSMAX_SLOTS values of the plan's REAL
 * dtype holding partial max|x| (sharded atomicMax targets; the true max is the
 * max over the slots).  A contraction multiplies its result by
 * 1 / (max|A| * max|B|) -- i.e. consumes operands as if they had been divided by
 * their max, which is what the reference does right after producing them -- and
 * reduces max|C| into absmax_out (caller zeroes it before the launch).  Any
 * pointer may be NULL (= scale 1 / no reduction).
 */
#define QAMD_ABSMAX_SLOTS 64
typedef struct {
  const void* scale_a;
  const void* scale_b;
  void* absmax_out;
} qamd_epilogue;

int qamd_abi_version(void);
const char* qamd_build_info(void);

/* Validate the plan, pick tile config / split-K / vector widths. */
int qamd_pair_plan_finalize(qamd_pair_plan* plan, int64_t align_a_bytes, int64_t align_b_bytes,
                            int64_t align_c_bytes);
/* Number of int64 entries the K-offset table needs (2 * padded K). */
int64_t qamd_pair_ktab_len(const qamd_pair_plan* plan);
/* Fill the K-offset table (device memory, int64[qamd_pair_ktab_len]). */
int qamd_pair_build_ktab(const qamd_pair_plan* plan, void* ktab_dev, void* stream);
/* Bytes of scratch needed by qamd_contract_pair (split-K slabs), may be 0. */
int64_t qamd_pair_workspace_bytes(const qamd_pair_plan* plan);
/* C = A . B (overwrites C). */
int qamd_contract_pair(const qamd_pair_plan* plan, const void* A, const void* B, void* C,
                       const void* ktab_dev, void* workspace, int64_t workspace_bytes, void* stream);

/* Name of the kernel instantiation a finalized plan launches (for profiles/benchmarks). */
int qamd_pair_describe(const qamd_pair_plan* plan, char* buf, int32_t buflen);
/* Same with the fused exponent-stripping epilogue (ep may be NULL). */
int qamd_contract_pair_ex(const qamd_pair_plan* plan, const void* A, const void* B, void* C,
                          const void* ktab_dev, void* workspace, int64_t workspace_bytes,
                          const qamd_epilogue* ep, void* stream);
/*
 * Fused PAIR of streaming contractions (two adjacent site absorptions of a boundary
 * sweep, quimb/tensor/tn2d/core.py:1402 twice) -- the intermediate never touches HBM:
 *
 *   X[x, y, v, m] = sum_k1     A[k1, v, m] * W1[k1, (x, y)]
 *   C[x, n2, m]   = sum_{y,v}  X[x, y, v, m] * W2[(y, v), n2]        n2 = (n2_out, n2_in)
 *
 * every index size is D (k1 is D*D).  A: element offset of row k1 from offK1_dev[k1],
 * v at stride sa_v, m over the bundle (dim_m, sa_m) with the innermost group stride-1
 * and a multiple of qamd_chain2_chunk().  W1p / W2p: dense [D*D][D*D] device copies laid
 * out [k1][x*D + y] and [y*D + v][n2_out*D + n2_in].  C: the innermost m group has
 * stride D*D and is followed by the contiguous block [x][n2_in]; n2_out at element
 * offset offCo_dev[n2_out].  scale_* / absmax_out: slots as described for the epilogue struct above; any may be NULL.
 *
 * Which kernel runs is decided inside, from the plan alone (qamd_chain2_describe names it): chain2q (fp32, D = 6 / 4,
 * innermost m group a multiple of 64, >= 1024 chunks: v_mfma_f32_4x4x1_16b, one wave per SIMD), chain2r (fp32, D <= 6:
 * v_mfma_f32_16x16x4, 16-m chunks) or chain2 (LDS tile: fp64, D = 7, results that are not 16-byte aligned).  The
 * QAMD_CHAIN2_FORCE_* flag bits pin one of them for a shape another would take; no environment variable is read.
 */
#define QAMD_CHAIN2_C_ALIGNED16 1
/* row-start shape: k1 is ONE index of size D (offK1_dev has D entries, W1p is [D][D*D]) */
#define QAMD_CHAIN2_K1_SINGLE 2
/* row-end shape: n2 = n2_in only (W2p is [D*D][D], offCo_dev has one entry, C ends [.., m_inner, x, n2_in]) */
#define QAMD_CHAIN2_NO_N2OUT 4
/* W1p / W2p are the ORIGINAL small tensors, addressed with w1_strides (k1 groups outermost first, x, y) and
 * w2_strides (y, v, n2_out, n2_in) in elements -- no packed copies (register kernel only) */
#define QAMD_CHAIN2_W_STRIDED 8
/* explicit kernel pins (tests, measurements): the LDS-tile kernel / never the quad kernel / the quad kernel at any size */
#define QAMD_CHAIN2_FORCE_LDS 16
#define QAMD_CHAIN2_FORCE_REG 32
#define QAMD_CHAIN2_FORCE_QUAD 64
typedef struct {
  int32_t dtype, D, nm;
  int32_t flags;   /* QAMD_CHAIN2_C_ALIGNED16: every offCo_dev entry and every sc_m of the outer m groups is a multiple of 4
                      elements (with a 16-byte aligned C this lets fp32 use the register-resident kernel) */
  int64_t dim_m[QAMD_MAX_GROUPS], sa_m[QAMD_MAX_GROUPS], sc_m[QAMD_MAX_GROUPS];
  int64_t sa_v;
  int64_t w1_strides[4], w2_strides[4];   /* read only with QAMD_CHAIN2_W_STRIDED */
} qamd_chain2_plan;
/* m-chunk the fused kernel works in for (dtype, D); 0 = combination not supported */
int qamd_chain2_chunk(int32_t dtype, int32_t D);
/* kernel instantiation the fused pair would run (matches rocprofv3's kernel names) */
int qamd_chain2_describe(const qamd_chain2_plan* plan, char* buf, int32_t buflen);
int qamd_contract_chain2(const qamd_chain2_plan* plan, const void* A, const void* W1p, const void* W2p, void* C,
                         const void* offK1_dev, const void* offCo_dev, const void* scale_a, const void* scale_1,
                         const void* scale_2, void* absmax_out, void* stream);

/*
 * One ROW of a 2D boundary sweep in ONE launch (csrc/rowpass.hip).  quimb absorbs a row into the boundary tensor site by
 * site -- quimb/tensor/tn2d/core.py:1393-1402, in exact mode five dependent pairwise contractions per row of a 5-wide
 * block -- which for the small rows of a corner sweep is dispatch latency, not bandwidth.  This entry is those five steps:
 *
 *   C[S, d1..d5, h] = sum_{v1..v5, b1..b4} A[S, v1..v5] W[0][v1, d1, b1] W[1][v2, b1, d2, b2] ... W[4][v5, b4, d5, h]
 *
 * fp32, nsites = 5, every up leg and bond of the site tensors of size D = 6 (qamd_rowpass_supported says what else is
 * served: nothing yet); the NEW legs may be shorter: ed[c] <= D values of d_c, eh <= D of h (0 = D) -- the range-sliced cut
 * bonds of one rank's share of a sharded contraction.  kernel: 0 = the library's choice, 1 = csrc/rowpass.hip (16x16x4 tiles,
 * two LDS images; needs every extent = D), 2 = csrc/rowq.hip (4x4x1 multi-block MFMA, one in-place image; rows of any size;
 * persistent workgroups with equal static shares of the work items), 3 = the same with a per-(device, stream) item queue
 * (the only entry that keeps state between calls: a 64-byte counter slot per stream, re-armed by the launch itself).  S = every other index of A, as up to 4 groups (dim_s, A strides sa_s, C strides sc_s, outermost first); sv / sd / sh:
 * element strides of the up legs in A and of the new down legs / the row's new open leg in C; w_strides[c] = element strides
 * of site c's (up, left bond, down, right bond) legs in W[c] (site 0 has no left bond, site 4's right bond is h).  A, C and
 * the W[c] are read and written in place at those strides: the call consumes and produces exactly the layouts the five
 * separate steps would have.  scale_a / scale_w[c] / absmax_out: absmax slots as for the epilogue struct above (any NULL;
 * scale_w itself may be NULL): C is scaled by 1 / (max|A| max|W[0]| ... max|W[4]|); the intermediates never exist.
 * nS = -1: the FIRST row of a sweep -- no boundary tensor (A ignored, may be NULL), the site tensors have no up legs:
 * C[d1..d5, h] = sum_b W[0][d1, b1] W[1][b1, d2, b2] ... W[4][b4, d5, h] (w_strides[c][0] unused).
 */
typedef struct {
  int32_t dtype, D, nsites, nS;
  int64_t sv[5], sd[5], sh;
  int64_t dim_s[4], sa_s[4], sc_s[4];
  int64_t w_strides[5][4];
  int32_t ed[5], eh;      /* extents of the new legs d1..d5, h (0 = D) */
  int32_t kernel, pad_;   /* 0 auto, 1 rowpass_kernel, 2 rowq_kernel, 3 rowq_kernel with the item queue, 4 / 5 rowq_kernel priority experiments */
} qamd_rowpass_plan;
int qamd_rowpass_supported(int32_t dtype, int32_t D, int32_t nsites);
int qamd_contract_rowpass(const qamd_rowpass_plan* plan, const void* A, const void* const* W, void* C, const void* scale_a,
                          const void* const* scale_w, void* absmax_out, void* stream);

/*
 * slots: n_tensors x QAMD_ABSMAX_SLOTS values (float for F32/C64, double
 * otherwise).  *out_dev (double, device) = sum_t log10(max over tensor t's slots),
 * tensors whose max is 0 are skipped.
 */
int qamd_absmax_log10_sum(const void* slots, int64_t n_tensors, int32_t dtype, void* out_dev, void* stream);
/* the same sum ADDED (atomically) to *acc_dev: the exponent accumulator of a contraction, which independent branches on
 * other streams also add their un-fused strips to */
int qamd_absmax_log10_sum_add(const void* slots, int64_t n_tensors, int32_t dtype, void* acc_dev, void* stream);
/* x[i] /= max(slots[0..QAMD_ABSMAX_SLOTS))  (no-op if that max is 0) */
int qamd_div_by_absmax(void* x, int64_t n, const void* slots, int32_t dtype, void* stream);

/*
 * dst (C-contiguous, given shape) <- src viewed with arbitrary element
 * strides + element offset.  Covers transpose, index fusion (transpose +
 * reshape), isel/take slices, diagonals (summed strides), broadcast (stride 0).
 */
int qamd_permute(void* dst, const void* src, int32_t ndim, const int64_t* shape,
                 const int64_t* src_strides, int64_t src_offset, int32_t dtype, void* stream);

/* out[o] = sum_r x[off_o(o) + off_r(r)];  out is C-contiguous over the kept dims. */
int qamd_reduce_sum(void* out, const void* x, int32_t ndim_keep, const int64_t* shape_keep,
                    const int64_t* strides_keep, int32_t ndim_red, const int64_t* shape_red,
                    const int64_t* strides_red, int32_t dtype, void* stream);

/* out (C-contiguous, shape) = a[view] (op) b[view];  op: 0 add, 1 mul, 2 sub, 3 true division
 * (do("divide") / Tensor.__truediv__ with an array divisor, tensor_core.py:3813). */
int qamd_binary(void* out, const void* a, const int64_t* a_strides, const void* b,
                const int64_t* b_strides, int32_t ndim, const int64_t* shape, int32_t op,
                int32_t dtype, void* stream);

/* x *= (re + i im)   (im ignored for real dtypes) */
int qamd_scale(void* x, int64_t n, double re, double im, int32_t dtype, void* stream);
/* y = y * fy + x * fx  with real factors (slice accumulation). */
int qamd_axpby(void* y, const void* x, int64_t n, double fy, double fx, int32_t dtype, void* stream);
/*
 * Slice accumulation with device-resident exponents (no host round trip per slice):
 *   y * 10^(*y_exp) + x * 10^(*x_exp)  ->  y * 10^max(*y_exp, *x_exp);  *y_exp <- the max.
 * Exponents are doubles in device memory; -inf marks "nothing accumulated yet".
 */
int qamd_axpby_exp(void* y, const void* x, int64_t n, void* y_exp_dev, const void* x_exp_dev, int32_t dtype,
                   void* stream);
int qamd_conj(void* dst, const void* src, int64_t n, int32_t dtype, void* stream);
int qamd_cast(void* dst, int32_t dst_dtype, const void* src, int32_t src_dtype, int64_t n, void* stream);
int qamd_fill(void* dst, int64_t n, double re, double im, int32_t dtype, void* stream);
/*
 * Complex contraction support: dst (4n reals) <- for every complex src[i] the 2x2
 * real block [[re, im], [-im, re]] (im negated first if conj).  A complex pairwise
 * contraction is then ONE real qamd_contract_pair with the K and N bundles doubled:
 * A and C are used in place through their interleaved (re, im) real views.
 */
int qamd_complex_expand(void* dst, const void* src, int64_t n, int32_t conj, int32_t dtype, void* stream);

/*
 * Exponent stripping: x /= max|x|; *exponent_dev (double, device) +=
 * log10(max|x|).  scratch_dev: >= 8 bytes of device memory, zeroed by the
 * call itself.  If max|x| == 0 the data is left untouched and exponent
 * unchanged (cotengra's check_zero is handled by the host).
 */
int qamd_strip_exponent(void* x, int64_t n, int32_t dtype, void* scratch_dev,
                        void* exponent_dev, void* stream);
/* out_dev[0] (double, device) = max|x|; out_dev must provide 16 bytes (8 of scratch) */
int qamd_absmax(void* out_dev, const void* x, int64_t n, int32_t dtype, void* stream);

/*
 * dst[i] = f(src[i]);  op: 0 abs, 1 sqrt, 2 exp, 3 log, 4 log10.  Real dtypes; for C64 / C128 only
 * op 0, with dst REAL (float / double magnitudes).  Backs do("abs") / do("log10") ... of the
 * strip_exponent and norm fallbacks (quimb/tensor/array_ops.py:257-263, tensor_core.py:330-340).
 */
int qamd_unary(void* dst, const void* src, int64_t n, int32_t op, int32_t dtype, void* stream);
/* out_dev[0] (same real dtype as x) = max (want_min = 0) or min (want_min = 1) over x[0..n) */
int qamd_minmax(void* out_dev, const void* x, int64_t n, int32_t want_min, int32_t dtype, void* stream);

/*
 * A pairwise contraction whose result is consumed by ONE inner product over all its indices with a tensor T of the same
 * layout -- the closing pair of steps of a two-sided / four-quadrant contraction (cotengra's last two steps of
 * ctg.array_contract, quimb/tensor/contraction.py:285: the second join, then the `tensordot` over everything):
 *     out_dev[0] = sum_{b,m,n} (A . B)[b, m, n] * T[b, m, n]  /  (scale_a * scale_b * scale_t)
 * The result tensor never reaches memory (no 242 MB store and no 484 MB read-back on the 10x10 D=6 headline).  T is
 * addressed exactly as C would have been (plan->sc_*).  Only plans the planner put on the k-outer fp32 kernel
 * (qamd_pair_describe: gemmk_kernel<..>) or its split-product variant (gemmh_kernel<..>) are supported, anything else returns QAMD_EUNSUPPORTED and the caller issues
 * the two steps separately.  workspace: qamd_pair_dot_workspace_bytes(plan) bytes (one double per workgroup, summed in a
 * fixed order).  ep->scale_a / scale_b as in qamd_contract_pair_ex, scale_t = T's slots; ep->absmax_out[0] = |out|.
 */
int64_t qamd_pair_dot_workspace_bytes(const qamd_pair_plan* plan);
int qamd_contract_pair_dot(const qamd_pair_plan* plan, const void* A, const void* B, const void* T, void* out_dev,
                           void* workspace, int64_t workspace_bytes, const qamd_epilogue* ep, const void* scale_t,
                           void* stream);

/*
 * The vector work of one Lanczos step, on the device (the Krylov solver around TNLinearOperator.matvec in DMRG's local
 * solve: quimb/tensor/tn1d/dmrg.py:626-645 -> quimb/linalg/base_linalg.py:80, ARPACK on HOST vectors in the
 * reference).  Q: `rows` basis vectors of n elements, row stride ldq (elements); w: the matvec result, updated in place.
 *   project : h[i] = <Q_i, w> (conjugating Q), i < rows; with h_sum_dev != NULL also h_sum[i] = h[i] (accumulate == 0)
 *             or h_sum[i] += h[i] (a second Gram-Schmidt pass: subtract takes h, alpha is read from h_sum)
 *   subtract: w -= sum_i h[i] Q_i; want_norm != 0 leaves the partial sums of ||w||^2 in the workspace
 *   extend  : beta = ||w|| from those partial sums, alpha = Re h_j[0]; alpha_beta_dev[0..1] = (alpha, beta);
 *             q_next = w / beta, or ZERO when beta <= breakdown_eps * max(|alpha|, 1) (Krylov space exhausted: the
 *             host sees it in beta and the later steps stay finite)
 * No host read anywhere: a whole restart cycle is enqueued and (alpha, beta) are fetched once.  Reductions are
 * two-stage with a fixed order.  ws_dev: qamd_krylov_workspace_bytes(rows, n, dtype) bytes, shared by the three
 * calls of one step.  Not recordable in launch programs (-9 while recording).
 */
int64_t qamd_krylov_workspace_bytes(int32_t rows, int64_t n, int32_t dtype);
int qamd_krylov_project(void* h_dev, void* h_sum_dev, const void* Q, int64_t ldq, int32_t rows, const void* w, int64_t n,
                        int32_t accumulate, int32_t dtype, void* ws_dev, void* stream);
int qamd_krylov_subtract(void* w, const void* Q, int64_t ldq, int32_t rows, const void* h_dev, int64_t n,
                         int32_t want_norm, int32_t dtype, void* ws_dev, void* stream);
int qamd_krylov_extend(void* q_next, const void* w, int64_t n, const void* h_j_dev, double* alpha_beta_dev,
                       double breakdown_eps, int32_t dtype, const void* ws_dev, void* stream);

/*
 * A whole tree of SMALL contractions walked on the device (quimb's circuit amplitudes: hundreds of
 * pairwise steps on <= 2^10-element tensors, quimb/tensor/circuit/exact.py:417-501 -> the per-step
 * tensordot loop of ctg.array_contract, contraction.py:285).  steps_dev / etab_dev / ktab_dev: the lowered
 * plan (below) in device memory; inputs_dev: ninst x ninputs device pointers (instance i uses the i-th row; rows may share
 * tensors); arena_dev: ninst x arena_elems elements of scratch for the intermediates, or NULL to carve
 * arena_elems elements out of each workgroup's LDS (dependent steps then hand over through LDS); the step with
 * c_off < 0 writes the result to out_dev (ninst x out_elems).  One 256-thread workgroup per instance.
 * Strides and offsets are in ELEMENTS of the dtype (complex: one element = (re, im)).
 */
#define QAMD_MICRO_LDS_ARENA_BYTES (112 * 1024)   /* arena_dev == NULL: the arena lives in LDS, at most this big */
/* One pairwise step, fully lowered by the host: element e < total of the result is
 *   C[etab[3*(eoff+e)+2]] = sum_{k<K} A[etab[3*(eoff+e)] + ktab[2*(koff+k)]] * B[etab[3*(eoff+e)+1] + ktab[2*(koff+k)+1]]
 * (the mixed-radix tensor addressing of the GETT, tabulated: the tensors are tiny, the tables cheap, and the
 * device does no index arithmetic on the dependent path). */
typedef struct {
  int32_t a_kind, b_kind;      /* operand location: 0 = inputs[ref], 1 = arena + ref */
  int64_t a_ref, b_ref;
  int64_t c_off;               /* arena offset of the result, or -1: the output buffer */
  uint32_t total, K;           /* result elements (B*M*N) and contraction length */
  uint32_t eoff, koff;         /* first entry of this step in etab (triples) / ktab (pairs) */
} qamd_micro_step;
int qamd_microtree_run(int32_t dtype, const qamd_micro_step* steps_dev, int32_t nsteps, const int32_t* etab_dev,
                       const int32_t* ktab_dev, const void* const* inputs_dev, int32_t ninputs, void* arena_dev,
                       int64_t arena_elems, void* out_dev, int64_t out_elems, int32_t ninst, void* stream);
/* ... with flags.  QAMD_MICRO_WIDE (F32 / C64 trees; ignored for F64 / C128): the INTERMEDIATES are carried in double
 * precision -- every step accumulates in fp64, the arena's elements are 8 / 16 bytes (arena_dev, when given, must hold
 * ninst x arena_elems of THOSE; the LDS budget is checked against them), the inputs and the result keep the tree's dtype.
 * ~900 chained fp32 steps of a circuit amplitude (BASELINE config #2) lose 1-2e-6 to per-step rounding; this mode holds
 * the reference's 1e-6 (measured 3e-8) at 1.35 us per dependent step instead of 0.96.  flags = 0 is qamd_microtree_run. */
#define QAMD_MICRO_WIDE 1
int qamd_microtree_run_ex(int32_t dtype, const qamd_micro_step* steps_dev, int32_t nsteps, const int32_t* etab_dev,
                          const int32_t* ktab_dev, const void* const* inputs_dev, int32_t ninputs, void* arena_dev,
                          int64_t arena_elems, void* out_dev, int64_t out_elems, int32_t ninst, int32_t flags, void* stream);

/*
 * Launch programs: the launch sequence of ONE contraction recorded once and replayed with one host call -- the
 * replacement for the per-step Python loop of the reference's executor (ctg.array_contract under
 * quimb/tensor/contraction.py:285; quimb caches the planned expression and re-runs that loop on every call,
 * tests/test_tensor/test_contract.py:155-172).
 *
 *   P = qamd_program_create(nlanes);  qamd_program_record_begin(P);
 *   ... the ordinary qamd_* calls of one contraction, on this thread: each is APPENDED to P instead of launched (its
 *       stream argument is ignored); qamd_program_set_lane(P, l) routes what follows to lane l, qamd_program_wait(P, l, m)
 *       makes lane l wait for everything recorded on lane m so far; qamd_program_mark(P, tag) brackets the NEXT
 *       recorded launch with timing events ...
 *   qamd_program_record_end(P);
 *   qamd_program_bind_inputs(P, n, ptrs, nbytes);      (device pointers inside [ptrs[i], ptrs[i] + nbytes[i]) are
 *                                                        re-based onto input_ptrs[i] at every run)
 *   qamd_program_run(P, lane_streams, input_ptrs, timing);   (lane 0 = the caller's stream: forked from / joined to)
 *
 * Recorded are: qamd_contract_pair(_ex), qamd_contract_pair_dot, qamd_contract_chain2, qamd_contract_rowpass, qamd_permute, qamd_reduce_sum, qamd_binary,
 * qamd_scale, qamd_axpby(_exp), qamd_conj, qamd_cast, qamd_fill, qamd_complex_expand, qamd_strip_exponent,
 * qamd_absmax_log10_sum(_add), qamd_div_by_absmax, qamd_unary, qamd_minmax, qamd_absmax.  Plan compilation
 * (qamd_pair_build_ktab) and qamd_microtree_run execute immediately.  Every buffer a recorded call names -- other
 * than the bound inputs -- must stay allocated, at the same address, for as long as the program is run.  A program is
 * not thread-safe; recording is per thread.
 */
typedef struct qamd_program qamd_program;
qamd_program* qamd_program_create(int32_t nlanes);
void qamd_program_destroy(qamd_program* prog);
int qamd_program_record_begin(qamd_program* prog);
int qamd_program_set_lane(qamd_program* prog, int32_t lane);
int qamd_program_wait(qamd_program* prog, int32_t lane, int32_t on_lane);
int qamd_program_mark(qamd_program* prog, int32_t tag);
int qamd_program_record_end(qamd_program* prog);
int qamd_program_bind_inputs(qamd_program* prog, int32_t n, const void* const* ptrs, const int64_t* nbytes);
int32_t qamd_program_num_ops(const qamd_program* prog);        /* launches + waits */
int32_t qamd_program_num_launches(const qamd_program* prog);
int32_t qamd_program_num_marks(const qamd_program* prog);
/* lane_streams: nlanes hipStream_t; input_ptrs: as many device pointers as were bound (may be NULL if none);
 * timing = 0: no timing; timing = s + 1: the marked launches record their timing events into slot s (a slot per run of a
 * timed region keeps every reading; slots are created on first use, so a WARM-UP run per slot keeps event creation out
 * of the timed region) */
int qamd_program_run(qamd_program* prog, void* const* lane_streams, const void* const* input_ptrs, int32_t timing);
/* milliseconds the i-th marked launch took in the last run that used ``slot`` (synchronise first); *tag_out = its tag */
int qamd_program_mark_ms(qamd_program* prog, int32_t i, int32_t slot, int32_t* tag_out, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* QUIMB_AMD_H */
