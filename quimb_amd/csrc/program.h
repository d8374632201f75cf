// program.h -- hooks between the C-ABI entry points and the launch-program recorder (program.cpp).
#pragma once
#include <stdint.h>

#include "../../include/quimb_amd.h"

// kinds of the "simple" (pointers + a few scalars) recorded calls; the structured ones have recorders of their own
enum QamdpKind : int32_t {
  QP_PAIR, QP_CHAIN2, QP_PERMUTE, QP_REDUCE, QP_BINARY, QP_SCALE, QP_AXPBY, QP_AXPBY_EXP, QP_CONJ, QP_CAST,
  QP_FILL, QP_CEXPAND, QP_STRIP, QP_LOG10SUM, QP_LOG10SUM_ADD, QP_DIVABS, QP_UNARY, QP_MINMAX, QP_ABSMAX, QP_PAIRDOT, QP_ROWPASS, QP_WAIT
};

// true while THIS thread records a program: the entry points then append their arguments instead of launching
bool qamdp_recording();
int qamdp_rec_pair(const qamd_pair_plan* p, const void* A, const void* B, void* C, const void* ktab, void* ws,
                   int64_t ws_bytes, const qamd_epilogue* ep);
int qamdp_rec_pair_dot(const qamd_pair_plan* p, const void* A, const void* B, const void* T, void* out, void* ws,
                       int64_t ws_bytes, const qamd_epilogue* ep, const void* scale_t);
int qamdp_rec_chain2(const qamd_chain2_plan* p, const void* A, const void* W1, const void* W2, void* C, const void* k1,
                     const void* co, const void* sa, const void* s1, const void* s2, void* amax);
int qamdp_rec_rowpass(const qamd_rowpass_plan* p, const void* A, const void* const* W, void* C, const void* sa,
                      const void* const* sw, void* amax);
int qamdp_rec_permute(void* dst, const void* src, int32_t ndim, const int64_t* shape, const int64_t* strides,
                      int64_t offset, int32_t dtype);
int qamdp_rec_reduce(void* out, const void* x, int32_t ndk, const int64_t* shk, const int64_t* stk, int32_t ndr,
                     const int64_t* shr, const int64_t* str, int32_t dtype);
int qamdp_rec_binary(void* out, const void* a, const int64_t* as, const void* b, const int64_t* bs, int32_t ndim,
                     const int64_t* shape, int32_t op, int32_t dtype);
// (kind, up to four device pointers, three integers, two doubles): the meaning of each slot is fixed per kind in
// program.cpp's run_op
int qamdp_rec_simple(int32_t kind, const void* p0, const void* p1, const void* p2, const void* p3, int64_t i0,
                     int64_t i1, int64_t i2, double d0, double d1);
