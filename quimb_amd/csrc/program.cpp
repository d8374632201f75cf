// program.cpp -- launch programs: record the C-ABI calls of ONE contraction once, replay them with one host call.
//
// The reference evaluates a contraction tree as a Python loop over pairwise steps (cotengra's executor under
// quimb/tensor/contraction.py:285); the first rounds of this backend kept that loop in Python and paid ~15 us of
// interpreter + ctypes time per launch -- 1.6 ms for the ~100 launches of one rank's share of the headline network,
// more than the device needs for its corner sweeps.  A program is the same launch sequence as plain data:
//
//   record   while qamd_program_record_begin() is in force on a thread, every stream-ordered entry point of this
//            library (qamd_contract_pair_ex, qamd_contract_chain2/3, qamd_permute, qamd_reduce_sum, qamd_binary, the
//            elementwise / exponent-stripping calls) appends its arguments to the program INSTEAD of launching;
//            qamd_program_set_lane / qamd_program_wait say which of the program's lanes (HIP streams at run time:
//            independent branches of the tree) a call belongs to and where one lane waits for another;
//   bind     qamd_program_bind_inputs names the address ranges of the network's input tensors: device pointers inside
//            them are re-based onto the caller's arrays at every run, everything else (the intermediates, owned by the
//            recorder for the program's lifetime) is replayed as recorded;
//   run      qamd_program_run walks the list in C: one fork event from lane 0, the recorded launches on their lanes'
//            streams with an event per cross-lane dependency, one join back onto lane 0.  No Python, no allocation,
//            no synchronisation.  Marked launches are bracketed by timing events (qamd_program_mark /
//            qamd_program_mark_ms): the bench's per-kernel durations come from inside the timed region.
//
// Not a hipGraph on purpose: the runtime maps the parallel branches of a captured graph onto fewer hardware queues
// than explicit streams get (measured in round 3), and a program re-bases its inputs without a re-capture.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/quimb_amd.h"
#include "program.h"

namespace {


constexpr int kMaxPtr = 14;

struct Op {
  int32_t kind = 0, lane = 0, nptr = 0;
  const void* ptr[kMaxPtr] = {};
  int32_t in_idx[kMaxPtr];
  int64_t in_off[kMaxPtr] = {};
  int64_t iv[8] = {};
  double dv[2] = {};
  std::vector<int64_t> arr;   // shapes / strides
  std::vector<char> blob;     // a plan struct, by value
  int32_t on_lane = -1;       // QP_WAIT
  hipEvent_t ev = nullptr;    // QP_WAIT
  int32_t mark = -1;          // index into Program::marks, or -1
  Op() { for (int i = 0; i < kMaxPtr; ++i) in_idx[i] = -1; }
};

// a marked launch: one (start, end) timing-event pair per SLOT, so that the K runs of a timed region keep K readings
struct Mark { int32_t tag; std::vector<hipEvent_t> e0, e1; };

}  // namespace

struct qamd_program {
  int32_t nlanes = 1;
  std::vector<Op> ops;
  std::vector<Mark> marks;
  std::vector<hipEvent_t> join_ev;   // [nlanes]: lane -> lane 0 at the end of a run
  hipEvent_t fork_ev = nullptr;
  int32_t pending_tag = -1;          // set by qamd_program_mark: the next recorded launch is timed
  bool recording = false, bound = false, events_ready = false;
  int32_t cur_lane = 0;
  std::vector<bool> lane_used;
};

static thread_local qamd_program* g_rec = nullptr;

bool qamdp_recording() { return g_rec != nullptr; }

static Op& push(QamdpKind k, std::initializer_list<const void*> ptrs) {
  qamd_program* P = g_rec;
  P->ops.emplace_back();
  Op& o = P->ops.back();
  o.kind = k;
  o.lane = P->cur_lane;
  P->lane_used[o.lane] = true;
  for (const void* p : ptrs) o.ptr[o.nptr++] = p;
  if (P->pending_tag >= 0) {
    Mark m{P->pending_tag, {}, {}};   // (events are created by the runs that time it: recording needs no device)
    o.mark = (int32_t)P->marks.size();
    P->marks.push_back(m);
    P->pending_tag = -1;
  }
  return o;
}

template <class T> static void put_blob(Op& o, const T* p) {
  o.blob.resize(sizeof(T));
  memcpy(o.blob.data(), p, sizeof(T));
}
static void put_arr(Op& o, const int64_t* a, int n) { o.arr.insert(o.arr.end(), a, a + n); }

// ---- recorders (called from the entry points while a recording is in force) ----------------------------------
int qamdp_rec_pair(const qamd_pair_plan* p, const void* A, const void* B, void* C, const void* ktab, void* ws,
                   int64_t ws_bytes, const qamd_epilogue* ep) {
  if (!p) return QAMD_EINVAL;
  Op& o = push(QP_PAIR, {A, B, C, ktab, ws, ep ? ep->scale_a : nullptr, ep ? ep->scale_b : nullptr,
                        ep ? ep->absmax_out : nullptr});
  put_blob(o, p);
  o.iv[0] = ws_bytes;
  o.iv[1] = ep ? 1 : 0;
  return QAMD_OK;
}
int qamdp_rec_pair_dot(const qamd_pair_plan* p, const void* A, const void* B, const void* T, void* out, void* ws,
                       int64_t ws_bytes, const qamd_epilogue* ep, const void* scale_t) {
  if (!p) return QAMD_EINVAL;
  Op& o = push(QP_PAIRDOT, {A, B, T, out, ws, ep ? ep->scale_a : nullptr, ep ? ep->scale_b : nullptr,
                           ep ? ep->absmax_out : nullptr, scale_t});
  put_blob(o, p);
  o.iv[0] = ws_bytes;
  o.iv[1] = ep ? 1 : 0;
  return QAMD_OK;
}
int qamdp_rec_chain2(const qamd_chain2_plan* p, const void* A, const void* W1, const void* W2, void* C, const void* k1,
                     const void* co, const void* sa, const void* s1, const void* s2, void* amax) {
  if (!p) return QAMD_EINVAL;
  Op& o = push(QP_CHAIN2, {A, W1, W2, C, k1, co, sa, s1, s2, amax});
  put_blob(o, p);
  return QAMD_OK;
}
int qamdp_rec_rowpass(const qamd_rowpass_plan* p, const void* A, const void* const* W, void* C, const void* sa,
                      const void* const* sw, void* amax) {
  if (!p || !W) return QAMD_EINVAL;
  // pointer slots: A, C, W[0..4], scale_a, scale_w[0..4], absmax_out
  Op& o = push(QP_ROWPASS, {A, C, W[0], W[1], W[2], W[3], W[4], sa, sw ? sw[0] : nullptr, sw ? sw[1] : nullptr,
                            sw ? sw[2] : nullptr, sw ? sw[3] : nullptr, sw ? sw[4] : nullptr, amax});
  put_blob(o, p);
  return QAMD_OK;
}
int qamdp_rec_permute(void* dst, const void* src, int32_t ndim, const int64_t* shape, const int64_t* strides,
                      int64_t offset, int32_t dtype) {
  if (ndim < 0 || ndim > QAMD_MAX_NDIM) return QAMD_EINVAL;
  Op& o = push(QP_PERMUTE, {dst, src});
  o.iv[0] = ndim; o.iv[1] = offset; o.iv[2] = dtype;
  put_arr(o, shape, ndim);
  put_arr(o, strides, ndim);
  return QAMD_OK;
}
int qamdp_rec_reduce(void* out, const void* x, int32_t ndk, const int64_t* shk, const int64_t* stk, int32_t ndr,
                     const int64_t* shr, const int64_t* str, int32_t dtype) {
  if (ndk < 0 || ndr < 0 || ndk > QAMD_MAX_NDIM || ndr > QAMD_MAX_NDIM) return QAMD_EINVAL;
  Op& o = push(QP_REDUCE, {out, x});
  o.iv[0] = ndk; o.iv[1] = ndr; o.iv[2] = dtype;
  put_arr(o, shk, ndk); put_arr(o, stk, ndk); put_arr(o, shr, ndr); put_arr(o, str, ndr);
  return QAMD_OK;
}
int qamdp_rec_binary(void* out, const void* a, const int64_t* as, const void* b, const int64_t* bs, int32_t ndim,
                     const int64_t* shape, int32_t op, int32_t dtype) {
  if (ndim < 0 || ndim > QAMD_MAX_NDIM) return QAMD_EINVAL;
  Op& o = push(QP_BINARY, {out, a, b});
  o.iv[0] = ndim; o.iv[1] = op; o.iv[2] = dtype;
  put_arr(o, as, ndim); put_arr(o, bs, ndim); put_arr(o, shape, ndim);
  return QAMD_OK;
}
int qamdp_rec_simple(int32_t kind, const void* p0, const void* p1, const void* p2, const void* p3, int64_t i0,
                     int64_t i1, int64_t i2, double d0, double d1) {
  Op& o = push((QamdpKind)kind, {p0, p1, p2, p3});
  o.iv[0] = i0; o.iv[1] = i1; o.iv[2] = i2;
  o.dv[0] = d0; o.dv[1] = d1;
  return QAMD_OK;
}

// ---- the public face ---------------------------------------------------------------------------------------------
extern "C" qamd_program* qamd_program_create(int32_t nlanes) {
  if (nlanes < 1 || nlanes > 64) return nullptr;
  qamd_program* P = new qamd_program();
  P->nlanes = nlanes;
  P->lane_used.assign(nlanes, false);
  P->join_ev.assign(nlanes, nullptr);
  return P;
}

extern "C" void qamd_program_destroy(qamd_program* P) {
  if (!P) return;
  if (g_rec == P) g_rec = nullptr;
  for (Op& o : P->ops)
    if (o.ev) (void)hipEventDestroy(o.ev);
  for (Mark& m : P->marks) {
    for (hipEvent_t e : m.e0) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : m.e1) if (e) (void)hipEventDestroy(e);
  }
  for (hipEvent_t e : P->join_ev)
    if (e) (void)hipEventDestroy(e);
  if (P->fork_ev) (void)hipEventDestroy(P->fork_ev);
  delete P;
}

extern "C" int qamd_program_record_begin(qamd_program* P) {
  if (!P || g_rec || P->recording || !P->ops.empty()) return QAMD_EINVAL;
  P->recording = true;
  P->cur_lane = 0;
  g_rec = P;
  return QAMD_OK;
}

extern "C" int qamd_program_set_lane(qamd_program* P, int32_t lane) {
  if (!P || g_rec != P || lane < 0 || lane >= P->nlanes) return QAMD_EINVAL;
  P->cur_lane = lane;
  return QAMD_OK;
}

extern "C" int qamd_program_wait(qamd_program* P, int32_t lane, int32_t on_lane) {
  if (!P || g_rec != P || lane < 0 || lane >= P->nlanes || on_lane < 0 || on_lane >= P->nlanes) return QAMD_EINVAL;
  if (lane == on_lane) return QAMD_OK;
  P->ops.emplace_back();
  Op& o = P->ops.back();
  o.kind = QP_WAIT;
  o.lane = lane;
  o.on_lane = on_lane;
  P->lane_used[lane] = P->lane_used[on_lane] = true;
  return QAMD_OK;
}

extern "C" int qamd_program_mark(qamd_program* P, int32_t tag) {
  if (!P || g_rec != P || tag < 0) return QAMD_EINVAL;
  P->pending_tag = tag;
  return QAMD_OK;
}

extern "C" int qamd_program_record_end(qamd_program* P) {
  if (!P || g_rec != P) return QAMD_EINVAL;
  g_rec = nullptr;
  P->recording = false;
  P->pending_tag = -1;
  return QAMD_OK;
}

extern "C" int qamd_program_bind_inputs(qamd_program* P, int32_t n, const void* const* ptrs, const int64_t* nbytes) {
  if (!P || P->recording || n < 0 || (n && (!ptrs || !nbytes))) return QAMD_EINVAL;
  // the ranges must be disjoint: a recorded pointer is re-based onto THE input that contained it, and with overlapping
  // inputs (the same array passed twice, overlapping views) that is ambiguous -- a replay on distinct arrays would read
  // the wrong one.  The caller records on private copies instead (quimb_amd/program.py).
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      const char *a = (const char*)ptrs[i], *b = (const char*)ptrs[j];
      if (a && b && nbytes[i] > 0 && nbytes[j] > 0 && a < b + nbytes[j] && b < a + nbytes[i]) return QAMD_EINVAL;
    }
  for (Op& o : P->ops)
    for (int i = 0; i < o.nptr; ++i) {
      o.in_idx[i] = -1;
      const char* p = (const char*)o.ptr[i];
      if (!p) continue;
      for (int j = 0; j < n; ++j) {
        const char* b = (const char*)ptrs[j];
        if (b && p >= b && p < b + nbytes[j]) {
          o.in_idx[i] = j;
          o.in_off[i] = p - b;
          break;
        }
      }
    }
  P->bound = true;
  return QAMD_OK;
}

extern "C" int32_t qamd_program_num_ops(const qamd_program* P) { return P ? (int32_t)P->ops.size() : -1; }
extern "C" int32_t qamd_program_num_launches(const qamd_program* P) {
  if (!P) return -1;
  int32_t n = 0;
  for (const Op& o : P->ops) n += o.kind != QP_WAIT;
  return n;
}
extern "C" int32_t qamd_program_num_marks(const qamd_program* P) { return P ? (int32_t)P->marks.size() : -1; }

// elapsed time of marked launch i in the last run that used timing slot ``slot`` (the caller synchronises first);
// tag_out may be NULL
extern "C" int qamd_program_mark_ms(qamd_program* P, int32_t i, int32_t slot, int32_t* tag_out, float* ms_out) {
  if (!P || i < 0 || i >= (int32_t)P->marks.size() || !ms_out || slot < 0) return QAMD_EINVAL;
  Mark& m = P->marks[i];
  if (tag_out) *tag_out = m.tag;
  if (slot >= (int32_t)m.e0.size() || !m.e0[slot]) return QAMD_EINVAL;
  return hipEventElapsedTime(ms_out, m.e0[slot], m.e1[slot]) == hipSuccess ? QAMD_OK : QAMD_ELAUNCH;
}

static int run_op(const Op& o, const void* const* q, void* st) {
  auto P0 = [&](int i) { return const_cast<void*>(q[i]); };
  switch (o.kind) {
    case QP_PAIR: {
      qamd_epilogue ep{q[5], q[6], P0(7)};
      return qamd_contract_pair_ex((const qamd_pair_plan*)o.blob.data(), q[0], q[1], P0(2), q[3], P0(4), o.iv[0],
                                   o.iv[1] ? &ep : nullptr, st);
    }
    case QP_PAIRDOT: {
      qamd_epilogue ep{q[5], q[6], P0(7)};
      return qamd_contract_pair_dot((const qamd_pair_plan*)o.blob.data(), q[0], q[1], q[2], P0(3), P0(4), o.iv[0],
                                    o.iv[1] ? &ep : nullptr, q[8], st);
    }
    case QP_CHAIN2:
      return qamd_contract_chain2((const qamd_chain2_plan*)o.blob.data(), q[0], q[1], q[2], P0(3), q[4], q[5], q[6],
                                  q[7], q[8], P0(9), st);
    case QP_ROWPASS: {
      const void* W[5] = {q[2], q[3], q[4], q[5], q[6]};
      const void* sw[5] = {q[8], q[9], q[10], q[11], q[12]};
      return qamd_contract_rowpass((const qamd_rowpass_plan*)o.blob.data(), q[0], W, P0(1), q[7], sw, P0(13), st);
    }
    case QP_PERMUTE: {
      const int nd = (int)o.iv[0];
      return qamd_permute(P0(0), q[1], nd, o.arr.data(), o.arr.data() + nd, o.iv[1], (int32_t)o.iv[2], st);
    }
    case QP_REDUCE: {
      const int nk = (int)o.iv[0], nr = (int)o.iv[1];
      const int64_t* a = o.arr.data();
      return qamd_reduce_sum(P0(0), q[1], nk, a, a + nk, nr, a + 2 * nk, a + 2 * nk + nr, (int32_t)o.iv[2], st);
    }
    case QP_BINARY: {
      const int nd = (int)o.iv[0];
      const int64_t* a = o.arr.data();
      return qamd_binary(P0(0), q[1], a, q[2], a + nd, nd, a + 2 * nd, (int32_t)o.iv[1], (int32_t)o.iv[2], st);
    }
    case QP_SCALE: return qamd_scale(P0(0), o.iv[0], o.dv[0], o.dv[1], (int32_t)o.iv[1], st);
    case QP_AXPBY: return qamd_axpby(P0(0), q[1], o.iv[0], o.dv[0], o.dv[1], (int32_t)o.iv[1], st);
    case QP_AXPBY_EXP: return qamd_axpby_exp(P0(0), q[1], o.iv[0], P0(2), q[3], (int32_t)o.iv[1], st);
    case QP_CONJ: return qamd_conj(P0(0), q[1], o.iv[0], (int32_t)o.iv[1], st);
    case QP_CAST: return qamd_cast(P0(0), (int32_t)o.iv[1], q[1], (int32_t)o.iv[2], o.iv[0], st);
    case QP_FILL: return qamd_fill(P0(0), o.iv[0], o.dv[0], o.dv[1], (int32_t)o.iv[1], st);
    case QP_CEXPAND: return qamd_complex_expand(P0(0), q[1], o.iv[0], (int32_t)o.iv[2], (int32_t)o.iv[1], st);
    case QP_STRIP: return qamd_strip_exponent(P0(0), o.iv[0], (int32_t)o.iv[1], P0(1), P0(2), st);
    case QP_LOG10SUM: return qamd_absmax_log10_sum(q[0], o.iv[0], (int32_t)o.iv[1], P0(1), st);
    case QP_LOG10SUM_ADD: return qamd_absmax_log10_sum_add(q[0], o.iv[0], (int32_t)o.iv[1], P0(1), st);
    case QP_DIVABS: return qamd_div_by_absmax(P0(0), o.iv[0], q[1], (int32_t)o.iv[1], st);
    case QP_UNARY: return qamd_unary(P0(0), q[1], o.iv[0], (int32_t)o.iv[2], (int32_t)o.iv[1], st);
    case QP_MINMAX: return qamd_minmax(P0(0), q[1], o.iv[0], (int32_t)o.iv[2], (int32_t)o.iv[1], st);
    case QP_ABSMAX: return qamd_absmax(P0(0), q[1], o.iv[0], (int32_t)o.iv[1], st);
    default: return QAMD_EINVAL;
  }
}

extern "C" int qamd_program_run(qamd_program* P, void* const* lane_streams, const void* const* input_ptrs,
                                int32_t timing) {
  if (!P || P->recording || g_rec || !lane_streams) return QAMD_EINVAL;
  auto S = [&](int lane) { return (hipStream_t)lane_streams[lane]; };
  if (!P->events_ready) {      // first run: the events of the fork, the joins, the cross-lane waits and the marks
    if (hipEventCreateWithFlags(&P->fork_ev, hipEventDisableTiming) != hipSuccess) return QAMD_ELAUNCH;
    for (int l = 1; l < P->nlanes; ++l)
      if (P->lane_used[l] && hipEventCreateWithFlags(&P->join_ev[l], hipEventDisableTiming) != hipSuccess)
        return QAMD_ELAUNCH;
    for (Op& o : P->ops)
      if (o.kind == QP_WAIT && hipEventCreateWithFlags(&o.ev, hipEventDisableTiming) != hipSuccess) return QAMD_ELAUNCH;
    P->events_ready = true;
  }
  // fork: every side lane starts behind whatever precedes the run on lane 0 (the inputs are ready there)
  bool side = false;
  for (int l = 1; l < P->nlanes; ++l) side = side || P->lane_used[l];
  if (side) {
    if (hipEventRecord(P->fork_ev, S(0)) != hipSuccess) return QAMD_ELAUNCH;
    for (int l = 1; l < P->nlanes; ++l)
      if (P->lane_used[l] && hipStreamWaitEvent(S(l), P->fork_ev, 0) != hipSuccess) return QAMD_ELAUNCH;
  }
  const void* q[kMaxPtr];
  for (Op& o : P->ops) {
    if (o.kind == QP_WAIT) {
      if (hipEventRecord(o.ev, S(o.on_lane)) != hipSuccess) return QAMD_ELAUNCH;
      if (hipStreamWaitEvent(S(o.lane), o.ev, 0) != hipSuccess) return QAMD_ELAUNCH;
      continue;
    }
    for (int i = 0; i < o.nptr; ++i) {
      if (o.in_idx[i] >= 0) {
        if (!input_ptrs || !input_ptrs[o.in_idx[i]]) return QAMD_EINVAL;
        q[i] = (const char*)input_ptrs[o.in_idx[i]] + o.in_off[i];
      } else {
        q[i] = o.ptr[i];
      }
    }
    const bool timed = timing > 0 && o.mark >= 0;
    if (timed) {
      Mark& m = P->marks[o.mark];
      const size_t slot = (size_t)timing - 1;
      if (m.e0.size() <= slot) { m.e0.resize(slot + 1, nullptr); m.e1.resize(slot + 1, nullptr); }
      if (!m.e0[slot] && (hipEventCreate(&m.e0[slot]) != hipSuccess || hipEventCreate(&m.e1[slot]) != hipSuccess))
        return QAMD_ELAUNCH;
      (void)hipEventRecord(m.e0[slot], S(o.lane));
    }
    const int rc = run_op(o, q, (void*)S(o.lane));
    if (rc) return rc;
    if (timed) (void)hipEventRecord(P->marks[o.mark].e1[(size_t)timing - 1], S(o.lane));
  }
  // join: lane 0 ends behind every side lane, so whoever waits on lane 0 waits for the whole program
  for (int l = 1; l < P->nlanes; ++l)
    if (P->lane_used[l]) {
      if (hipEventRecord(P->join_ev[l], S(l)) != hipSuccess) return QAMD_ELAUNCH;
      if (hipStreamWaitEvent(S(0), P->join_ev[l], 0) != hipSuccess) return QAMD_ELAUNCH;
    }
  return QAMD_OK;
}
