// Kernel-argument blocks shared by elementwise.hip (device) and api.cpp (host).
#pragma once
#include <stdint.h>

#define QAMD_PG 24  // groups per bundle (after fusion)

struct PermArgs {
  int32_t nx, ny, nz;
  int32_t direct;  // 1: X is fast in both src and dst, no LDS transpose
  int32_t TX, TY;
  uint32_t X, Y, Z;
  uint32_t tiles_x, tiles_y;
  uint32_t dim_x[QAMD_PG], dim_y[QAMD_PG], dim_z[QAMD_PG];
  int64_t ss_x[QAMD_PG], sd_x[QAMD_PG];
  int64_t ss_y[QAMD_PG], sd_y[QAMD_PG];
  int64_t ss_z[QAMD_PG], sd_z[QAMD_PG];
  int64_t src_offset;
  // permute_stream_kernel: bundle X = the tile's groups in source order, Z = the loop; see elementwise.hip
  int32_t xorder[QAMD_PG];   // X groups sorted by destination stride, fastest first
  uint32_t zchunk;     // z values per workgroup
};

struct ReduceArgs {
  int32_t nd_keep, nd_red;
  int32_t wave_per_out;
  uint32_t n_keep, n_red;
  uint32_t dim_keep[QAMD_PG], dim_red[QAMD_PG];
  int64_t s_keep[QAMD_PG], s_red[QAMD_PG];
};

struct BinaryArgs {
  int32_t nd, op;
  int64_t n;
  int64_t dim[QAMD_PG], sa[QAMD_PG], sb[QAMD_PG];
};

#ifdef __cplusplus
extern "C" {
#endif
int qamd_permute_launch(int esize, void* dst, const void* src, const PermArgs* p, void* stream);
int qamd_permute_stream_launch(int esize, void* dst, const void* src, const PermArgs* p, void* stream);
int qamd_reduce_sum_launch(int dtype, void* out, const void* x, const ReduceArgs* p, void* stream);
int qamd_binary_launch(int dtype, void* out, const void* a, const void* b, const BinaryArgs* p, void* stream);
#ifdef __cplusplus
}
#endif
