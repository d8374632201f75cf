// One pairwise step of a device-resident contraction plan (microtree.hip).  Plain C: the same struct is
// declared in include/quimb_amd.h and mirrored with ctypes.
#ifndef QAMD_MICRO_ARGS_H
#define QAMD_MICRO_ARGS_H
#include "../../include/quimb_amd.h"
#endif
