// TEST INFRASTRUCTURE: host stand-in for contrastboundary_amd/csrc/knn_wave.h (same names).
#pragma once
#include <hip/hip_runtime.h>

static inline int kw_ffbl(unsigned v) { return v ? __builtin_ctz(v) : -1; }
static inline int kw_in_vgpr(int s) { return s; }
