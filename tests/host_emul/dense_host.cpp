// TEST INFRASTRUCTURE (not product code): the product's BatchNorm passes (contrastboundary_amd/csrc/bn_rows.hip, with the residual tail of a block) and the criterion's
// cross entropy (cross_entropy.hip) compiled for the HOST with wave semantics (tests/host_emul/wave).  tests/test_dense_host.py calls the C entry points themselves.
#include "amdgcn.h"
#include "../../contrastboundary_amd/csrc/bn_rows.hip"
#define up256 xe_up256
#include "../../contrastboundary_amd/csrc/cross_entropy.hip"
