// wx_debug.h -- NOT part of the C ABI (include/waveletsext_hip.h): dispatch override used by the parity suite to run the same
// inputs through more than one kernel family.  Process-global; nothing in the product path calls it.
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
/* 0 = normal dispatch; 1 = one level per launch instead of the fused kernels; 2 = keep the fused LDS kernels but skip the
 * register-resident ones (Haar Walsh-Hadamard, lattice) */
void wx_debug_set_dispatch(int mode);
#ifdef __cplusplus
}
#endif
