/* marlin_hip_testhooks.h -- what libmarlin_hip_testhooks.so exports ON TOP of include/marlin_hip.h: fault injection and the device
 * self-test of the 30-bit field arithmetic.  Test infrastructure: libmarlin_hip.so exports none of it (nm -D shows no debug /
 * selftest symbol); the hooks library is the same objects plus marlin_amd/csrc/testhooks.hip, and a test that needs a hook
 * loads it INSTEAD of the product library (MARLIN_AMD_LIB=<path>).  Nothing here replaces anything in the reference. */
#ifndef MARLIN_HIP_TESTHOOKS_H
#define MARLIN_HIP_TESTHOOKS_H
#include "marlin_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)
/* Device self-test of the 30-bit-limb base-field arithmetic used by the fixed-base MSM path against the 32-bit
 * Montgomery arithmetic: n pseudo-random operand pairs (field operations, XYZZ doubling / addition incl. the equal-x
 * path); *mismatches_out = number of operand pairs with any disagreement (0 expected). */
int mh_selftest_fq30(uint64_t n, uint64_t seed, uint64_t* mismatches_out);
/* Test hook: the nth (>= 1) request for device scratch memory from now on fails with MH_ENOMEM, as if the device had run out of
 * memory, whether or not that request would have had to allocate (0 disarms the hook); *calls_out (may be NULL) = the number of
 * such requests made so far.  How the tests make ONE rank of a sharded proof fail mid-prove: a rank that fails locally keeps
 * entering the collectives of the proof (with a meaningless payload) up to the next all-gather of partial points, whose error word
 * makes EVERY rank return non-zero from the same commit round -- the job fails, nobody hangs, the next proof can run.  Replaces
 * nothing in the reference (it has no FFI and no device memory). */
int mh_debug_fail_scratch(int nth, uint64_t* calls_out);
/* Test hook: on != 0 fills every device allocation the library makes from now on (scratch buffers, prover-key buffers) with 0xA5
 * bytes, so that a kernel reading memory nothing has written yet gets garbage for sure instead of whatever the heap held (usually
 * zeros -- the identity, the zero polynomial -- on a fresh process): tests/test_gpu_poisoned_allocations.py. */
int mh_debug_poison_scratch(int on);
int mh_debug_corrupt(int stage);        /* the next fixed-base MSM batch damages its own data after that stage, once: 1 = a sorted entry,
                                           2 = a bucket becomes garbage, 3 = a bucket becomes its neighbour, 4 = a plane becomes another */

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* MARLIN_HIP_TESTHOOKS_H */
