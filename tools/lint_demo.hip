// Not part of libskm.so.  tests/test_isa_lint.py compiles this file as it is and with one hand-placed wait removed
// (-DSKM_LINT_DEMO_...), and checks that tools/isa_lint.py passes the first object and fails each of the others:
//   SKM_LINT_DEMO_ODD_NO_SETTLE   rule i: the run-time logits ending (odd pair count, then the branch on the tail group).
//                                 In k_estep_solo<1, 1, -1> this is round 2's wrong-logits bug (DESIGN.md section 4.1:
//                                 register copies at a merge behind asm MFMAs read the accumulators too early); the
//                                 shipped k_estep_solo has a compile-time ending since, the team kernels kept the run-time
//                                 one and the lint found the same copies there.
//   SKM_LINT_DEMO_TAIL_NO_PAD     rule ii: a compiler copy of the logits accumulator one wait state ahead of the asm
//                                 MFMA that reads it as its C operand (the class of round 2's accumulator-clear bug)
//   SKM_LINT_DEMO_NO_DMA_WAIT     rule iii: no s_waitcnt vmcnt between the LDS-direct frame loads and the next tile's reads
#include "../speaker-recognition_amd/csrc/skm_gmm_kernels.h"

template __global__ void k_estep_solo<1, 1, -1>(FusedArgs);
template __global__ void k_estep_solo<2, 1, -1>(FusedArgs);
template __global__ void k_seg_team<1, 1, 1, true, -1, true, 2>(TeamArgs);
template __global__ void k_seg_team<2, 3, 1, true, -1, false, 4>(TeamArgs);
