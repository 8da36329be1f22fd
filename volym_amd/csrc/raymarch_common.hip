// The common instantiation of the persistent-workgroup march (raymarch_pq.h: table mode, opacity on, no importance mode; linear layout) in a translation unit of its own, because it is compiled with another instruction scheduler than the rest:
// -mllvm -amdgpu-sched-strategy=iterative-ilp -O2 (Makefile).  Measured on one MI355X box with both builds side by side
// (scripts/lib_ab_rows.sh): the headline frame 32.6 -> 32.0 us with the scheduler, -> 31.8 with -O2 on top; every other instantiation is 0.5-4 per cent SLOWER with it
// (3840x2160 +0.5, importance +1, smoothing +2, trilinear +4; the bricked twin of this instantiation at 1024^3 @ 4K +0.5: it stays
// in raymarch.hip), so the flag is not global.  The price here: 24 bytes of
// scratch per lane in the ray set-up (profiles/r03_kernel_resources.txt).  Scheduling cannot change a pixel: -ffp-contract=off
// holds in both units.  raymarch.hip declares it `extern template` and launches it.
#include <hip/hip_runtime.h>

#include "raymarch_pq.h"

namespace volym {

template __global__ void volym_raymarch_pq_kernel<true, false, false, 4, false, false, false, PQ_WAVES, 0, false>(
    const uint8_t* __restrict__, const uint8_t* __restrict__, const FrameTables* __restrict__, const uint8_t* __restrict__, const uint2* __restrict__, uint32_t,
    uint16_t* __restrict__, uint32_t* __restrict__, uint32_t* __restrict__, float4* __restrict__, Counters* __restrict__, uint4* __restrict__, const FrameParams);

}  // namespace volym
