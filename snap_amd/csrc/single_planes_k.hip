// single_planes_k.hip -- the single-end kernels WITH the plane Landau-Vishkin compiled in (align_single.h: Aligner<.., PLANES>; planes.h),
// one affine-gap variant per translation unit.  Launched instead of the default instantiations when the context was created under
// SNAPGPU_LV_PLANES=1 (an option that measured slower): kept out of the default kernels, whose register allocation
// paid for it (exact form 520 -> 408 bytes of scratch per lane, fast form 712 -> 592).  Plain launches only: the secondary-result and
// timed instantiations, and the paired-end kernels' single-end fallback, run without planes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DSINGLE_AGC=<3|4|6|0> -c single_planes_k.hip
#include <hip/hip_runtime.h>
#include "../../include/snapgpu.h"
#include "dev_common.h"
#include "probe.h"
#include "lv.h"
#include "ag_win.h"
#include "align_single.h"
#include "kernel_common.h"
#include "single_kernel.h"

#ifndef SINGLE_AGC
#error "SINGLE_AGC must be defined (3, 4, 6 or 0)"
#endif
#define SP_CAT2(a, b) a##b
#define SP_CAT(a, b) SP_CAT2(a, b)

extern "C" void SP_CAT(snapgpu_launch_single_planes_, SINGLE_AGC)(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL((k_align_single<SINGLE_AGC, false, false, false, true>), dim3(blocks), dim3(256), lds_bytes, s, *a);
}

#if SINGLE_AGC == 0 || SINGLE_AGC == 3
extern "C" void SP_CAT(snapgpu_launch_single_exact_planes_, SINGLE_AGC)(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL((k_align_single<SINGLE_AGC, false, true, false, true>), dim3(blocks), dim3(256), lds_bytes, s, *a);
}
#endif
