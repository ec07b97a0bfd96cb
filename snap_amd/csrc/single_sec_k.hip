// single_sec_k.hip -- the single-end kernel with secondary results (-om), one affine-gap variant per translation unit.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DSINGLE_AGC=<3|4|6|0> -c single_sec_k.hip
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
#define SE_CAT2(a, b) a##b
#define SE_CAT(a, b) SE_CAT2(a, b)

extern "C" void SE_CAT(snapgpu_launch_single_sec_, SINGLE_AGC)(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL((k_align_single<SINGLE_AGC, true>), dim3(blocks), dim3(256), lds_bytes, s, *a);
}

#if 1
// exact replay of flagged reads (kernel_common.h: AlignArgs::flag_list): every variant has its exact twin since round 6 (AGC 4 / 6 are AGC 3's code plus the LDS form beyond 192 positions)
extern "C" void SE_CAT(snapgpu_launch_single_exact_, SINGLE_AGC)(const AlignArgs *a, int sec, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    if (sec) hipLaunchKernelGGL((k_align_single<SINGLE_AGC, true, true>), dim3(blocks), dim3(256), lds_bytes, s, *a);
    else     hipLaunchKernelGGL((k_align_single<SINGLE_AGC, false, true>), dim3(blocks), dim3(256), lds_bytes, s, *a);
}
#endif
