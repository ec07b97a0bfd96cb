// paired_k.hip -- one affine-gap variant of the paired-end kernel per translation unit.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DPAIRED_AGC=<3|4|6|0> [-DPAIRED_SEC] -c paired_k.hip
// (k_align_paired inlines the whole paired + single-end control flow; compiling the four variants in parallel keeps
// the build at minutes instead of tens of minutes).
#include <hip/hip_runtime.h>
// The affine-gap and Landau-Vishkin code as one function each instead of an inlined copy per call site (ag_win.h: ag_dispatch_fn): this
// kernel runs 2 waves per SIMD, so the functions have all the registers they want.  The single-end kernels (6 waves per SIMD, two call
// sites) keep the inlined form: a callee cannot be given the kernel's register budget from HIP source.
#define SNAPGPU_AG_LV_FUNCTIONS 1
#include "paired_dev.h"

#ifndef PAIRED_AGC
#error "PAIRED_AGC must be defined (3, 4, 6 or 0)"
#endif
#define PE_CAT2(a, b) a##b
#define PE_CAT(a, b) PE_CAT2(a, b)

#ifdef PAIRED_SEC       // the variant that also produces secondary results (-om)
extern "C" void PE_CAT(snapgpu_launch_paired_sec_, PAIRED_AGC)(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL((k_align_paired<PAIRED_AGC, true>), dim3(blocks), dim3(256), lds_bytes, s, *a);
}
#else
extern "C" void PE_CAT(snapgpu_launch_paired_, PAIRED_AGC)(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL((k_align_paired<PAIRED_AGC, false>), dim3(blocks), dim3(256), lds_bytes, s, *a);
}
#endif

#if PAIRED_AGC == 3 && !defined(PAIRED_SEC)
extern "C" void snapgpu_launch_collect_flagged(snapgpu_paired_result *primary, uint32_t n, uint32_t *list, uint32_t *count, int stale, hipStream_t s)
{
    hipLaunchKernelGGL(k_collect_flagged<0>, dim3((n + 255) / 256), dim3(256), 0, s, primary, n, list, count, stale);
}
#endif

// exact replay of flagged pairs (paired_args.h: PairedArgs::persist): every variant has its exact twin since round 6 -- AGC 4 / 6 are AGC 3's code plus the LDS form beyond 192
// positions, and replaying a long-read batch's flagged pairs (the heavy ones, as a rule) through the LDS form alone took longer than the batch's main pass (profiles/r06y);
// the secondary-results units exist for AGC 3 and 0 only
#if !defined(PAIRED_SEC) || PAIRED_AGC == 0 || PAIRED_AGC == 3
#ifdef PAIRED_SEC
extern "C" void PE_CAT(snapgpu_launch_paired_sec_exact_, PAIRED_AGC)(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL((k_align_paired<PAIRED_AGC, true, true>), dim3(blocks), dim3(256), lds_bytes, s, *a);
}
#else
extern "C" void PE_CAT(snapgpu_launch_paired_exact_, PAIRED_AGC)(const PairedArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL((k_align_paired<PAIRED_AGC, false, true>), dim3(blocks), dim3(256), lds_bytes, s, *a);
}
#endif
#endif
