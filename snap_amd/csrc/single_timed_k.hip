// single_timed_k.hip -- the single-end main kernel (192-position exact form) WITH the s_memtime phase timers compiled in
// (align_single.h: Aligner<.., TIMED>).  Launched instead of the production instantiation when the context was created under
// SNAPGPU_PHASE_TIMERS=1: a breakdown run, never the timed one (every clock read drains lgkmcnt; ~40 of them per read).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c single_timed_k.hip
#include <hip/hip_runtime.h>
#include "../../include/snapgpu.h"
#include "dev_common.h"
#include "probe.h"
#include "lv.h"
#include "ag_win.h"
#include "align_single.h"
#include "kernel_common.h"
#include "single_kernel.h"

extern "C" void snapgpu_launch_single_exact_3_timed(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL((k_align_single<3, false, true, true>), dim3(blocks), dim3(256), lds_bytes, s, *a);
}

// the same for the fast form (the main pass when the help for heavy reads is on: se_help.h)
extern "C" void snapgpu_launch_single_3_timed(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL((k_align_single<3, false, false, true>), dim3(blocks), dim3(256), lds_bytes, s, *a);
}
