// single_resolve_k.hip -- the single-end kernel's fast form WITH ag_resolve.h (align_single.h: Aligner<.., RESOLVE>): an affine-gap call
// whose traceback leaves its band is answered on the spot from the list of the object's earlier calls of the read, so there are no
// traceback images and nothing to replay.  Launched as the one pass of a context created under SNAPGPU_SINGLE_RESOLVE=1 (192-position
// variant); an instantiation of its own so that the default kernels do not carry it (not measured on hardware yet: DESIGN.md section 16).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c single_resolve_k.hip
#include <hip/hip_runtime.h>
#include "../../include/snapgpu.h"
#include "dev_common.h"
#include "probe.h"
#include "lv.h"
#include "ag_win.h"
#include "align_single.h"
#include "kernel_common.h"
#include "single_kernel.h"

extern "C" void snapgpu_launch_single_resolve_3(const AlignArgs *a, uint32_t blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL((k_align_single<3, false, false, false, false, true>), dim3(blocks), dim3(256), lds_bytes, s, *a);
}
