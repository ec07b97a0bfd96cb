// TEST INFRASTRUCTURE -- wavefront emulator stand-in for <hip/hip_runtime.h>.
//
// Compiling snap_amd/csrc/*.hip with g++ and `-I tests/emu/include` (this directory first on the include
// path) yields tests/emu/_build/libsnapgpu_emu.so: the same C ABI, the same kernels, executed on the host by
// tests/emu/wave_emu.cpp -- one fiber per lane, 64 fibers per wavefront, every cross-lane operation
// (ballot, readlane, readfirstlane, ds_bpermute / __shfl*, DPP, wave_barrier) a rendezvous of the lanes
// that reach it.  It exists so that the CPU test-suite can execute the *device* control flow and lane
// code without a GPU (tests/test_emu_*.py).  It is never loaded by the product: snap_amd.aligner.load_library
// only ever opens snap_amd/libsnapgpu.so, and this header is not on any product include path.
//
// What it does not model: timing, occupancy, the memory hierarchy, and instruction-level lockstep *between*
// rendezvous points (a lane runs from one cross-lane operation to the next on its own).  Code that relies on
// lockstep without a WAVE_SYNC()/cross-lane operation in between would behave differently here; the kernels
// already need WAVE_SYNC() in those places to stop the compiler reordering, so the two requirements coincide.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <functional>
#include <utility>

#define SNAPGPU_WAVE_EMU 1

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ __thread      /* (GNU TLS: no dynamic-initialisation wrapper call) */
#define __restrict__ __restrict

struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
enum Op { OP_BARRIER, OP_BALLOT, OP_READLANE, OP_READFIRST, OP_BPERMUTE, OP_SHFL_UP, OP_SHFL_DOWN, OP_SHFL_XOR, OP_DPP, OP_STORE };
// one cross-lane operation of the calling lane: blocks until every lane of the wave that is going to take part has arrived
uint64_t xlane(Op op, uint64_t a, int64_t b, int64_t c, uint32_t ctrl, void *site);
uint3 thread_idx();
uint3 block_idx();
dim3 block_dim();
dim3 grid_dim();
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body);
}

#define threadIdx (emu::thread_idx())
#define blockIdx (emu::block_idx())
#define blockDim (emu::block_dim())
#define gridDim (emu::grid_dim())

#define EMU_SITE() __builtin_extract_return_addr(__builtin_return_address(0))
// The wrappers are never inlined: their return address identifies the call site in the kernel code, which is how the
// scheduler tells apart lanes that wait at different cross-lane operations (divergent control flow).
#define EMU_NOINLINE static __attribute__((noinline))

EMU_NOINLINE unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return emu::xlane(emu::OP_BALLOT, p ? 1 : 0, 0, 0, 0, EMU_SITE()); }
EMU_NOINLINE int __builtin_amdgcn_readlane(int v, int src) { return (int)emu::xlane(emu::OP_READLANE, (uint32_t)v, src, 0, 0, EMU_SITE()); }
EMU_NOINLINE int __builtin_amdgcn_readfirstlane(int v) { return (int)emu::xlane(emu::OP_READFIRST, (uint32_t)v, 0, 0, 0, EMU_SITE()); }
EMU_NOINLINE int __builtin_amdgcn_ds_bpermute(int addr, int v) { return (int)emu::xlane(emu::OP_BPERMUTE, (uint32_t)v, (addr >> 2) & 63, 0, 0, EMU_SITE()); }
EMU_NOINLINE int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    return (int)emu::xlane(emu::OP_DPP, (uint32_t)src, (uint32_t)old, (row_mask << 8) | (bank_mask << 4) | (bound_ctrl ? 1 : 0), (uint32_t)ctrl, EMU_SITE());
}
EMU_NOINLINE void __builtin_amdgcn_wave_barrier() { emu::xlane(emu::OP_BARRIER, 0, 0, 0, 0, EMU_SITE()); }
static inline void __builtin_amdgcn_fence(int, const char *) {}
static inline uint64_t __builtin_amdgcn_s_memtime() { return __builtin_ia32_rdtsc(); }
static inline void __builtin_amdgcn_s_setprio(int) {}
#define __ATOMIC_ACQ_REL_EMU 4

EMU_NOINLINE int __shfl(int v, int src, int width = 64) { (void)width; return (int)emu::xlane(emu::OP_BPERMUTE, (uint32_t)v, src & 63, 0, 0, EMU_SITE()); }
EMU_NOINLINE int __shfl_up(int v, unsigned d, int width = 64) { (void)width; return (int)emu::xlane(emu::OP_SHFL_UP, (uint32_t)v, d, 0, 0, EMU_SITE()); }
EMU_NOINLINE int __shfl_down(int v, unsigned d, int width = 64) { (void)width; return (int)emu::xlane(emu::OP_SHFL_DOWN, (uint32_t)v, d, 0, 0, EMU_SITE()); }
EMU_NOINLINE int __shfl_xor(int v, int m, int width = 64) { (void)width; return (int)emu::xlane(emu::OP_SHFL_XOR, (uint32_t)v, m, 0, 0, EMU_SITE()); }

EMU_NOINLINE int __all(int p) { unsigned long long m = emu::xlane(emu::OP_BALLOT, p ? 0 : 1, 0, 0, 0, EMU_SITE()); return m == 0; }
EMU_NOINLINE int __any(int p) { unsigned long long m = emu::xlane(emu::OP_BALLOT, p ? 1 : 0, 0, 0, 0, EMU_SITE()); return m != 0; }
static inline void __threadfence_block() {}
static inline void __threadfence() {}
static inline unsigned long long __brevll(unsigned long long v) {
    v = ((v >> 1) & 0x5555555555555555ull) | ((v & 0x5555555555555555ull) << 1);
    v = ((v >> 2) & 0x3333333333333333ull) | ((v & 0x3333333333333333ull) << 2);
    v = ((v >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((v & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(v);
}
static inline unsigned __brev(unsigned v) { return (unsigned)(__brevll(v) >> 32); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
static inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }

// blocks run on several host threads: device-scope atomics are host atomics
static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicMax(unsigned *p, unsigned v) {
    unsigned o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return o;
}
static inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicExch(unsigned *p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicSub(unsigned *p, unsigned v) { return __atomic_fetch_sub(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicCAS(unsigned *p, unsigned expect, unsigned v) { __atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return expect; }
static inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long expect, unsigned long long v) { __atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return expect; }
#include <sched.h>
struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline void emu_yield() { sched_yield(); }

// ------------------------------------------------------------------------------------------------ host runtime API
typedef int hipError_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipErrorOutOfMemory 2
typedef struct emu_stream *hipStream_t;
typedef struct emu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
    size_t sharedMemPerBlock;
    int warpSize;
};

template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *nb, K, int, size_t) { *nb = 8; return hipSuccess; }      // (no occupancy on the emulated device: any grid runs)
hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
extern "C" hipError_t hipHostRegister(void *p, size_t n, unsigned flags);       // (C linkage as in the real runtime: the native tool finds them by name)
extern "C" hipError_t hipHostUnregister(void *p);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st = nullptr);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);

template <typename K, typename... Args>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t, Args... args) {
    emu::launch(grid, block, lds_bytes, [&]() { kernel(args...); });
}
