// dev_common.h -- device-side structs and wave helpers shared by the gfx950 kernels.
//
// Execution model used throughout: one wavefront (64 lanes) owns one unit of work (one
// read, one LV problem, ...).  Control flow is wave-uniform; lanes cooperate inside the
// primitives (hash-slot probing, hit-list loads, LV diagonals, affine-gap SSE-lane
// emulation).  State that the reference keeps in its per-thread aligner object lives in
// LDS (hot) or in a per-wave slab of HBM scratch (large, L2-resident).
#pragma once
// Landau-Vishkin and the affine-gap forms as ONE function per kernel (lv.h: lv_compute_fn, ag_win.h: ag_dispatch_fn) instead of one inlined
// copy per call site: the paired-end kernels since round 2; the single-end kernels since round 4 -- the row loops then get a register
// allocation of their own instead of sharing the caller's 80 VGPRs / 100 SGPRs (the inlined row loop reloaded ~60 spilled SGPRs per row),
// measured +2 .. 3 % (profiles/r04c: 9.07 -> 9.34 M reads/s with the old row loop, 9.85 -> 10.07 M with the new).  -DSNAPGPU_AG_LV_INLINE
// builds the inlined form.
#if !defined(SNAPGPU_AG_LV_INLINE) && !defined(SNAPGPU_AG_LV_FUNCTIONS)
#define SNAPGPU_AG_LV_FUNCTIONS 1
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>
#if defined(SNAPGPU_WAVE_EMU)
#include <stdio.h>
#include <stdlib.h>
#endif

#define WAVE 64
// Pointers known to point into LDS: inside a function that is NOT inlined into the kernel the compiler cannot see that a generic
// pointer came from the kernel's LDS block and emits FLAT loads (both counters, slower) -- the explicit address space gives ds_read / ds_write.
#if defined(SNAPGPU_WAVE_EMU)
#define LDS_AS
#else
#define LDS_AS __attribute__((address_space(3)))
#endif

// One wavefront's LDS operations, and its vector-memory operations, are each executed in
// issue order by the hardware (LLVM AMDGPU memory model: wavefront scope needs no cache
// action), so a value stored by one lane is visible to the wave's later loads without any
// wait.  What must be prevented is the *compiler* moving a load above the store it depends
// on through another lane (per-thread alias analysis cannot see that dependence).  A pure
// compiler barrier does that without emitting s_waitcnt vmcnt(0), which a
// __builtin_amdgcn_fence would (it made every DP row wait for its traceback stores).
// Pointers known to point into HBM (the per-wave pools, the index blobs): the paired-end kernel keeps its objects in LDS, and a pointer loaded
// back from an object is a GENERIC one -- every access through it a FLAT instruction, which counts on vmcnt AND lgkmcnt, so that a wait for
// an LDS read also waits for every load and store to the pools that is still on its way.  Declared G(T) * the members are global pointers
// and the accesses global_load / global_store.  (Host passes and the emulator: plain pointers, same layout.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SNAPGPU_WAVE_EMU)
#define GLB_AS __attribute__((address_space(1)))
#else
#define GLB_AS
#endif
#define G(T) GLB_AS T
// the value type behind an lvalue that may carry an address space (what a wave-uniform load of it returns)
template <class T> struct strip_as { typedef T type; };
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SNAPGPU_WAVE_EMU)
template <class T> struct strip_as<__attribute__((address_space(1))) T> { typedef T type; };
template <class T> struct strip_as<__attribute__((address_space(3))) T> { typedef T type; };
#endif

// Pointer MEMBERS of the kernels' objects: stored with their address space (LDS: 32 bits; HBM: global), handed out as the plain pointers the
// code works with -- the cast back is an addrspacecast FROM a known space, which the compiler's address-space inference follows through the
// (inlined) users, so the accesses are ds_* / global_* instructions wherever the object itself lives.
template <class T> struct LP {
    LDS_AS T *p;
    LP() = default;
    __host__ __device__ __forceinline__ LP(T *q) : p((LDS_AS T *)q) {}
    __host__ __device__ __forceinline__ operator T *() const { return (T *)p; }
    __host__ __device__ __forceinline__ T *operator->() const { return (T *)p; }
};
template <class T> struct GP {
    GLB_AS T *p;
    GP() = default;
    __host__ __device__ __forceinline__ GP(T *q) : p((GLB_AS T *)q) {}
    __host__ __device__ __forceinline__ operator T *() const { return (T *)p; }
    __host__ __device__ __forceinline__ T *operator->() const { return (T *)p; }
};

#define WAVE_SYNC()                                                   \
    do {                                                              \
        asm volatile("" ::: "memory");                                \
        __builtin_amdgcn_wave_barrier();                              \
    } while (0)

// Index blobs resident in HBM (see DESIGN.md "Data layout in HBM").
struct DevIndex {
    const uint8_t  *hash_blob;      // concatenated slot arrays, reference byte layout
    const uint64_t *table_offset;   // [n_hash_tables] byte offset of table t
    const uint64_t *table_size;     // [n_hash_tables] slots in table t
    const uint32_t *overflow;       // overflow table: [count][loc0 > loc1 > ...]*
    const uint8_t  *genome;         // base 0; genome_pad bytes of 'n' readable on both sides
    const uint64_t *contig_begin;   // [n_contigs]
    uint64_t n_bases;
    uint64_t overflow_size;
    uint64_t first_alt_location;
    uint32_t n_contigs;
    uint32_t seed_len;
    uint32_t key_bytes;
    uint32_t entry_bytes;           // 4*value_count + key_bytes
    uint32_t large;                 // 1: entry carries {fwd value, rc value}
    uint32_t n_hash_tables;
    uint32_t chromosome_padding;
    uint32_t genome_pad;
    // device-native hash layout (bucket.h); bucket_blob == NULL: this index shape keeps the reference's slot walk
    const uint8_t  *bucket_blob;
    const uint64_t *bucket_offset;  // [n_hash_tables] byte offset of table t's buckets
    const uint64_t *n_buckets;      // [n_hash_tables]
    // bit-plane shadow of the padded genome (planes.h): 3 words per 64 bases, block 0 = genome - genome_pad; NULL: not built
    const unsigned long long *planes;
};

// Probability tables, computed on the host with host libm (LandauVishkin.cpp:716-763,
// mapq.h:31-68) and uploaded, so that every FP64 value the kernels multiply is the value
// the reference multiplies.
#define N_INDEL_PROB   2048
#define N_PERFECT_PROB 1001
struct DevTables {
    double phred[256];               // lv_phredToProbability
    double indel[N_INDEL_PROB];      // lv_indelProbabilities
    double perfect[N_PERFECT_PROB];  // lv_perfectMatchProbability
    double mapq_threshold[72];       // x <= mapq_threshold[m]  <=>  (int)(-10*log10(x)) >= m  (host log10)
    double seed_prob;                // pow(1 - SNP_PROB, seedLen) where seedLen is an int: std::pow(double, int) == __builtin_powi in the reference's C++98
                                     // build (BaseAligner.cpp:1314 after :1141, IntersectingPairedEndAligner.cpp:3272 / :3379 / :3487 after their local `int seedLen`)
    double seed_prob_pow;            // the same expression where seedLen is the `unsigned` MEMBER (BaseAligner.cpp:907, BaseAligner.h:434): that call goes
                                     // to the promoting template, i.e. libm's pow(double, double) -- one ulp away at seed 20 (0x1.f5db509cb1434p-1 against
                                     // ...435p-1), which decides ties between equally good candidates of BaseAligner::alignAffineGap
    uint32_t wrapped_seed[33];       // GetWrappedNextSeedToTest(seedLen, wrapCount), SeedSequencer.cpp:36-109
};

// wave ballot as one compare into an SGPR pair (HIP's BALLOT() materialises the predicate in a VGPR first)
#define BALLOT(pred) ((unsigned long long)__builtin_amdgcn_ballot_w64((bool)(pred)))

// Issue priority of this wave among the waves of its SIMD (s_setprio, 0 .. 3).  A unit of work that turns out to be heavy -- a read out of
// a diverged repeat family scored against hundreds of copies, a pair with thousands of Phase-4 candidates -- is what a launch ends on
// (profiles/r03c: 65 of 1 M reads take 110-220 ms each while the average is 0.6 ms), and the SIMD it runs on is shared with up to five
// other waves that are not issue-bound: raising the heavy wave's priority shortens the launch's critical path at no cost in throughput.
#define WAVE_PRIO_HEAVY_AFTER 8        // affine-gap calls of one unit before its wave asks for priority
static __device__ __forceinline__ void wave_set_priority(int heavy) {
    if (heavy) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
}

static __device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }

static __device__ __forceinline__ uint32_t bcast_u32(uint32_t v, int src_lane = 0) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane);
}
static __device__ __forceinline__ int bcast_i32(int v, int src_lane = 0) {
    return __builtin_amdgcn_readlane(v, src_lane);
}
// wave-wide fill of n bytes at a 16-byte aligned address with zero (n a multiple of 16)
static __device__ __forceinline__ void wave_zero16(uint8_t *p, size_t n) {
    struct alignas(16) Z16 { uint64_t a, b; };
    Z16 *q = (Z16 *)p;
    const size_t nq = n >> 4;
    for (size_t i = (size_t)lane_id(); i < nq; i += WAVE) q[i] = Z16{0ull, 0ull};
}

static __device__ __forceinline__ uint32_t first_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
static __device__ __forceinline__ uint64_t first_u64(uint64_t v) {
    uint32_t lo = first_u32((uint32_t)v), hi = first_u32((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// Byte stores of the affine-gap traceback rows: a wave-uniform slab, a wave-uniform row offset and a 32-bit per-lane offset.  Spelled as a
// raw buffer store (descriptor + scalar offset in SGPRs, lane offset in one VGPR) because the compiler, left with a plain pointer, hoists
// "base + lane" into a 64-bit VGPR pair -- which the 80-VGPR builds spill, so that every row of the affine-gap loop reloaded it from scratch
// and waited on vmcnt(0) (i.e. on the previous row's store as well) before it could store.  (An inline-asm global_store with an SGPR base
// faulted on hardware: nothing pads the wait states between such a store and the next write of its address registers.)
struct BtSink {
#if defined(SNAPGPU_WAVE_EMU)
    uint8_t *base; uint32_t n_bytes;     // (the emulator checks what the hardware descriptor silently drops)
#else
    __amdgpu_buffer_rsrc_t rsrc;
#endif
};
// (n_bytes = size of the slab: the descriptor's range check drops a store that would land outside it)
static __device__ __forceinline__ BtSink bt_sink(uint8_t *ubase, uint32_t n_bytes) {
    BtSink s;
#if defined(SNAPGPU_WAVE_EMU)
    s.base = ubase; s.n_bytes = n_bytes;
#else
    s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)first_u64((uint64_t)(uintptr_t)ubase), (short)0, (int)first_u32(n_bytes), 0x00020000);
#endif
    return s;
}
// EXACT traceback images (ag.h): a cell's byte uses bits 0, 1, 2 and 5; the other four carry the TAG of the read that wrote it (1 .. 15, the
// wave's read counter modulo 15), and a cell whose tag is not the current read's reads as zero -- which is what "every read starts with
// zeroed arrays" means.  The images are then cleared once per fifteen reads instead of once per read (57 KB of stores per read, half of
// the kernel's HBM writes: profiles/r04z).  Tag 0 = untagged images (the one-call test entries, the resolver's own images).
#define BT_TAG_MASK 0xD8u
static __host__ __device__ __forceinline__ uint32_t bt_tag_bits(uint32_t epoch) { return ((epoch & 3u) << 3) | ((epoch & 12u) << 4); }
static __device__ __forceinline__ int bt_cell(int raw, uint32_t tag) { return ((uint32_t)raw & BT_TAG_MASK) == tag ? (raw & 0x27) : 0; }

// sink[uoff + voff] = val   (uoff wave-uniform, voff per lane)
static __device__ __forceinline__ void bt_store(const BtSink &s, uint32_t uoff, uint32_t voff, uint32_t val) {
#if defined(SNAPGPU_WAVE_EMU)
    if ((size_t)uoff + voff >= s.n_bytes) { fprintf(stderr, "emulator: traceback store at %zu outside a slab of %u bytes\n", (size_t)uoff + voff, s.n_bytes); abort(); }
    s.base[(size_t)uoff + voff] = (uint8_t)val;
#else
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)val, s.rsrc, (int)voff, (int)first_u32(uoff), 0);
#endif
}

// Event counters of the wavefront emulator's build only (tests/emu, scripts/emu_stats.py: how many Landau-Vishkin levels, affine-gap rows, lazy-F
// rounds ... a read costs): nothing in a device build.
#if defined(SNAPGPU_WAVE_EMU) && defined(SNAPGPU_AG_WIN_STATS)
extern unsigned long long g_emu_stats[64];
#define EMU_STAT(i, n) do { if (lane_id() == 0) __atomic_fetch_add(&g_emu_stats[i], (unsigned long long)(n), __ATOMIC_RELAXED); } while (0)
#else
#define EMU_STAT(i, n) do { } while (0)
#endif

// A wave-uniform 64-bit lane mask as a per-lane predicate at no VALU cost: the mask goes to VCC (or stays in its SGPR pair) and the consumer is
// a v_cndmask_b32_e64 / an exec update that reads it directly (llvm.amdgcn.inverse.ballot).  `mask` must be wave-uniform.
static __device__ __forceinline__ bool lane_in(unsigned long long mask) {
#if defined(SNAPGPU_WAVE_EMU)
    return ((mask >> lane_id()) & 1ull) != 0ull;
#else
    return __builtin_amdgcn_inverse_ballot_w64(mask);
#endif
}

static __device__ __forceinline__ double first_f64(double v) {
    return __longlong_as_double((long long)first_u64((uint64_t)__double_as_longlong(v)));
}

// A0 G1 C2 T3, everything else 4 (Tables.cpp:52-58).
static __device__ __forceinline__ uint32_t base_value(uint8_t c) {
    return c == 'A' ? 0u : c == 'G' ? 1u : c == 'C' ? 2u : c == 'T' ? 3u : 4u;
}
// rcTranslationTable of BaseAligner.cpp:199-210: ACGT complemented, everything else 'N'.
static __device__ __forceinline__ uint8_t rc_base(uint8_t c) {
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
}
