// order.h -- heavy-first dequeue order for a batch (single-end reads or pairs).
//
// A launch is persistent waves pulling units from a counter, so it lasts until its slowest unit is done; the cost of a unit is heavy-tailed
// (a read out of a diverged repeat family is scored against every copy: hundreds of affine-gap problems, tens of milliseconds on one wave,
// against ~0.1 ms for an ordinary read).  Dequeued in batch order, such a unit that starts late leaves the rest of the chip idle while it
// finishes (profiles/r02i: the average wave was resident for 45 % of the launch).  Dequeued heaviest first, the launch ends at
// max(balanced time, heaviest unit).  Results do not depend on the order: units are independent.
//
// weight = total hits of a unit's non-overlapping seeds (seeds the aligner would skip as too popular count 0), bucketed by log2; a counting
// sort in descending bucket order gives the permutation.  Three kernels, ~1 % of an alignment launch.
#pragma once
#include "dev_common.h"
#include "probe.h"

// unit i = reads i * rpu .. i * rpu + rpu - 1 (rpu = 1: single end, 2: pairs)
template <int UNUSED>
__global__ __launch_bounds__(256) void k_unit_weights(DevIndex ix, const uint8_t *bases, const uint64_t *offsets, uint32_t n_units, uint32_t rpu,
                                                      uint32_t max_hits, uint32_t *bucket, uint32_t *hist /* [34] */)
{
    const int lane = lane_id();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const int seed_len = (int)ix.seed_len;
    for (uint32_t i = wave; i < n_units; i += n_waves) {
        uint64_t w = 0;
        for (uint32_t r = 0; r < rpu; r++) {
            const uint64_t b = first_u64(offsets[(size_t)i * rpu + r]), e = first_u64(offsets[(size_t)i * rpu + r + 1]);
            const int len = (int)(e - b);
            for (int off = 0; off + seed_len <= len; off += seed_len) {
                SeedBits seed = pack_seed(bases + b + off, ix.seed_len);
                if (!seed.valid) continue;
                HitList hl[2];
                lookup_seed(ix, seed, hl);
                for (int d = 0; d < 2; d++) {
                    const int64_t nh = (int64_t)first_u64((uint64_t)hl[d].n_hits);
                    if (nh > 0 && nh <= (int64_t)max_hits) w += (uint64_t)nh;
                }
            }
        }
        const uint32_t ww = w > 0xffffffffull ? 0xffffffffu : (uint32_t)w;
        const uint32_t bk = ww == 0 ? 0u : 32u - (uint32_t)__clz(ww);      // 0 .. 32
        if (lane == 0) { bucket[i] = bk; atomicAdd(&hist[bk], 1u); }
    }
}
// hist[b] := first position of bucket b when buckets are laid out from the heaviest down; hist[33] := the number of units
template <int UNUSED>
__global__ void k_unit_weight_prefix(uint32_t *hist)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int b = 32; b >= 0; b--) { const uint32_t c = hist[b]; hist[b] = acc; acc += c; }
        hist[33] = acc;
    }
}
template <int UNUSED>
__global__ void k_unit_weight_scatter(const uint32_t *bucket, uint32_t n_units, uint32_t *hist, uint32_t *order)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_units) order[atomicAdd(&hist[bucket[i]], 1u)] = i;
}
