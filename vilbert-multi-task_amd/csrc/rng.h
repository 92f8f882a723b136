// Counter-based dropout mask: keep(seed, index) is a pure function, so forward and backward (and the
// attention kernels, which never materialise the probability matrix) regenerate identical masks.
#pragma once
#include <stdint.h>

__device__ __forceinline__ uint32_t vb_hash(uint64_t seed, uint64_t idx) {
    // splitmix64 finaliser over seed + idx * golden ratio
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

// true with probability 1 - p
__device__ __forceinline__ bool vb_keep(uint64_t seed, uint64_t idx, float p) {
    return (float)(vb_hash(seed, idx) >> 8) * (1.0f / 16777216.0f) >= p;
}

// Effective seed of a launch: the host-supplied seed plus a per-step term read from device memory (vb_set_seed_epoch).
// A training step replayed from a captured HIP graph re-runs with the SAME host seeds, so the step counter that makes
// every replay draw fresh masks has to live on the device; forward and backward of one step see the same value.
__device__ __forceinline__ uint64_t vb_seed_with_epoch(uint64_t seed, const uint64_t* __restrict__ epoch) {
    return epoch != nullptr ? seed + *epoch * 0xD1B54A32D192ED03ull : seed;
}
