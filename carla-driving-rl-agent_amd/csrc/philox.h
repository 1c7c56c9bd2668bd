// Philox-4x32-10 counter RNG (Random123), shared by the Beta sampler and the augmentation kernels.
// Stream = (seed, offset), element index idx (< 2^48): counter = {idx.lo, idx.hi | block << 16, offset.lo, offset.hi},
// key = {seed.lo, seed.hi}; words are consumed in order, and the further blocks of one element are counted in the TOP 16 BITS OF
// COUNTER WORD 1 -- never in the offset words -- so the streams of different offsets (consecutive rollout steps, ranks) are
// disjoint whatever number of blocks an element consumes (a Marsaglia-Tsang Beta draw uses >= 2).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace cdrl {

// stream contract, checked by the host entry points (philox_words; beta_sample and augment_images index far below it: rows x actions
// and pixels of one call): element index < 2^48, at most 2^16 blocks (2^18 words) per element
constexpr uint64_t PHILOX_MAX_ELEMENTS = (uint64_t)1 << 48;
constexpr int PHILOX_MAX_BLOCKS = 1 << 16;

struct Philox {
    uint32_t c[4], k[2], out[4];
    int used;
    __device__ Philox(uint64_t seed, uint64_t offset, uint64_t idx) {
        k[0] = (uint32_t)seed;
        k[1] = (uint32_t)(seed >> 32);
        c[0] = (uint32_t)idx;
        c[1] = (uint32_t)(idx >> 32) & 0xFFFFu;
        c[2] = (uint32_t)offset;
        c[3] = (uint32_t)(offset >> 32);
        used = 4;
    }
    __device__ void round(uint32_t* ctr, const uint32_t* key) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ ctr[1] ^ key[0];
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ ctr[3] ^ key[1];
        const uint32_t n3 = (uint32_t)p0;
        ctr[0] = n0; ctr[1] = n1; ctr[2] = n2; ctr[3] = n3;
    }
    __device__ void refill() {
        uint32_t ctr[4] = {c[0], c[1], c[2], c[3]};
        uint32_t key[2] = {k[0], k[1]};
        for (int i = 0; i < 10; ++i) {
            round(ctr, key);
            key[0] += 0x9E3779B9u;
            key[1] += 0xBB67AE85u;
        }
        out[0] = ctr[0]; out[1] = ctr[1]; out[2] = ctr[2]; out[3] = ctr[3];
        // next block of the stream for this element
        c[1] += 0x10000u;
        used = 0;
    }
    __device__ double uniform() {       // (0, 1), 32 bits
        if (used >= 4) refill();
        return ((double)out[used++] + 0.5) * (1.0 / 4294967296.0);
    }
    __device__ double normal() {        // Box-Muller (one of the pair)
        const double u1 = uniform(), u2 = uniform();
        return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
};

}  // namespace cdrl
