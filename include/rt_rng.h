/*
 * rt_rng.h — the counter-based generator that replaces the reference's shared, unseeded
 * rand() / std::mt19937 state (main.cpp:1010-1011, 1375-1376, 520-521, 543-544).
 *
 * One value is a pure function of (seed, pixel, sample, counter), so a render does not depend
 * on launch order, tile split or GPU count.  The value range is that of glibc's rand()
 * ([0, 2^31-1] = [0, RAND_MAX]) so the reference's own conversion formulas are applied to it
 * unchanged.  The same header is compiled into the HIP kernels, the CPU oracle and the
 * reference harness (which feeds it to the reference code in place of rand()).
 *
 * counter layout per primary sample:
 *   0, 1                      pixel-sample jitter x, y          (Jitter::generateSamples order)
 *   2 + a*2*n*n + (x*n+y)*2 + {0,1}   ambient-occlusion sample (x,y) of the a-th Ambient::L
 *                             evaluation along the path (n = occlusion_samples); this is the
 *                             order in which the reference calls rand() (main.cpp:1371-1376).
 */
#ifndef RT_RNG_H
#define RT_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ static inline
#else
#define RT_HD static inline
#endif

#define RT_RAND_MAX 2147483647

RT_HD uint32_t rt_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

/* rt_rand31 in two steps: the part that is the same for every counter of one primary sample ... */
RT_HD uint32_t rt_rand31_prefix(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    uint32_t h = rt_mix32(seed + 0x9E3779B9U);
    h = rt_mix32(h ^ pixel);
    return rt_mix32(h ^ (sample * 0x85EBCA6BU + 0xC2B2AE35U));
}

/* ... and the counter step (the GPU path stores the prefix with each hit and runs only this per AO ray) */
RT_HD uint32_t rt_rand31_finish(uint32_t prefix, uint32_t counter)
{
    return rt_mix32(prefix ^ (counter * 0x27D4EB2FU + 0x165667B1U)) >> 1;
}

RT_HD uint32_t rt_rand31(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t counter)
{
    return rt_rand31_finish(rt_rand31_prefix(seed, pixel, sample), counter);
}

#endif /* RT_RNG_H */
