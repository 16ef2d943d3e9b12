// rt_kernels.hpp — what the two kernel translation units share: rt_gpu.hip (pipeline kernels, launch geometry, C ABI)
// and rt_bvh.hip (the persistent triangle-BVH kernel, compiled with its own instruction-scheduling strategy).
#pragma once
#include <hip/hip_runtime.h>

#include "rt_pipeline.hpp"

#ifndef RT_BLOCK
#define RT_BLOCK 256
#endif
#ifndef RT_MAX_GRID
#define RT_MAX_GRID (256 * 8)   // persistent k_bvh, and the default of the grid-stride kernels
#endif
#ifndef RT_LDS_STACK
// k_bvh: stack entries per lane kept in LDS; deeper ones spill to global memory.  Round 3: 12 (26 KB per block) so that
// six blocks fit a CU's 160 KB — with the register diet of rt_bvh.hip / RayCull4 (80 VGPRs, no scratch) the kernel runs
// 6 waves per SIMD instead of 5: k_bvh<AO> 102.9 -> 98.5 ms on config 3.  Measured around it: 14 entries at 5 waves
// 102.9, 12 at 5 waves 104.1, 10 at 6 waves 103.6, 10 at 7 waves (3 dwords of scratch) 98.5, 16 at 4 waves 116.7.
// Later in round 3 the lane's parked leaf moved from LDS into a register (rt_bvh.hip), which made room for a 13th entry
// (26 KB per block, six blocks per CU as before): 89.5 -> 88.8 ms.
#define RT_LDS_STACK 13
#endif
#ifndef RT_BVH_BLOCK
#define RT_BVH_BLOCK RT_BLOCK   // threads per block and blocks of the persistent kernel (swept 64-512 / 1024-8192: no effect)
#endif
#ifndef RT_BVH_GRID
#define RT_BVH_GRID RT_MAX_GRID
#endif

namespace rt {

// cls 0 closest-hit queries, 1 shadow rays, 2 AO rays.  The ray counts (cls 1, 2) are always kept;
// the traversal counters only in the STATS variants.
template <bool STATS>
__device__ inline void flush_counters(const Counters& c, unsigned long long* stats, int cls)
{
    if (cls != 0) {
        uint32_t r = c.rays;
        for (int off = 32; off > 0; off >>= 1) r += __shfl_xor(r, off);
        if ((threadIdx.x & 63) == 0 && r) atomicAdd(&stats[cls == 1 ? ST_SHADOW : ST_AO], (unsigned long long)r);
    }
    if (!STATS) return;
    uint32_t n = c.nodes, t = c.tris, g = c.gates;
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_xor(n, off); t += __shfl_xor(t, off); g += __shfl_xor(g, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&stats[ST_NODES + cls], (unsigned long long)n);
        atomicAdd(&stats[ST_TRIS + cls], (unsigned long long)t);
        atomicAdd(&stats[ST_GATES + cls], (unsigned long long)g);
    }
}


// rt_bvh.hip: k_bvh<ao, stats> on RT_BVH_GRID x RT_BVH_BLOCK over the queue counted by counters[counter]
void launch_k_bvh(int mode /* BVH_CLOSEST, BVH_AO, BVH_SHADOW */, bool stats, hipStream_t stream, const DScene& S, const Batch& B, int counter, int next_counter);

} // namespace rt
