// rt_gpu.hip — gfx950 kernels of the wavefront pipeline and the C ABI of include/rt_amd.h.
//
// One context per GPU (one process per GPU); all work of a render is queued on one HIP stream.
// Kernel bodies live in rt_pipeline.hpp / rt_core.hpp; this file supplies the launch geometry
// (256-thread workgroups = 4 wave64, LDS-resident traversal stacks laid out [entry][lane] so a
// wave's accesses to one stack level are bank-conflict free, grid-stride loops whose item counts
// are read from device-side queue counters so no host round trip separates the depth levels),
// the buffers, and the entry points.  There is no CPU fallback behind any entry point.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rt_kernels.hpp"
#include "rt_lbvh.hpp"
#include "rt_scene.hpp"

using namespace rt;

// Grid-stride kernels: a grid slightly above what is resident at once ends in a thin second round of blocks, so
// a kernel launches either what fits in one round (even work per item: k_prep, k_shade at 7 waves/SIMD hold 1792
// blocks; 1536 measured best) or many rounds of short blocks that the dispatcher balances (uneven work per item:
// the gate walks).  Measured on config 3 (ms per frame): k_ao 55.3 / 52.3 / 50.0 / 48.8 / 46.3 / 51.0 at 2048 /
// 4096 / 8192 / 16384 / 32768 / 65536 blocks; k_shadow 11.6 / 11.2 / 11.0 at 2048 / 4096 / 8192 (256-thread blocks); k_prep 5.3 -> 4.8
// and k_shade 4.7 -> 4.0 at 1536; k_trace best at 2048.
#ifndef RT_GRID_TRACE
#define RT_GRID_TRACE 4096   // round 3, one batch per 4K frame: k_trace 2.22 -> 1.96 ms against 2048 blocks
#endif
#ifndef RT_GRID_PREP
#define RT_GRID_PREP 1024   // re-swept at the end of round 3 on the one-batch frame (512 ... 8192): k_prep 4.47 -> 4.40 ms
#endif
#ifndef RT_GRID_SHADOW
#define RT_GRID_SHADOW 16384   // blocks of RT_SHADOW_BLOCK threads
#endif
#ifndef RT_GRID_AO
#define RT_GRID_AO 32768
#endif
#ifndef RT_GRID_SHADE
#define RT_GRID_SHADE 1792   // 7 blocks per CU = the kernel's 7 waves per SIMD, one resident round: k_shade 4.80 -> 4.58 ms (2048: 6.2)
#endif
constexpr uint32_t rt_max_u32(uint32_t a, uint32_t b) { return a > b ? a : b; }

// ---- kernels -------------------------------------------------------------------------------------
// The first RT_LDS_GATES nodes of the gate array (all of it for the reference's scenes) are staged
// in LDS once per block: the walk reads one 48-byte node per step, every lane of a wave mostly the
// same one, and an LDS broadcast read has a fraction of the L1/L2 round trip's latency.
#define RT_LDS_GATES 64
__device__ inline void stage_gates(DScene& S, GateNode* s_gates)
{
    const uint32_t nf = S.num_gates < RT_LDS_GATES ? S.num_gates : RT_LDS_GATES;
    for (uint32_t k = threadIdx.x; k < nf * 3u; k += blockDim.x) ((q4*)s_gates)[k] = ((const q4*)S.gates)[k];
    __syncthreads();
    S.gates_fast = s_gates; S.num_gates_fast = nf;
}

__global__ void __launch_bounds__(RT_BLOCK) k_raygen(DScene S, RenderArgs A, Batch B, uint32_t n)
{
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += gridDim.x * RT_BLOCK) raygen_item(S, A, B, i);
}

// One wave per entry of Batch::ao_ovf (k_trace_overflow, k_ao_overflow); normally there is none and the blocks leave at once.
#define RT_OVF_BLOCK 64

// The closest-hit kernel traverses nothing (trace_item): a stack of RT_DEFER_ROOTS mesh roots per lane is all it needs.
template <bool STATS>
__global__ void __launch_bounds__(RT_BLOCK) k_trace(DScene S, RenderArgs A, Batch B, int depth, uint32_t fixed_n)
{
    __shared__ uint32_t s_stack[RT_DEFER_ROOTS * RT_BLOCK];
    __shared__ GateNode s_gates[RT_LDS_GATES];
    stage_gates(S, s_gates);
    const uint32_t n = depth == 0 ? fixed_n : B.counters[C_QUEUE + depth];
    Stack st{s_stack + threadIdx.x, RT_BLOCK};
    Counters cnt{0, 0, 0, 0};
    QCursor qc{0, 0, queue_chunk(n, gridDim.x * (RT_BLOCK / 64), RT_QCHUNK)};
    // wave-uniform trip count: every lane of a wave takes part in the queue cursor's ballots
    for (uint32_t base = blockIdx.x * RT_BLOCK; base < n; base += gridDim.x * RT_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        bool overflow = false;
        trace_item<STATS>(S, A, B, depth, i, i < n, st, cnt, qc, overflow);
        if (__builtin_amdgcn_ballot_w64(overflow) != 0ull && (threadIdx.x & 63u) == 0u) {   // the wave's rays are redone by k_trace_overflow
            const uint32_t e = atomicAdd(&B.counters[C_TR_OVF + depth], 1u);
            if (e < B.ovf_cap) B.ao_ovf[e] = base + (threadIdx.x & ~63u);
        }
    }
    close_cursor(qc, B.bq_roots, (uint32_t)REF_NONE, B.cq_cap);
    flush_counters<STATS>(cnt, B.stats, 0);
}

template <bool STATS>
__global__ void __launch_bounds__(RT_OVF_BLOCK) k_trace_overflow(DScene S, RenderArgs A, Batch B, int depth, uint32_t fixed_n)
{
    const uint32_t count = B.counters[C_TR_OVF + depth] < B.ovf_cap ? B.counters[C_TR_OVF + depth] : B.ovf_cap;
    if (blockIdx.x >= count) return;
    __shared__ uint32_t s_stack[RT_STACK_DEPTH * RT_OVF_BLOCK];
    __shared__ GateNode s_gates[RT_LDS_GATES];
    stage_gates(S, s_gates);
    const uint32_t n = depth == 0 ? fixed_n : B.counters[C_QUEUE + depth];
    Stack st{s_stack + threadIdx.x, RT_OVF_BLOCK};
    Counters cnt{0, 0, 0, 0};
    for (uint32_t e = blockIdx.x; e < count; e += gridDim.x) trace_overflow_item<STATS>(S, A, B, depth, B.ao_ovf[e], threadIdx.x, n, st, cnt);
    flush_counters<STATS>(cnt, B.stats, 0);
}

__global__ void __launch_bounds__(RT_BLOCK) k_prep(DScene S, RenderArgs A, Batch B, int depth, uint32_t fixed_n)
{
    const uint32_t n = depth == 0 ? fixed_n : B.counters[C_QUEUE + depth];
    QCursor qsh{0, 0, queue_chunk((uint64_t)n * S.num_lights, gridDim.x * (RT_BLOCK / 64), RT_SHCHUNK)};
    // wave-uniform trip count: every lane of a wave takes part in the queue cursor's ballots
    uint32_t skipped = 0;
    for (uint32_t base = blockIdx.x * RT_BLOCK; base < n; base += gridDim.x * RT_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        prep_item(S, A, B, depth, i, i < n, qsh, skipped);
    }
    close_cursor(qsh, B.sh_queue, 0xFFFFFFFFu, B.sh_cap);
    for (int off = 32; off > 0; off >>= 1) skipped += __shfl_xor(skipped, off);
    if ((threadIdx.x & 63u) == 0u && skipped) atomicAdd(&B.stats[ST_SKIPPED], (unsigned long long)skipped);
}

// k_shadow traverses inline, lane by lane: half-size blocks (2 waves) even out its uneven rays a little better (11.1 -> 10.3 ms)
#ifndef RT_SHADOW_BLOCK
#define RT_SHADOW_BLOCK 128
#endif
template <bool STATS, int DEPTH>
__global__ void __launch_bounds__(RT_SHADOW_BLOCK) k_shadow(DScene S, RenderArgs A, Batch B, int depth)
{
    // DEPTH entries per lane (24 / 32 / 48, chosen at launch from the depth of the scene's triangle BVHs, RT_STACK_NEED): at 48
    // the 24 KB of a block leave 3 waves per SIMD; config 4 (trees of <= 14 levels) takes 51.6 ms at 48, 37.3 at 32, 35.6 at 24
    __shared__ uint32_t s_stack[DEPTH * RT_SHADOW_BLOCK];
    __shared__ GateNode s_gates[RT_LDS_GATES];
    stage_gates(S, s_gates);
    const uint32_t n = B.counters[C_SH + depth] < B.sh_cap ? B.counters[C_SH + depth] : B.sh_cap;
    Stack st{s_stack + threadIdx.x, RT_SHADOW_BLOCK};
    Counters cnt{0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * RT_SHADOW_BLOCK + threadIdx.x; i < n; i += gridDim.x * RT_SHADOW_BLOCK)
        shadow_item<STATS, DEPTH>(S, A, B, depth, i, st, cnt);
    flush_counters<STATS>(cnt, B.stats, 1);
}

// Shadow rays of scenes with DScene::shadow_defer (shadow_defer_item): gate walk + queue append, no traversal.
template <bool STATS>
__global__ void __launch_bounds__(RT_BLOCK) k_shadow_defer(DScene S, RenderArgs A, Batch B, int depth)
{
    __shared__ uint32_t s_stack[RT_DEFER_ROOTS * RT_BLOCK];
    __shared__ GateNode s_gates[RT_LDS_GATES];
    stage_gates(S, s_gates);
    const uint32_t n = B.counters[C_SH + depth] < B.sh_cap ? B.counters[C_SH + depth] : B.sh_cap;
    Stack st{s_stack + threadIdx.x, RT_BLOCK};
    Counters cnt{0, 0, 0, 0};
    QCursor qc{0, 0, queue_chunk(n, gridDim.x * (RT_BLOCK / 64), RT_QCHUNK)};
    // wave-uniform trip count: every lane of a wave takes part in the queue cursor's ballots
    for (uint32_t base = blockIdx.x * RT_BLOCK; base < n; base += gridDim.x * RT_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        bool overflow = false;
        shadow_defer_item<STATS>(S, A, B, depth, i, i < n, st, cnt, qc, overflow);
        if (__builtin_amdgcn_ballot_w64(overflow) != 0ull && (threadIdx.x & 63u) == 0u) {   // the wave's rays are redone by k_shadow_overflow
            const uint32_t e = atomicAdd(&B.counters[C_SH_OVF + depth], 1u);
            if (e < B.ovf_cap) B.ao_ovf[e] = base + (threadIdx.x & ~63u);
        }
    }
    close_cursor(qc, B.bq_roots, (uint32_t)REF_NONE, B.bq_cap);
    flush_counters<STATS>(cnt, B.stats, 1);
}

template <bool STATS>
__global__ void __launch_bounds__(RT_OVF_BLOCK) k_shadow_overflow(DScene S, RenderArgs A, Batch B, int depth)
{
    const uint32_t count = B.counters[C_SH_OVF + depth] < B.ovf_cap ? B.counters[C_SH_OVF + depth] : B.ovf_cap;
    if (blockIdx.x >= count) return;
    __shared__ uint32_t s_stack[RT_STACK_DEPTH * RT_OVF_BLOCK];
    __shared__ GateNode s_gates[RT_LDS_GATES];
    stage_gates(S, s_gates);
    const uint32_t n = B.counters[C_SH + depth] < B.sh_cap ? B.counters[C_SH + depth] : B.sh_cap;
    Stack st{s_stack + threadIdx.x, RT_OVF_BLOCK};
    Counters cnt{0, 0, 0, 0};
    for (uint32_t e = blockIdx.x; e < count; e += gridDim.x) shadow_overflow_item<STATS>(S, A, B, depth, B.ao_ovf[e], threadIdx.x, n, st, cnt);
    cnt.rays = 0;   // counted by k_shadow_defer
    flush_counters<STATS>(cnt, B.stats, 1);
}

// The hot AO kernel traverses nothing (ao_item): a stack of RT_DEFER_ROOTS mesh roots per lane is all it needs.
// One lane = one hit and its n*n rays in turn (the AO frame, 48 bytes, is read once per hit; as one item per ray in a
// blocked stratum-major order the bare item loop — decode, three loads, one store — was 15 of the kernel's 46 ms).
// Round 3: k_ao runs 6 waves per SIMD (77 VGPRs, no scratch) instead of 4.  Its time follows the waves that fit — with LDS
// padding that leaves 4 / 3 / 2 waves per SIMD it takes 39.7 / 47.8 / 65.8 ms on config 3 (T ~ 12 + 108 / waves) — and what
// had held it at 118 VGPRs was not its own state but hoisted constants: (i) the 17 fp64 constants of the AO direction
// (sincos_f64, div_rand_max), pulled out of the stratum loop into 34 vector registers that stayed live through the gate
// walk: they now arrive through scalar registers (RT_K64, rt_core.hpp; 118 -> 98 VGPRs); (ii) ~20 registers of small
// constants hoisted by the machine-level LICM pass, now switched off for this translation unit (Makefile GPU_FLAGS;
// 94 -> 77).  39.6 ms at 4 waves -> 36.2 at 5 -> 34.2 at 6.
#ifndef RT_AO_WAVES
#define RT_AO_WAVES 6
#endif
#ifndef RT_WAVE_WALK_MIN
#define RT_WAVE_WALK_MIN 4u   // gate nodes behind the lead planes from which k_ao<WAVE> is used: with the fused gate classification and 7 waves per SIMD it also wins on config 3 (7 nodes: k_ao 33.3 -> 32.5 ms; 12 until then); the reference's main() scene has 30
#endif
#ifndef RT_AO_WAVES_INSTEP
#define RT_AO_WAVES_INSTEP 7   // k_ao<WAVE>: 71 VGPRs without scratch (the per-lane walk needs 32 bytes of scratch at 7 waves: 33.9 -> 36.3 ms); config 4 k_ao 24.2 -> 23.7 ms at half size
#endif
#ifndef RT_WAVE_WALK_LONG
#define RT_WAVE_WALK_LONG 12u   // ... and from this many, at 8 waves per SIMD (55 VGPRs either way): the longer in-step walk has more scalar-load latency to cover (config 4, 30 nodes: k_ao 320.7 -> 312.6 ms at its stated size; config 3, 7 nodes: 30.5 -> 30.7)
#endif
#define RT_AO_ATTR(W) __attribute__((amdgpu_waves_per_eu(W, W)))
// WAVE: the wave walks the gate array in step, node data through scalar registers (walk_gates_wave): the reference's own
// scenes (configs 1, 2, 4), where the per-lane walk diverges over node kinds.  No LDS copy of the gates is needed then.
template <bool STATS, bool WAVE, int WAVES>
__global__ void __launch_bounds__(RT_BLOCK) RT_AO_ATTR(WAVES) k_ao(DScene S, RenderArgs A, Batch B, int depth, uint32_t fixed_n)
{
    __shared__ uint32_t s_stack[RT_DEFER_ROOTS * RT_BLOCK];
    __shared__ GateNode s_gates[WAVE ? 1 : RT_LDS_GATES];
#ifdef RT_AO_LDS_PAD   // experiment: fewer resident waves (issue- or latency-bound?)
    __shared__ uint32_t s_pad[RT_AO_LDS_PAD / 4];
    if (depth == 0x7FFFFFFF) s_pad[threadIdx.x] = 1u;
#endif
    if (!WAVE) stage_gates(S, s_gates);
    const uint32_t n2 = (uint32_t)(S.ambient.occlusion_samples * S.ambient.occlusion_samples);
    const uint32_t nslots = depth == 0 ? fixed_n : B.counters[C_QUEUE + depth];
    Stack st{s_stack + threadIdx.x, RT_BLOCK};
    Counters cnt{0, 0, 0, 0};
    QCursor qc{0, 0, queue_chunk((uint64_t)nslots * n2, gridDim.x * (RT_BLOCK / 64), RT_QCHUNK)};
    // wave-uniform trip count: every lane of a wave takes part in the queue cursor's ballots
    for (uint32_t base = blockIdx.x * RT_BLOCK; base < nslots; base += gridDim.x * RT_BLOCK) {
        const uint32_t slot = base + threadIdx.x;
        bool overflow = false;
        ao_item<STATS, WAVE>(S, A, B, depth, slot, slot < nslots, st, cnt, qc, overflow);
        if (__builtin_amdgcn_ballot_w64(overflow) != 0ull && (threadIdx.x & 63u) == 0u) {   // queue full: the rays of the wave's 64 slots are redone by k_ao_overflow
            const uint32_t e = atomicAdd(&B.counters[C_AO + depth], 1u);
            if (e < B.ovf_cap) B.ao_ovf[e] = base + (threadIdx.x & ~63u);
        }
    }
    close_cursor(qc, B.bq_roots, (uint32_t)REF_NONE, B.bq_cap);
    flush_counters<STATS>(cnt, B.stats, 2);
}

// The AO gate walk for scenes with MANY gate nodes (the reference's BVHAccel over thousands of shapes: config 5).
// There a ray tests ~100 nodes, the walks of a wave's 64 rays differ in length by 2-3x, and k_ao above idles every
// lane until the wave's longest walk ends (43 % active lanes).  Here the waves are persistent, claim chunks of
// RT_WALK_CHUNK consecutive AO items with one atomic, and their lanes REFILL from the chunk as their walks end, as
// k_bvh does with its queue: every round each lane holding a ray takes up to RT_WALK_REPS gate nodes (gate_step),
// finished rays store their distance, rays with RT_DEFER_ROOTS roots (or at the end of their walk) append a queue
// item, idle lanes fetch the next items once RT_WALK_REFILL of them wait.  Same results as ao_item (only the
// position of a ray's items in the queue changes).  Measured on config 5 (k_ao 56.0 ms): 50.4 ms — the lanes of a
// wave no longer sit on the same gate node, so the node fetches stop coalescing (each lane-step is its own 48-byte
// gather from L2), which eats most of what the fuller waves gain; refill at 8-32 idle lanes, 8-64 steps per round
// and chunks of 128-1024 items are within 3 % of each other, 4096-item chunks or a static split lose the gain to the
// end-of-launch imbalance.  On scenes with few gate nodes (config 3: 2 per ray) k_ao stays in use.
#ifndef RT_WALK_REPS
#define RT_WALK_REPS 32
#endif
#ifndef RT_WALK_REFILL
#define RT_WALK_REFILL 16
#endif
#ifndef RT_GRID_AO_WALK
#define RT_GRID_AO_WALK RT_MAX_GRID   // persistent: the waves claim chunks of items
#endif
#ifndef RT_WALK_CHUNK
#define RT_WALK_CHUNK 512u            // items per claim, a multiple of 64
#endif
// Round 3: 7 waves per SIMD (72 VGPRs, 12 dwords of scratch) instead of 4 — on config 5 the kernel takes 51.2 ms at 4 waves
// (122 VGPRs, round 2), 45.8 at 5, 42.8 at 6, 41.6 at 7 and 48.1 at 8 (64 VGPRs: 23 dwords of scratch).  What freed the
// registers: the fp64 constants through scalar registers and no machine LICM (see k_ao above).  Without SLP vectorisation
// (Makefile GPU_FLAGS: 38.6 -> 36.5 ms at 7 waves) the 8-wave build spills 3 dwords only: 35.7 ms.
#ifndef RT_AOW_WAVES
#define RT_AOW_WAVES 8
#endif
#define RT_AOW_ATTR __attribute__((amdgpu_waves_per_eu(RT_AOW_WAVES, RT_AOW_WAVES)))
template <bool STATS>
__global__ void __launch_bounds__(RT_BLOCK) RT_AOW_ATTR k_ao_walk(DScene S, RenderArgs A, Batch B, int depth, uint32_t fixed_n)
{
    __shared__ uint32_t s_stack[RT_DEFER_ROOTS * RT_BLOCK];
    // no LDS copy of the first gate nodes here: with thousands of nodes the lanes sit on nodes far beyond the first 64, and
    // the two-source node fetch (LDS | global, both executed by a wave whose lanes differ) was a tenth of a step's instructions
    // (gate_step<..., FAST = false>: the by-value DScene is not written here)
    const uint32_t n2 = (uint32_t)(S.ambient.occlusion_samples * S.ambient.occlusion_samples);
    const uint32_t nslots = depth == 0 ? fixed_n : B.counters[C_QUEUE + depth];
    const uint32_t n = nslots * n2;
    const uint32_t waves = gridDim.x * (RT_BLOCK / 64);
    const uint32_t lane = threadIdx.x & 63u;
    if ((uint64_t)(blockIdx.x * (RT_BLOCK / 64) + (threadIdx.x >> 6)) * 64u >= n) return;   // more waves than groups of 64 items
    uint32_t next = 0, end = 0;   // wave-uniform: the claimed range of items still to hand out
    bool exhausted = false;
    Stack st{s_stack + threadIdx.x, RT_BLOCK};
    Counters cnt{0, 0, 0, 0};
    QCursor qc{0, 0, queue_chunk(n, waves, RT_QCHUNK)};
    const uint32_t ng = S.num_gates;
    bool active = false;
    f3 o = F3(0, 0, 0), dir = F3(0, 0, 1);
    RayGate rg = {};
    Hit h; h.t = 0; h.prim = H_MISS; h.u = 0; h.v = 0; h.rank = 0;
    uint32_t dst = 0, sp = 0, i = ng, item = 0, slot_of = 0;
    for (;;) {
        // ---- rays whose walk ended or whose root stack is full: queue item, distance ----
        const bool done = active && i >= ng;
        const bool want = active && sp > 0 && (done || sp >= RT_DEFER_ROOTS);
        const uint32_t slot = chunk_append<RT_QCHUNK>(&B.counters[C_BQA + depth], want, qc, B.bq_roots, (uint32_t)REF_NONE, B.bq_cap);
        if (want) {
            if (slot < B.bq_cap) {
                const q4 a = {o.x, o.y, o.z, h.t};
                const q4 b = {dir.x, dir.y, dir.z, u2f(bq_word(dst, sp))};
                B.bq_a[slot] = a; B.bq_b[slot] = b;
                bq_store_roots(B, slot, st, sp);
            } else {   // queue full: all rays of the 64 AO slots around this ray's are redone by k_ao_overflow (entered once)
                const uint32_t g = slot_of >> 6;
                if (!(atomicOr(&B.ovf_bits[g >> 5], 1u << (g & 31u)) & (1u << (g & 31u)))) {
                    const uint32_t e = atomicAdd(&B.counters[C_AO + depth], 1u);
                    if (e < B.ovf_cap) B.ao_ovf[e] = g << 6;
                }
            }
            sp = 0;
        }
        if (done) { B.ao_key[dst] = t_key(h.t); active = false; }
        // ---- refill ----
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!active);
        const uint32_t nidle = (uint32_t)__popcll(idle);
        if (next >= end && !exhausted && nidle >= RT_WALK_REFILL) {   // claim the next RT_WALK_CHUNK items (whole groups of 64)
            uint32_t b0 = 0;
            if (lane == 0) b0 = atomicAdd(&B.counters[C_AOW + depth], (uint32_t)RT_WALK_CHUNK);
            b0 = __shfl(b0, 0);
            next = b0 < n ? b0 : n; end = b0 + RT_WALK_CHUNK < n ? b0 + RT_WALK_CHUNK : n;
            exhausted = b0 >= n || n - b0 <= RT_WALK_CHUNK;   // n < 2^31: no wrap
        }
        if (next < end && nidle >= RT_WALK_REFILL) {
            const uint32_t mine = next + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (!active && mine < end) {
                item = mine;
                uint32_t slot_ao, k;
                ao_coords(S, nslots, item & ~255u, item & 255u, slot_ao, k);
                const q4 rw = B.ao_w[slot_ao], ro = B.ao_o[slot_ao], ru = B.ao_u[slot_ao];   // one round trip (the records of a path without AO rays are in range, stale, and not used)
                if (f2u(rw.w) != 0xFFFFFFFFu) {   // else: a path without AO rays
                    cnt.rays++;
                    o = xyz(ro);
                    dir = ao_dir(S.ambient, xyz(ru), xyz(rw), f2u(ro.w), f2u(rw.w), k);
                    dst = (uint32_t)ao_index(B, slot_ao, k);
                    slot_of = slot_ao;
                    rg = make_gate(S.cull_pad, S.extent, o, dir);
                    h.t = S.ambient.occlusion_dist; h.prim = H_MISS; h.rank = 0;
                    i = (S.debug & 2u) ? ng : lead_planes_hit<true>(S, o, dir, rg, h);
                    sp = 0;
                    active = true;
                }
            }
            next = next + nidle < end ? next + nidle : end;
        }
        if (__builtin_amdgcn_ballot_w64(active) == 0ull) {
            if (next >= end && exhausted) break;
            continue;
        }
        // ---- RT_WALK_REPS gate nodes per lane ----
#pragma nounroll
        for (int rep = 0; rep < RT_WALK_REPS; ++rep) {
            const bool go = i < ng && sp < RT_DEFER_ROOTS;   // (a lane without a ray holds i >= ng: it starts there and a ray ends there)
            // (two ballots of plain compares: the ballot of their conjunction goes through a select and a second compare)
            if ((__builtin_amdgcn_ballot_w64(i < ng) & __builtin_amdgcn_ballot_w64(sp < RT_DEFER_ROOTS)) == 0ull) break;
            if (go) i = gate_step<STATS, true, false>(S, o, dir, rg, i, h, st, sp, cnt);   // DIST: only the distance leaves the walk
        }
    }
    close_cursor(qc, B.bq_roots, (uint32_t)REF_NONE, B.bq_cap);
    flush_counters<STATS>(cnt, B.stats, 2);
}

// k_ao_walk with the whole gate array resident in LDS as 16-byte compact nodes (QGate, rt_core.hpp) — round 3, built on
// VERDICT r2's proposal, measured, and NOT the production path (RT_DEBUG bit 128 selects it; the tests keep it correct).
// k_ao_walk was bound by the texture-address / L1 path (TA busy 77 %): once the lanes of a wave sit on different gate nodes
// every lane-step is a 48-byte gather of its own cache lines.  Here a step reads 16 bytes from LDS; only a box that the
// ray may enter (about half of those tested) costs a gather, of the exact 32-byte box that decides it.  One
// 1024-thread block per CU holds 120 KB of nodes + the lanes' root stacks (32 KB) of the CU's 160 KB.
// Measured on config 5 (profiles/r03a5_*): TA busy 77 -> 49 %, and 47.2 ms against k_ao_walk's 51.2 when both ran 4 waves
// per SIMD.  But a wave pays the gather whenever ANY of its lanes may enter its box (always, in practice), so the latency
// per wave-step did not fall, and the block geometry pins this kernel at 16 waves per CU, while the register diet of
// round 3 lets k_ao_walk run 28: 41.6 ms.  The global walk at 7 waves per SIMD is the production path.
#define RT_WALKQ_BLOCK 1024
template <bool STATS>
__global__ void __launch_bounds__(RT_WALKQ_BLOCK) k_ao_walk_lds(DScene S, RenderArgs A, Batch B, int depth, uint32_t fixed_n)
{
    __shared__ uint32_t s_stack[RT_DEFER_ROOTS * RT_WALKQ_BLOCK];
    __shared__ QGate s_q[RT_QGATE_MAX];
    for (uint32_t k = threadIdx.x; k < S.num_qgates; k += RT_WALKQ_BLOCK) s_q[k] = S.qgates[k];
    __syncthreads();
    const uint32_t n2 = (uint32_t)(S.ambient.occlusion_samples * S.ambient.occlusion_samples);
    const uint32_t nslots = depth == 0 ? fixed_n : B.counters[C_QUEUE + depth];
    const uint32_t n = nslots * n2;
    const uint32_t waves = gridDim.x * (RT_WALKQ_BLOCK / 64);
    const uint32_t lane = threadIdx.x & 63u;
    if ((uint64_t)(blockIdx.x * (RT_WALKQ_BLOCK / 64) + (threadIdx.x >> 6)) * 64u >= n) return;   // more waves than groups of 64 items (no barrier follows)
    uint32_t next = 0, end = 0;   // wave-uniform: the claimed range of items still to hand out
    bool exhausted = false;
    Stack st{s_stack + threadIdx.x, RT_WALKQ_BLOCK};
    Counters cnt{0, 0, 0, 0};
    QCursor qc{0, 0, queue_chunk(n, waves, RT_QCHUNK)};
    const uint32_t ng = S.num_gates;
    bool active = false;
    f3 o = F3(0, 0, 0), dir = F3(0, 0, 1);
    RayGate rg = {};
    RayCull4 gc = {};   // the ray's constants on the compact gate nodes' grid
    Hit h; h.t = 0; h.prim = H_MISS; h.u = 0; h.v = 0; h.rank = 0;
    uint32_t dst = 0, sp = 0, i = ng, item = 0, slot_of = 0;
    for (;;) {
        // ---- rays whose walk ended or whose root stack is full: queue item, distance ----
        const bool done = active && i >= ng;
        const bool want = active && sp > 0 && (done || sp >= RT_DEFER_ROOTS);
        const uint32_t slot = chunk_append<RT_QCHUNK>(&B.counters[C_BQA + depth], want, qc, B.bq_roots, (uint32_t)REF_NONE, B.bq_cap);
        if (want) {
            if (slot < B.bq_cap) {
                const q4 a = {o.x, o.y, o.z, h.t};
                const q4 b = {dir.x, dir.y, dir.z, u2f(bq_word(dst, sp))};
                B.bq_a[slot] = a; B.bq_b[slot] = b;
                bq_store_roots(B, slot, st, sp);
            } else {   // queue full: all rays of the 64 AO slots around this ray's are redone by k_ao_overflow (entered once)
                const uint32_t g = slot_of >> 6;
                if (!(atomicOr(&B.ovf_bits[g >> 5], 1u << (g & 31u)) & (1u << (g & 31u)))) {
                    const uint32_t e = atomicAdd(&B.counters[C_AO + depth], 1u);
                    if (e < B.ovf_cap) B.ao_ovf[e] = g << 6;
                }
            }
            sp = 0;
        }
        if (done) { B.ao_key[dst] = t_key(h.t); active = false; }
        // ---- refill ----
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!active);
        const uint32_t nidle = (uint32_t)__popcll(idle);
        if (next >= end && !exhausted && nidle >= RT_WALK_REFILL) {   // claim the next RT_WALK_CHUNK items (whole groups of 64)
            uint32_t b0 = 0;
            if (lane == 0) b0 = atomicAdd(&B.counters[C_AOW + depth], (uint32_t)RT_WALK_CHUNK);
            b0 = __shfl(b0, 0);
            next = b0 < n ? b0 : n; end = b0 + RT_WALK_CHUNK < n ? b0 + RT_WALK_CHUNK : n;
            exhausted = b0 >= n || n - b0 <= RT_WALK_CHUNK;   // n < 2^31: no wrap
        }
        if (next < end && nidle >= RT_WALK_REFILL) {
            const uint32_t mine = next + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (!active && mine < end) {
                item = mine;
                uint32_t slot_ao, k;
                ao_coords(S, nslots, item & ~255u, item & 255u, slot_ao, k);
                const q4 rw = B.ao_w[slot_ao], ro = B.ao_o[slot_ao], ru = B.ao_u[slot_ao];   // one round trip (the records of a path without AO rays are in range, stale, and not used)
                if (f2u(rw.w) != 0xFFFFFFFFu) {   // else: a path without AO rays
                    cnt.rays++;
                    o = xyz(ro);
                    dir = ao_dir(S.ambient, xyz(ru), xyz(rw), f2u(ro.w), f2u(rw.w), k);
                    dst = (uint32_t)ao_index(B, slot_ao, k);
                    slot_of = slot_ao;
                    rg = make_gate(S.cull_pad, S.extent, o, dir);
                    gc = make_cull4_from(make_cull_grid(S.cull_pad, S.extent, S.gq_base, S.gq_scale, o, dir));
                    h.t = S.ambient.occlusion_dist; h.prim = H_MISS; h.rank = 0;
                    i = (S.debug & 2u) ? ng : lead_planes_hit<true>(S, o, dir, rg, h);
                    sp = 0;
                    active = true;
                }
            }
            next = next + nidle < end ? next + nidle : end;
        }
        if (__builtin_amdgcn_ballot_w64(active) == 0ull) {
            if (next >= end && exhausted) break;
            continue;
        }
        // ---- RT_WALK_REPS gate nodes per lane ----
#pragma nounroll
        for (int rep = 0; rep < RT_WALK_REPS; ++rep) {
            const bool go = active && i < ng && sp < RT_DEFER_ROOTS;
            if (__builtin_amdgcn_ballot_w64(go) == 0ull) break;
            if (go) i = gate_step_q<STATS>(S, s_q, o, dir, rg, gc, i, h, st, sp, cnt);
        }
    }
    close_cursor(qc, B.bq_roots, (uint32_t)REF_NONE, B.bq_cap);
    flush_counters<STATS>(cnt, B.stats, 2);
}

template <bool STATS>
__global__ void __launch_bounds__(RT_OVF_BLOCK) k_ao_overflow(DScene S, RenderArgs A, Batch B, int depth, uint32_t fixed_n)
{
    const uint32_t count = B.counters[C_AO + depth] < B.ovf_cap ? B.counters[C_AO + depth] : B.ovf_cap;
    if (blockIdx.x >= count) return;
    __shared__ uint32_t s_stack[RT_STACK_DEPTH * RT_OVF_BLOCK];
    __shared__ GateNode s_gates[RT_LDS_GATES];
    stage_gates(S, s_gates);
    const uint32_t nslots = depth == 0 ? fixed_n : B.counters[C_QUEUE + depth];
    Stack st{s_stack + threadIdx.x, RT_OVF_BLOCK};
    Counters cnt{0, 0, 0, 0};
    for (uint32_t e = blockIdx.x; e < count; e += gridDim.x) ao_overflow_item<STATS>(S, A, B, B.ao_ovf[e], threadIdx.x, nslots, st, cnt);
    cnt.rays = 0;   // counted by k_ao
    flush_counters<STATS>(cnt, B.stats, 2);
}

__global__ void __launch_bounds__(RT_BLOCK) k_shade(DScene S, RenderArgs A, Batch B, int depth, uint32_t fixed_n)
{
    const uint32_t n = depth == 0 ? fixed_n : B.counters[C_QUEUE + depth];
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += gridDim.x * RT_BLOCK) shade_item(S, A, B, depth, i);
}

__global__ void __launch_bounds__(RT_BLOCK) k_resolve(DScene S, RenderArgs A, Batch B, uint32_t n, int levels_run)
{
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += gridDim.x * RT_BLOCK) resolve_item(S, A, B, i);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned n2 = (unsigned)(S.ambient.occlusion_samples * S.ambient.occlusion_samples);
        unsigned long long bo = 0;
        for (int d = 1; d < levels_run; ++d) bo += B.counters[C_QUEUE + d];
        B.stats[ST_BOUNCE] += bo;   // shadow and AO rays are counted by their kernels
    }
}

// slab [tile][RT_TILE*RT_TILE][3] of the tiles owned by (rank, world) -> row-major frame
__global__ void __launch_bounds__(RT_BLOCK) k_unpack(const float* slab, float* frame, int width, int x0, int y0, int x1, int y1,
                                                     uint32_t tiles_x, uint32_t tiles_y, int rank, int world, uint32_t owned)
{
    const uint32_t n = owned * RT_TILE * RT_TILE;
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += gridDim.x * RT_BLOCK) {
        uint32_t k = i / (RT_TILE * RT_TILE), l = i % (RT_TILE * RT_TILE);
        uint32_t tile = (uint32_t)rank + k * (uint32_t)world;
        uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
        int x = x0 + (int)(tx * RT_TILE + l % RT_TILE), y = y0 + (int)(ty * RT_TILE + l / RT_TILE);
        if (ty >= tiles_y || x >= x1 || y >= y1) continue;
        const float* s = slab + (size_t)i * 3;
        float* o = frame + ((size_t)y * width + x) * 3;
        o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
    }
}

// saveToFile's conversion (reference main.cpp:1847-1849): static_cast<unsigned char>(p * 255)
// Self-test of the hand-written device arithmetic of the AO direction (rt_selftest_device_math): each against the plain
// formulation the compiler expands itself.  out[0]: sqrt_not_tiny vs sqrtf; out[1] / out[2]: sin / cos of sincos_f64 vs the
// same Horner steps written as fma() on literals; out[3]: div_rand_max vs the fp64 division.
__device__ inline void sincos_f64_plain(double x, double& sn, double& cs)
{
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int n = (int)k & 3;
    sn = (n & 1) ? c : s; cs = (n & 1) ? s : c;
    if (n & 2) sn = -sn;
    if ((n + 1) & 2) cs = -cs;
}
__global__ void __launch_bounds__(RT_BLOCK) k_selftest_math(unsigned long long n, unsigned long long seed, unsigned long long* out)
{
    unsigned bad[4] = {0, 0, 0, 0};
    for (unsigned long long i = (unsigned long long)blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * RT_BLOCK) {
        unsigned long long h = (i + seed) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const uint32_t a = (uint32_t)h, b = (uint32_t)(h >> 32);
        // (0) any bit pattern of the function's stated domain (not a positive number below 2^-96, not a negative denormal), and the AO argument 1 - s*s
        float x = u2f(a);
        if ((x > 0.0f && x < 0x1p-96f) || (x < 0.0f && x > -0x1p-126f)) x = 0.0f;
        const float s01 = (float)(b >> 8) * 0x1p-24f, y = 1.0f - s01 * s01;
        const float r0 = sqrt_not_tiny(x), e0 = sqrtf(x), r1 = sqrt_not_tiny(y), e1 = sqrtf(y);
        if ((f2u(r0) != f2u(e0) && !(r0 != r0 && e0 != e0)) || f2u(r1) != f2u(e1)) bad[0]++;
        // (1, 2) phi = 2*pi*sampleY as ao_dir forms it
        const float phi = 2 * 3.14159265358979323846264338327950288f * ((float)(a >> 8) * 0x1p-24f);
        double sn, cs, sn2, cs2;
        sincos_f64((double)phi, sn, cs);
        sincos_f64_plain((double)phi, sn2, cs2);
        if (__double_as_longlong(sn) != __double_as_longlong(sn2)) bad[1]++;
        if (__double_as_longlong(cs) != __double_as_longlong(cs2)) bad[2]++;
        // (3)
        const int r = (int)(b & 0x7FFFFFFFu);
        if (__double_as_longlong(div_rand_max(r)) != __double_as_longlong((double)r / 2147483647.0)) bad[3]++;
    }
    for (int k = 0; k < 4; ++k) if (bad[k]) atomicAdd(&out[k], (unsigned long long)bad[k]);
}

__global__ void __launch_bounds__(RT_BLOCK) k_quantize(const float* rgb, uint8_t* out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * RT_BLOCK)
        out[i] = (uint8_t)(int)(rgb[i] * 255.0f);
}

struct BmpHeader { uint8_t b[54]; };
// one thread per output byte of a row quad would be finer-grained than needed: one thread per pixel
__global__ void __launch_bounds__(RT_BLOCK) k_bmp24(const float* rgb, uint8_t* file, int width, int height, uint32_t row_bytes, BmpHeader hdr)
{
    const size_t npx = (size_t)width * (size_t)height;
    for (size_t i = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x; i < npx; i += (size_t)gridDim.x * RT_BLOCK) {
        const uint32_t y = (uint32_t)(i / (size_t)width), x = (uint32_t)(i % (size_t)width);
        uint8_t* o = file + 54 + (size_t)((uint32_t)height - 1u - y) * row_bytes + (size_t)x * 3;
        o[2] = (uint8_t)(int)(rgb[i * 3 + 0] * 255.0f);
        o[1] = (uint8_t)(int)(rgb[i * 3 + 1] * 255.0f);
        o[0] = (uint8_t)(int)(rgb[i * 3 + 2] * 255.0f);
        if (x == (uint32_t)width - 1u) for (uint32_t k = (uint32_t)width * 3u; k < row_bytes; ++k) o[3 + (k - (uint32_t)width * 3u)] = 0;
    }
    // the 54 header bytes: threads 0..53 of block 0, whatever the number of pixels (a 4x4 image has 16)
    if (blockIdx.x == 0 && threadIdx.x < 54) file[threadIdx.x] = hdr.b[threadIdx.x];
}

// ---- rt_render_device_noalloc's scratch ---------------------------------------------------------------
// The allocation-free entry point (include/rt_amd.h) keeps the wavefront pipeline's per-batch buffers (ray records,
// hit records, the path / shadow / BVH work queues, AO distances, per-level shading terms, k_bvh's spill stacks, the
// queue counters and the ray statistics) in this file-scope device array instead of hipMalloc'ed memory.  Every GPU
// the library runs on holds its own copy at its own address for the life of the process.  In host code the name
// `scratch` is only a handle for hipGetSymbolAddress: rt_init resolves it once per device, with that device
// current, and host code uses the resolved pointer (g_scratch_ptr[device]) everywhere — kernel arguments,
// hipMemsetAsync — never the array's name.  RT_NOALLOC_SCRATCH_BYTES: include/rt_amd.h.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
#define RT_MAX_DEVICES 64
static unsigned char* g_scratch_ptr[RT_MAX_DEVICES];   // per device, set by rt_init under g_scratch_mu
// End of the last rt_render_device_noalloc call that used this device's array, on whatever context and stream it ran:
// the next one waits for it on the device, so calls through different contexts of one GPU serialise instead of racing
// (calls replayed from a captured graph are the caller's to order: nothing can be recorded for them here).
static hipEvent_t g_scratch_ev[RT_MAX_DEVICES];
static bool g_scratch_ev_recorded[RT_MAX_DEVICES];
static std::mutex g_scratch_mu;

// ---- context / scene objects ------------------------------------------------------------------------
struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    unsigned char* scratch_ptr = nullptr;   // this device's copy of `scratch` (resolved address), RT_NOALLOC_SCRATCH_BYTES long
    std::string error;
    std::mutex mu;
    // cached per-render buffers
    void* batch_mem = nullptr; size_t batch_bytes = 0;
    unsigned long long* d_stats = nullptr;
    uint32_t* d_counters = nullptr;
    float* d_rgb = nullptr; uint32_t* d_prim = nullptr; float* d_t = nullptr; size_t out_pixels = 0;
    // rt_render_multi: this context's tile slab, and (on the gathering context) the slabs of all ranks
    float* d_slab = nullptr; size_t slab_floats = 0;
    float* d_gather = nullptr; size_t gather_floats = 0;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    // Stream ordering.  ev_done: end of the last piece of work that used the context's shared buffers (batch
    // buffers, counters, statistics), on whatever stream it ran: the next one waits for it, so one context carries
    // one render at a time even when callers alternate streams.  ev_null: see order_after_null_stream().
    hipEvent_t ev_done = nullptr, ev_null = nullptr;
    bool done_recorded = false;
    hipEvent_t ev_slab = nullptr;            // rt_render_multi: this rank's slab has landed on the gathering GPU
    std::vector<int> peers_enabled;          // devices this context's device may write to directly (hipDeviceEnablePeerAccess done)
};

struct rt_scene {
    rt_ctx* ctx = nullptr;
    HostScene host;
    bool committed = false;
    DScene dev{};
    std::vector<void*> allocs;
    int builder = RT_BUILDER_HOST_SAH;        // requested (rt_scene_set_builder)
    int builder_used = RT_BUILDER_HOST_SAH;   // what the last commit built the triangle BVHs with
    double build_seconds = 0.0;               // triangle BVHs of the last commit (host: all of HostScene::build)
};

static std::string g_init_error;

#define HIPCHECK(ctx, call)                                                                      \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (ctx)->error = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return RT_ERR_HIP;                                                                   \
        }                                                                                        \
    } while (0)

// A call with stream == NULL runs on the context's own stream.  That stream is created without
// hipStreamNonBlocking, so it is already ordered with the legacy default (null) stream; the explicit event makes
// the contract independent of that flag: everything the caller queued on the null stream before the call
// (e.g. a framework's fill or copy of a buffer handed to us) completes before our kernels touch the buffers.
// Work the caller queued on OTHER streams is not covered: pass that stream instead of NULL.
static int order_after_null_stream(rt_ctx* ctx)
{
    HIPCHECK(ctx, hipEventRecord(ctx->ev_null, nullptr));
    HIPCHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_null, 0));
    return RT_OK;
}
// `st` is about to use the context's shared buffers: wait for the previous user; mark_done() after queueing.
static int order_after_previous_work(rt_ctx* ctx, hipStream_t st)
{
    if (ctx->done_recorded) HIPCHECK(ctx, hipStreamWaitEvent(st, ctx->ev_done, 0));
    return RT_OK;
}
static int mark_done(rt_ctx* ctx, hipStream_t st)
{
    HIPCHECK(ctx, hipEventRecord(ctx->ev_done, st));
    ctx->done_recorded = true;
    return RT_OK;
}

static inline uint32_t grid_for(uint64_t n, uint32_t cap = RT_MAX_GRID)
{
    uint64_t b = (n + RT_BLOCK - 1) / RT_BLOCK;
    if (b < 1) b = 1;
    return (uint32_t)(b > cap ? cap : b);
}

// ---- HIP launcher backend of render_frame -------------------------------------------------------------
namespace {
struct GpuBackend {
    rt_ctx* ctx; DScene S; Batch B; hipStream_t stream; bool stats; bool timing;
    int levels_run;
    struct Span { int cls; size_t e0, e1; };
    std::vector<Span> spans;
    size_t ev_used = 0;
    hipError_t err = hipSuccess;
    bool wave_walk = false;   // k_ao<WAVE>: the gate walk in step (the reference's own scenes: a few dozen gate nodes of mixed kinds)
    bool walk_refill = false; size_t ovf_bits_bytes = 0;   // k_ao_walk instead of k_ao (scenes with many gate nodes)
    uint32_t stack_need = RT_STACK_DEPTH;   // entries an in-place traversal of this scene's triangle BVHs can need (RT_STACK_NEED): sizes k_shadow's LDS stack

    size_t ev()
    {
        if (ev_used == ctx->ev_pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { err = hipErrorOutOfMemory; return 0; } ctx->ev_pool.push_back(e); }
        hipEvent_t e = ctx->ev_pool[ev_used];
        hipError_t r = hipEventRecord(e, stream);
        if (r != hipSuccess) err = r;
        return ev_used++;
    }
    template <class F> void timed(int cls, F&& f)
    {
        size_t a = 0;
        if (timing) a = ev();
        f();
        hipError_t r = hipGetLastError();
        if (r != hipSuccess) err = r;
        if (timing) spans.push_back({cls, a, ev()});
    }
    void begin_batch(const RenderArgs&)
    {
        hipError_t r = hipMemsetAsync(B.counters, 0, sizeof(uint32_t) * C_TOTAL, stream);
        if (r != hipSuccess) err = r;

    }
    void raygen(const RenderArgs& A, uint32_t n) { timed(RT_K_RAYGEN, [&] { hipLaunchKernelGGL(k_raygen, dim3(grid_for(n)), dim3(RT_BLOCK), 0, stream, S, A, B, n); }); }
    // many short blocks for the uneven kernels, but at least 16 items per thread (small frames: the per-block set-up
    // and the per-wave queue chunks would dominate), and never fewer blocks than the default grid offers
    static uint32_t grid_many(uint64_t items, uint32_t cap)
    {
        const uint32_t lo = grid_for(items, RT_MAX_GRID), want = grid_for(items / 16, cap);
        return want > lo ? want : lo;
    }
    // n = paths of the batch: the item count at depth 0 and an upper bound of the device-side counts below it
    uint32_t grid_d(int, uint32_t n, uint32_t cap) { return grid_for(n, cap); }
    void trace(const RenderArgs& A, int d, uint32_t n)
    {
        timed(RT_K_TRACE, [&] {
            if (stats) hipLaunchKernelGGL(k_trace<true>, dim3(grid_d(d, n, RT_GRID_TRACE)), dim3(RT_BLOCK), 0, stream, S, A, B, d, n);
            else hipLaunchKernelGGL(k_trace<false>, dim3(grid_d(d, n, RT_GRID_TRACE)), dim3(RT_BLOCK), 0, stream, S, A, B, d, n);
        });
    }
    void trace_overflow(const RenderArgs& A, int d, uint32_t n)
    {
        timed(RT_K_TRACE, [&] {
            if (stats) hipLaunchKernelGGL(k_trace_overflow<true>, dim3(RT_MAX_GRID), dim3(RT_OVF_BLOCK), 0, stream, S, A, B, d, n);
            else hipLaunchKernelGGL(k_trace_overflow<false>, dim3(RT_MAX_GRID), dim3(RT_OVF_BLOCK), 0, stream, S, A, B, d, n);
        });
    }
    void bvh_trace(const RenderArgs& A, int d)
    {
        timed(RT_K_TRACE_BVH, [&] {
            launch_k_bvh(BVH_CLOSEST, stats, stream, S, B, C_BQ + d, C_BQ_NEXT + d);
        });
    }
    void bvh_ao(const RenderArgs& A, int d)
    {
        if (S.ambient.occlusion_samples <= 0) return;
        timed(RT_K_AO_BVH, [&] {
            launch_k_bvh(BVH_AO, stats, stream, S, B, C_BQA + d, C_BQA_NEXT + d);
        });
    }
    void prep(const RenderArgs& A, int d, uint32_t n) { timed(RT_K_PREP, [&] { hipLaunchKernelGGL(k_prep, dim3(grid_d(d, n, RT_GRID_PREP)), dim3(RT_BLOCK), 0, stream, S, A, B, d, n); }); }
    void shadow(const RenderArgs& A, int d)
    {
        if (S.num_lights == 0) return;
        const uint32_t g = grid_many((uint64_t)B.P * S.num_lights * (RT_BLOCK / RT_SHADOW_BLOCK), d == 0 ? RT_GRID_SHADOW : RT_MAX_GRID);   // the queue cannot hold more than this; deeper levels are usually short
        if (S.shadow_defer) {   // gate walk -> any-hit k_bvh over the queued meshes -> the (normally empty) overflow list
            const uint32_t gd = grid_many((uint64_t)B.P * S.num_lights, d == 0 ? RT_GRID_SHADOW : RT_MAX_GRID);
            timed(RT_K_SHADOW, [&] {
                if (stats) hipLaunchKernelGGL(k_shadow_defer<true>, dim3(gd), dim3(RT_BLOCK), 0, stream, S, A, B, d);
                else hipLaunchKernelGGL(k_shadow_defer<false>, dim3(gd), dim3(RT_BLOCK), 0, stream, S, A, B, d);
            });
            timed(RT_K_SHADOW, [&] { launch_k_bvh(BVH_SHADOW, stats, stream, S, B, C_BQS + d, C_BQS_NEXT + d); });
            timed(RT_K_SHADOW, [&] {
                if (stats) hipLaunchKernelGGL(k_shadow_overflow<true>, dim3(RT_MAX_GRID), dim3(RT_OVF_BLOCK), 0, stream, S, A, B, d);
                else hipLaunchKernelGGL(k_shadow_overflow<false>, dim3(RT_MAX_GRID), dim3(RT_OVF_BLOCK), 0, stream, S, A, B, d);
            });
            return;
        }
        timed(RT_K_SHADOW, [&] {
            if (stack_need <= 24u) {
                if (stats) hipLaunchKernelGGL((k_shadow<true, 24>), dim3(g), dim3(RT_SHADOW_BLOCK), 0, stream, S, A, B, d);
                else hipLaunchKernelGGL((k_shadow<false, 24>), dim3(g), dim3(RT_SHADOW_BLOCK), 0, stream, S, A, B, d);
            } else if (stack_need <= 32u) {
                if (stats) hipLaunchKernelGGL((k_shadow<true, 32>), dim3(g), dim3(RT_SHADOW_BLOCK), 0, stream, S, A, B, d);
                else hipLaunchKernelGGL((k_shadow<false, 32>), dim3(g), dim3(RT_SHADOW_BLOCK), 0, stream, S, A, B, d);
            } else {
                if (stats) hipLaunchKernelGGL((k_shadow<true, RT_STACK_DEPTH>), dim3(g), dim3(RT_SHADOW_BLOCK), 0, stream, S, A, B, d);
                else hipLaunchKernelGGL((k_shadow<false, RT_STACK_DEPTH>), dim3(g), dim3(RT_SHADOW_BLOCK), 0, stream, S, A, B, d);
            }
        });
    }
    void ao(const RenderArgs& A, int d, uint32_t n)
    {
        if (S.ambient.occlusion_samples <= 0) return;
        if (walk_refill) {   // many gate nodes: the refilling walk (k_ao_walk)
            // its "group already entered in the overflow list" bits hold for ONE launch (one depth level of one batch)
            hipError_t r = hipMemsetAsync(B.ovf_bits, 0, ovf_bits_bytes, stream);
            if (r != hipSuccess) err = r;
            if (S.num_qgates && (S.debug & 128u)) {   // experiment switch: the walk over the LDS-resident compact copy (measured slower, see k_ao_walk_lds)
                timed(RT_K_AO, [&] {
                    if (stats) hipLaunchKernelGGL(k_ao_walk_lds<true>, dim3(256), dim3(RT_WALKQ_BLOCK), 0, stream, S, A, B, d, n);
                    else hipLaunchKernelGGL(k_ao_walk_lds<false>, dim3(256), dim3(RT_WALKQ_BLOCK), 0, stream, S, A, B, d, n);
                });
                return;
            }
            const uint32_t gw = grid_many((uint64_t)n * (uint32_t)(S.ambient.occlusion_samples * S.ambient.occlusion_samples), d == 0 ? RT_GRID_AO_WALK : RT_MAX_GRID);
            timed(RT_K_AO, [&] {
                if (stats) hipLaunchKernelGGL(k_ao_walk<true>, dim3(gw), dim3(RT_BLOCK), 0, stream, S, A, B, d, n);
                else hipLaunchKernelGGL(k_ao_walk<false>, dim3(gw), dim3(RT_BLOCK), 0, stream, S, A, B, d, n);
            });
            return;
        }
        const uint32_t g = grid_for(n, d == 0 ? RT_GRID_AO : RT_MAX_GRID);   // one thread per hit (n*n rays each): many short blocks
        timed(RT_K_AO, [&] {
            if (wave_walk && S.num_gates - S.num_lead_planes >= RT_WAVE_WALK_LONG) {
                if (stats) hipLaunchKernelGGL((k_ao<true, true, 8>), dim3(g), dim3(RT_BLOCK), 0, stream, S, A, B, d, n);
                else hipLaunchKernelGGL((k_ao<false, true, 8>), dim3(g), dim3(RT_BLOCK), 0, stream, S, A, B, d, n);
            } else if (wave_walk) {
                if (stats) hipLaunchKernelGGL((k_ao<true, true, RT_AO_WAVES_INSTEP>), dim3(g), dim3(RT_BLOCK), 0, stream, S, A, B, d, n);
                else hipLaunchKernelGGL((k_ao<false, true, RT_AO_WAVES_INSTEP>), dim3(g), dim3(RT_BLOCK), 0, stream, S, A, B, d, n);
            } else {
                if (stats) hipLaunchKernelGGL((k_ao<true, false, RT_AO_WAVES>), dim3(g), dim3(RT_BLOCK), 0, stream, S, A, B, d, n);
                else hipLaunchKernelGGL((k_ao<false, false, RT_AO_WAVES>), dim3(g), dim3(RT_BLOCK), 0, stream, S, A, B, d, n);
            }
        });
    }
    void ao_overflow(const RenderArgs& A, int d, uint32_t n)
    {
        if (S.ambient.occlusion_samples <= 0) return;
        timed(RT_K_AO, [&] {
            if (stats) hipLaunchKernelGGL(k_ao_overflow<true>, dim3(RT_MAX_GRID), dim3(RT_OVF_BLOCK), 0, stream, S, A, B, d, n);
            else hipLaunchKernelGGL(k_ao_overflow<false>, dim3(RT_MAX_GRID), dim3(RT_OVF_BLOCK), 0, stream, S, A, B, d, n);
        });
    }
    void shade(const RenderArgs& A, int d, uint32_t n) { timed(RT_K_SHADE, [&] { hipLaunchKernelGGL(k_shade, dim3(grid_d(d, n, RT_GRID_SHADE)), dim3(RT_BLOCK), 0, stream, S, A, B, d, n); }); }
    void resolve(const RenderArgs& A, uint32_t n) { timed(RT_K_RESOLVE, [&] { hipLaunchKernelGGL(k_resolve, dim3(grid_for(n)), dim3(RT_BLOCK), 0, stream, S, A, B, n, levels_run); }); }
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// carve the per-batch buffers out of one allocation
size_t carve_batch(Batch& B, char* base, uint32_t P, int levels, uint32_t nl, uint32_t n2)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off = align_up(off + bytes, 256); return p; };
    B.P = P;
    for (int k = 0; k < 2; ++k) { B.ray_o[k] = (q4*)take(sizeof(q4) * P); B.ray_d[k] = (q4*)take(sizeof(q4) * P); B.queue[k] = (uint32_t*)take(4ull * P); }
    B.hit = (q4*)take(sizeof(q4) * P);
    B.nrm = (q4*)take(sizeof(q4) * P);
    B.pstate = (uint32_t*)take(4ull * P);
    B.ao_o = (q4*)take(sizeof(q4) * P); B.ao_u = (q4*)take(sizeof(q4) * P); B.ao_w = (q4*)take(sizeof(q4) * P);
    B.ao_slot = (uint32_t*)take(4ull * P);
    // Worst case of k_prep's chunked appends: an append wastes the rest of the wave's chunk only when that rest is
    // smaller than the append itself, so holes < items; plus one open chunk per wave.  (A shadow ray that found no
    // slot would be lost, so this bound is the true worst case; the BVH queues below fall back to inline traversal.)
    B.sh_cap = (uint32_t)std::min<size_t>(2 * (size_t)P * nl + (size_t)RT_GRID_PREP * (RT_BLOCK / 64) * RT_SHCHUNK, 0xFFFF0000u);
    B.sh_queue = (uint32_t*)take(4ull * B.sh_cap);
    B.ao_key = (uint32_t*)take(4ull * P * (n2 ? n2 : 1));
    B.ovf_cap = (uint32_t)(std::max<uint64_t>((uint64_t)P * (n2 ? n2 : 1), B.sh_cap) / 64u + 8u);   // one entry per wave of k_trace / k_shadow_defer / k_ao
    B.ao_ovf = (uint32_t*)take(4ull * B.ovf_cap);
    B.ovf_bits = (uint32_t*)take(4ull * (B.ovf_cap / 32u + 1u));
    B.sh_res = (float*)take(4ull * P * nl);
    B.lvl = (q4*)take(sizeof(q4) * (size_t)P * levels);
    B.lvl_ph = (float*)take(4ull * P * levels * (nl > 1 ? nl - 1 : 1));
    B.lvl_kind = (uint8_t*)take((size_t)P * levels);
    {   // closest-hit pass: one item per path at most, plus the holes of k_trace's chunked appends (one open chunk per wave + chunk tails)
        const uint32_t waves = (uint32_t)std::min<size_t>((size_t)RT_GRID_TRACE * (RT_BLOCK / 64), ((size_t)P + 63) / 64);
        B.cq_cap = (uint32_t)std::min<size_t>((size_t)P + (size_t)waves * queue_chunk(P, waves, RT_QCHUNK) + P / 8, 0xFFFF0000u);
    }
    B.bq_cap = B.cq_cap;
    if (n2) {   // the same for k_ao
        const size_t rays = (size_t)P * n2;
        const uint32_t waves = (uint32_t)std::min<size_t>((size_t)RT_GRID_AO * (RT_BLOCK / 64), (rays + 63) / 64);
        B.bq_cap = std::max(B.bq_cap, (uint32_t)std::min<size_t>(rays + (size_t)waves * queue_chunk(rays, waves, RT_QCHUNK) + rays / 8, 0xFFFF0000u));
    }
    {   // and for k_shadow_defer (at most one item per shadow ray)
        const size_t rays = (size_t)P * nl;
        const uint32_t waves = (uint32_t)std::min<size_t>((size_t)RT_GRID_SHADOW * (RT_BLOCK / 64), (rays + 63) / 64);
        B.bq_cap = std::max(B.bq_cap, (uint32_t)std::min<size_t>(rays + (size_t)waves * queue_chunk(rays, waves, RT_QCHUNK) + rays / 8, 0xFFFF0000u));
    }
    B.bq_a = (q4*)take(sizeof(q4) * (size_t)B.bq_cap);
    B.bq_b = (q4*)take(sizeof(q4) * (size_t)B.bq_cap);
    B.bq_c = (q4*)take(sizeof(q4) * (size_t)B.cq_cap);
    B.bq_roots = (uint32_t*)take(4ull * RT_DEFER_ROOTS * (size_t)B.bq_cap);
    B.bq_spill = (uint32_t*)take(8ull * (RT_STACK_DEPTH - RT_LDS_STACK) * RT_BVH_GRID * RT_BVH_BLOCK);   // {reference, entry distance}
    return off;
}
} // namespace

// ---- C ABI ------------------------------------------------------------------------------------------------
extern "C" {

int rt_version(void) { return RT_AMD_VERSION; }

void rt_params_default(rt_params* p)
{
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->width = 1000; p->height = 1000; p->x1 = 1000; p->y1 = 1000;
    p->num_samples = 4; p->num_sets = 83; p->sampler = RT_SAMPLER_JITTER;
    p->max_depth = 3; p->light_attenuation = 500000.0f; p->seed = 1;
    p->tile_rank = 0; p->tile_world = 1;
}

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_init_error.c_str(); }

int rt_init(int device, rt_ctx** out)
{
    if (!out) { g_init_error = "rt_init: null out pointer"; return RT_ERR_INVALID; }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) { g_init_error = std::string("rt_init: no HIP device (") + hipGetErrorString(e) + "); this library has no CPU fallback"; return RT_ERR_HIP; }
    if (device < 0 || device >= count) { g_init_error = "rt_init: device index out of range"; return RT_ERR_INVALID; }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { g_init_error = hipGetErrorString(e); return RT_ERR_HIP; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) { g_init_error = std::string("rt_init: kernels are built for gfx950 only, device is ") + prop.gcnArchName; return RT_ERR_HIP; }
    if ((e = hipSetDevice(device)) != hipSuccess) { g_init_error = hipGetErrorString(e); return RT_ERR_HIP; }
    if (device >= RT_MAX_DEVICES) { g_init_error = "rt_init: device index beyond RT_MAX_DEVICES"; return RT_ERR_UNSUPPORTED; }
    unsigned char* scratch_dev = nullptr;
    {   // once per device, with that device current: the address of this GPU's copy of `scratch`
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        if (!g_scratch_ptr[device]) {
            void* sp = nullptr;
            if ((e = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch))) != hipSuccess || !sp) {
                g_init_error = std::string("rt_init: hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
                return RT_ERR_HIP;
            }
            hipEvent_t ev = nullptr;
            if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) {
                g_init_error = std::string("rt_init: hipEventCreate(scratch): ") + hipGetErrorString(e);
                return RT_ERR_HIP;
            }
            g_scratch_ev[device] = ev; g_scratch_ev_recorded[device] = false;
            g_scratch_ptr[device] = static_cast<unsigned char*>(sp);
        }
        scratch_dev = g_scratch_ptr[device];
    }
    rt_ctx* c = new (std::nothrow) rt_ctx();
    if (!c) return RT_ERR_NOMEM;
    c->device = device;
    c->scratch_ptr = scratch_dev;
    // a blocking stream (no hipStreamNonBlocking): ordered with the legacy default stream in both directions
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamDefault)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_null, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_slab, hipEventDisableTiming)) != hipSuccess ||
        (e = hipMalloc((void**)&c->d_stats, sizeof(unsigned long long) * ST_TOTAL)) != hipSuccess ||
        (e = hipMalloc((void**)&c->d_counters, sizeof(uint32_t) * C_TOTAL)) != hipSuccess ||
        (e = hipEventCreate(&c->ev_begin)) != hipSuccess || (e = hipEventCreate(&c->ev_end)) != hipSuccess) {
        g_init_error = std::string("rt_init: ") + hipGetErrorString(e);
        delete c;
        return RT_ERR_HIP;
    }
    *out = c;
    return RT_OK;
}

int rt_shutdown(rt_ctx* c)
{
    if (!c) return RT_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->ev_null) (void)hipEventDestroy(c->ev_null);
    if (c->ev_slab) (void)hipEventDestroy(c->ev_slab);
    (void)hipFree(c->batch_mem); (void)hipFree(c->d_stats); (void)hipFree(c->d_counters);
    (void)hipFree(c->d_rgb); (void)hipFree(c->d_prim); (void)hipFree(c->d_t);
    (void)hipFree(c->d_slab); (void)hipFree(c->d_gather);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return RT_OK;
}

int rt_scene_create(rt_ctx* ctx, rt_scene** out)
{
    if (!ctx || !out) return RT_ERR_INVALID;
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return RT_ERR_NOMEM;
    s->ctx = ctx;
    s->host.ambient = rt_ambient{{1.0f, 1.0f, 1.0f}, 0.0f, 0, 1000.0f};
    *out = s;
    return RT_OK;
}

#define RT_SETTER(name, type, field)                                                     \
    int rt_scene_set_##name(rt_scene* s, const type* v, size_t n)                        \
    {                                                                                    \
        if (!s || (n && !v)) return RT_ERR_INVALID;                                      \
        s->host.field.assign(v, v + n);                                                  \
        s->committed = false;                                                            \
        return RT_OK;                                                                    \
    }
RT_SETTER(planes, rt_plane, planes)
RT_SETTER(spheres, rt_sphere, spheres)
RT_SETTER(triangles, rt_triangle, triangles)
RT_SETTER(meshes, rt_mesh, meshes)
RT_SETTER(multimeshes, rt_multimesh, multimeshes)
RT_SETTER(scene, rt_shape_ref, scene)
RT_SETTER(materials, rt_material, materials)
RT_SETTER(lights, rt_light, lights)

int rt_scene_set_bvhs(rt_scene* s, const rt_bvh* v, size_t n, const rt_shape_ref* leaves, size_t num_leaves)
{
    if (!s || (n && !v) || (num_leaves && !leaves)) return RT_ERR_INVALID;
    s->host.bvhs.assign(v, v + n);
    s->host.bvh_leaves.assign(leaves, leaves + num_leaves);
    s->committed = false;
    return RT_OK;
}

int rt_scene_set_textures(rt_scene* s, const rt_texture* v, size_t n)
{
    if (!s || (n && !v)) return RT_ERR_INVALID;
    s->host.textures.assign(v, v + n);
    s->host.texel_store.assign(n, {});
    for (size_t i = 0; i < n; ++i) {
        if (v[i].rgba && v[i].width > 0 && v[i].height > 0) {
            s->host.texel_store[i].assign(v[i].rgba, v[i].rgba + (size_t)v[i].width * v[i].height * 4);
            s->host.textures[i].rgba = s->host.texel_store[i].data();
        } else s->host.textures[i].rgba = nullptr;
    }
    s->committed = false;
    return RT_OK;
}

int rt_scene_set_ambient(rt_scene* s, const rt_ambient* a)
{
    if (!s || !a) return RT_ERR_INVALID;
    s->host.ambient = *a;
    s->committed = false;
    return RT_OK;
}

int rt_scene_set_desc(rt_scene* s, const rt_scene_desc* d)
{
    if (!s || !d) return RT_ERR_INVALID;
    int rc;
    if ((rc = rt_scene_set_planes(s, d->planes, d->num_planes))) return rc;
    if ((rc = rt_scene_set_spheres(s, d->spheres, d->num_spheres))) return rc;
    if ((rc = rt_scene_set_triangles(s, d->triangles, d->num_triangles))) return rc;
    if ((rc = rt_scene_set_meshes(s, d->meshes, d->num_meshes))) return rc;
    if ((rc = rt_scene_set_multimeshes(s, d->multimeshes, d->num_multimeshes))) return rc;
    if ((rc = rt_scene_set_bvhs(s, d->bvhs, d->num_bvhs, d->bvh_leaves, d->num_bvh_leaves))) return rc;
    if ((rc = rt_scene_set_scene(s, d->scene, d->num_scene))) return rc;
    if ((rc = rt_scene_set_materials(s, d->materials, d->num_materials))) return rc;
    if ((rc = rt_scene_set_textures(s, d->textures, d->num_textures))) return rc;
    if ((rc = rt_scene_set_lights(s, d->lights, d->num_lights))) return rc;
    return rt_scene_set_ambient(s, &d->ambient);
}

static void scene_free_device(rt_scene* s)
{
    for (void* p : s->allocs) (void)hipFree(p);
    s->allocs.clear();
    s->dev = DScene{};
}

int rt_scene_commit(rt_scene* s)
{
    if (!s) return RT_ERR_INVALID;
    rt_ctx* ctx = s->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    s->committed = false;   // a commit that fails half-way leaves no renderable scene behind
    HostScene& H = s->host;
    const auto t_build = std::chrono::steady_clock::now();
    H.gpu_bvh = s->builder == RT_BUILDER_GPU_LBVH;
    int rc = H.build();
    if (rc) { ctx->error = "rt_scene_commit: " + H.error; return rc; }
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    scene_free_device(s);
    q4 *lb_tris = nullptr, *lb_nodes = nullptr;
    s->builder_used = RT_BUILDER_HOST_SAH;
    if (H.gpu_bvh) {
        uint32_t depth = 0;
        double secs = 0.0;
        std::string err;
        // inner levels the traversal stacks allow (rt_scene.cpp kMaxDepth + 1); RT_LBVH_MAX_DEPTH lowers it to test the fallback
        const uint32_t max_ok = getenv("RT_LBVH_MAX_DEPTH") ? (uint32_t)atoi(getenv("RT_LBVH_MAX_DEPTH")) : 25u;
        rc = lbvh_build(H, ctx->stream, max_ok, &lb_tris, &lb_nodes, &depth, &secs, err);
        if (rc) { ctx->error = "rt_scene_commit: " + err; return rc; }
        if (depth > max_ok) {   // deeper than the traversal stacks allow (rt_scene.cpp kMaxDepth + 1): the host builder bounds its depth
            (void)hipFree(lb_tris); (void)hipFree(lb_nodes);
            lb_tris = lb_nodes = nullptr;
            H.gpu_bvh = false;
            rc = H.build();
            if (rc) { ctx->error = "rt_scene_commit: " + H.error; return rc; }
        } else {
            s->allocs.push_back(lb_tris); s->allocs.push_back(lb_nodes);
            s->builder_used = RT_BUILDER_GPU_LBVH;
            s->build_seconds = secs;
            H.bvh_max_depth = depth;
        }
    }
    if (s->builder_used == RT_BUILDER_HOST_SAH) s->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build).count();
    auto up = [&](const void* src, size_t bytes, const void** dst) -> int {
        void* d = nullptr;
        size_t b = bytes ? bytes : 16;
        HIPCHECK(ctx, hipMalloc(&d, b));
        s->allocs.push_back(d);
        if (bytes) HIPCHECK(ctx, hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
        *dst = d;
        return RT_OK;
    };
    DScene& D = s->dev;
    if ((rc = up(H.gates.data(), H.gates.size() * sizeof(GateNode), (const void**)&D.gates))) return rc;
    if ((rc = up(H.planes4.data(), H.planes4.size() * sizeof(q4), (const void**)&D.planes))) return rc;
    if ((rc = up(H.spheres4.data(), H.spheres4.size() * sizeof(q4), (const void**)&D.spheres))) return rc;
    if (lb_tris) { D.tris = lb_tris; D.nodes = lb_nodes; }
    else {
        if ((rc = up(H.tris4.data(), H.tris4.size() * sizeof(q4), (const void**)&D.tris))) return rc;
        if ((rc = up(H.nodes4.data(), H.nodes4.size() * sizeof(q4), (const void**)&D.nodes))) return rc;
    }
    if ((rc = up(H.attr4.data(), H.attr4.size() * sizeof(q4), (const void**)&D.tri_attr))) return rc;
    if ((rc = up(H.dmaterials.data(), H.dmaterials.size() * sizeof(DMaterial), (const void**)&D.materials))) return rc;
    if ((rc = up(H.dlights.data(), H.dlights.size() * sizeof(DLight), (const void**)&D.lights))) return rc;
    if ((rc = up(H.dtextures.data(), H.dtextures.size() * sizeof(DTexture), (const void**)&D.textures))) return rc;
    if ((rc = up(H.texels.data(), H.texels.size(), (const void**)&D.texels))) return rc;
    if ((rc = up(H.item_opaque.data(), H.item_opaque.size(), (const void**)&D.item_opaque))) return rc;
    if ((rc = up(H.qgates.data(), H.qgates.size() * sizeof(QGate), (const void**)&D.qgates))) return rc;
    D.num_qgates = (uint32_t)H.qgates.size();
    D.gq_base = F3(H.gq_base[0], H.gq_base[1], H.gq_base[2]); D.gq_scale = F3(H.gq_scale[0], H.gq_scale[1], H.gq_scale[2]);
    D.num_gates = (uint32_t)H.gates.size();
    D.gates_fast = nullptr; D.num_gates_fast = 0;
    D.num_lights = (uint32_t)H.dlights.size();
    D.num_textures = (uint32_t)H.dtextures.size();
    D.num_items = (uint32_t)H.item_opaque.size();
    D.shadow_defer = H.shadow_defer ? 1u : 0u;
    {   // bytes of triangle records + nodes the traversal gathers from (host-built or device-built)
        const size_t tri_bytes = lb_tris ? (size_t)H.gpu_slots * 3 * sizeof(q4) : H.tris4.size() * sizeof(q4);
        const size_t node_bytes = lb_nodes ? (size_t)H.gpu_nodes * 4 * sizeof(q4) : H.nodes4.size() * sizeof(q4);
        D.big_scene = tri_bytes + node_bytes > (256ull << 20) ? 1u : 0u;
    }
    D.ambient = H.ambient;
    D.prim_base_sphere = (uint32_t)H.planes.size();
    D.prim_base_tri = (uint32_t)(H.planes.size() + H.spheres.size());
    D.cull_pad = 1.0f / 262144.0f;
    D.extent = H.extent;
    D.qbase = F3(H.qbase[0], H.qbase[1], H.qbase[2]); D.qscale = F3(H.qscale[0], H.qscale[1], H.qscale[2]);
    H.fill_lead_planes(D);
    D.debug = getenv("RT_DEBUG") ? (uint32_t)atoi(getenv("RT_DEBUG")) : 0u;
    s->committed = true;
    return RT_OK;
}

int rt_scene_set_builder(rt_scene* s, int builder)
{
    if (!s) return RT_ERR_INVALID;
    if (builder != RT_BUILDER_HOST_SAH && builder != RT_BUILDER_GPU_LBVH) { s->ctx->error = "rt_scene_set_builder: unknown builder"; return RT_ERR_INVALID; }
    s->builder = builder;
    s->committed = false;
    return RT_OK;
}

int rt_scene_builder_used(rt_scene* s, int* builder, double* build_seconds)
{
    if (!s) return RT_ERR_INVALID;
    if (!s->committed) { s->ctx->error = "rt_scene_builder_used: scene not committed"; return RT_ERR_INVALID; }
    if (builder) *builder = s->builder_used;
    if (build_seconds) *build_seconds = s->build_seconds;
    return RT_OK;
}

int rt_scene_destroy(rt_scene* s)
{
    if (!s) return RT_ERR_INVALID;
    {
        std::lock_guard<std::mutex> lk(s->ctx->mu);
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
        scene_free_device(s);
    }
    delete s;
    return RT_OK;
}

static int check_params(rt_ctx* ctx, const rt_params* p)
{
    auto bad = [&](const char* m) { ctx->error = std::string("rt_render: ") + m; return (int)RT_ERR_INVALID; };
    if (p->width <= 0 || p->height <= 0) return bad("image size must be positive");
    if (p->x0 < 0 || p->y0 < 0 || p->x1 > p->width || p->y1 > p->height || p->x0 > p->x1 || p->y0 > p->y1) return bad("render rectangle outside the image");
    if (p->num_samples <= 0 || p->num_samples > 65536) return bad("num_samples out of range");
    if (p->sampler < RT_SAMPLER_REGULAR || p->sampler > RT_SAMPLER_RANDOM) return bad("unknown sampler");
    if (p->max_depth < 0 || p->max_depth + 2 > RT_MAX_LEVELS) return bad("max_depth out of range [0,60]");
    if (p->tile_world < 1 || p->tile_rank < 0 || p->tile_rank >= p->tile_world) return bad("tile_rank/tile_world inconsistent");
    if ((uint64_t)p->width * p->height > 0xFFFFFFFFull) return bad("image too large");
    return RT_OK;
}

static void tile_grid(const rt_params* p, uint32_t& tx, uint32_t& ty)
{
    tx = (uint32_t)((p->x1 - p->x0 + RT_TILE - 1) / RT_TILE);
    ty = (uint32_t)((p->y1 - p->y0 + RT_TILE - 1) / RT_TILE);
}

int rt_owned_tiles(const rt_params* p, int rank, int world, uint32_t* out)
{
    if (!p || !out || world < 1 || rank < 0 || rank >= world) return RT_ERR_INVALID;
    uint32_t tx, ty;
    tile_grid(p, tx, ty);
    uint32_t total = tx * ty;
    *out = total > (uint32_t)rank ? (total - (uint32_t)rank + (uint32_t)world - 1) / (uint32_t)world : 0;
    return RT_OK;
}

// fixed_mem != nullptr (rt_render_device_noalloc): every device buffer of the render is carved out of
// [fixed_mem, fixed_mem + fixed_bytes) and nothing is allocated, freed, created or read back.
static int render_locked(rt_scene* s, const rt_camera* cam, const rt_params* p, float* d_rgb, uint32_t* d_prim,
                         float* d_t, rt_stats* stats, hipStream_t stream, bool slab, bool wait,
                         unsigned char* fixed_mem = nullptr, size_t fixed_bytes = 0, bool capturing = false)
{
    rt_ctx* ctx = s->ctx;
    const DScene& S = s->dev;
    RenderArgs A{};
    A.cam = make_camera(*cam);
    A.width = p->width; A.height = p->height; A.x0 = p->x0; A.y0 = p->y0; A.x1 = p->x1; A.y1 = p->y1;
    A.spp = p->num_samples; A.sampler = p->sampler; A.max_depth = p->max_depth; A.levels = p->max_depth + 2;
    A.light_atten = p->light_attenuation; A.seed = p->seed;
    tile_grid(p, A.tiles_x, A.tiles_y);
    A.tile_rank = p->tile_rank; A.tile_world = p->tile_world;
    A.out_rgb = d_rgb; A.out_prim = d_prim; A.out_t = d_t; A.slab = slab ? 1 : 0;
    uint32_t owned = 0;
    rt_owned_tiles(p, p->tile_rank, p->tile_world, &owned);

    const bool bounces = s->host.has_bounce_materials;
    const int levels_alloc = bounces ? A.levels : 1;
    const uint64_t per_tile = (uint64_t)RT_TILE * RT_TILE * (uint64_t)A.spp;
    const uint32_t nl = S.num_lights ? S.num_lights : 1;
    const uint32_t n2 = (uint32_t)(S.ambient.occlusion_samples * S.ambient.occlusion_samples);
    // Paths per batch.  Default: as many as the 31-bit ray ids of the BVH queue items allow (P * n^2 < 2^31) and as fit
    // 85 % of the free device memory, in equal batches — every launch of a persistent kernel ends in a drain tail and
    // sizes its grid for the batch, so fewer, larger batches are faster (8 -> 4: +1 %; round 3: 4 -> 2 with the ids widened
    // from 29 to 31 bits, and 2 -> 1 without the former 64 Mi cap: 4K, 16 spp, 4x4 AO is ONE batch of 2040 tiles, 133.7 M
    // paths, ~208 GB of the 288 GB: 145.0 -> 141.5 ms).
    uint64_t want = p->batch_paths ? p->batch_paths : 0x7FFFFFFFull;
    if (!p->batch_paths) {
        want = std::min<uint64_t>(want, ((1ull << 31) - 1) / (n2 ? n2 : 1));
        want = std::min<uint64_t>(want, 0xFFFFFFFFull / nl);
    }
    uint32_t tiles_per_batch = (uint32_t)std::min<uint64_t>(want / per_tile, 0x7FFFFFFFull);
    if (tiles_per_batch < 1) tiles_per_batch = 1;
    if (tiles_per_batch > owned && owned > 0) tiles_per_batch = owned;
    Batch B{};
    uint32_t P = 0;
    // fixed memory: [counters][statistics][batch buffers]
    const size_t fixed_head = align_up(sizeof(uint32_t) * C_TOTAL, 256) + align_up(sizeof(unsigned long long) * ST_TOTAL, 256);
    if (fixed_mem) {
        if (fixed_bytes <= fixed_head) { ctx->error = "rt_render_device_noalloc: scratch smaller than its header"; return RT_ERR_UNSUPPORTED; }
        const size_t avail = fixed_bytes - fixed_head;
        for (;;) {
            if (owned > 0) {   // equal batches
                const uint32_t nb = (owned + tiles_per_batch - 1) / tiles_per_batch;
                tiles_per_batch = (owned + nb - 1) / nb;
            }
            size_t need = ~(size_t)0;
            const uint64_t paths = (uint64_t)tiles_per_batch * per_tile;
            if (paths <= 0x7FFFFFFFull && paths * nl <= 0xFFFFFFFFull && paths * (n2 ? n2 : 1) < (1ull << 31)) {
                P = (uint32_t)paths;
                need = carve_batch(B, nullptr, P, levels_alloc, nl, n2);
            }
            if (need <= avail) break;
            if (tiles_per_batch == 1) {
                char msg[320];
                std::snprintf(msg, sizeof msg, "rt_render_device_noalloc: one %dx%d tile at %d samples per pixel, %u light(s), %u AO rays per hit and "
                              "%d shading level(s) needs %zu bytes of queues, the fixed scratch holds %zu (RT_NOALLOC_SCRATCH_BYTES = %llu): "
                              "lower num_samples or use rt_render_device", RT_TILE, RT_TILE, A.spp, nl, n2, levels_alloc, need + fixed_head, fixed_bytes,
                              (unsigned long long)RT_NOALLOC_SCRATCH_BYTES);
                ctx->error = msg;
                return RT_ERR_UNSUPPORTED;
            }
            // need grows linearly with the tiles of a batch (plus a constant): aim below the estimate, then step down
            const uint64_t guess = need == ~(size_t)0 ? tiles_per_batch / 2 : (uint64_t)((double)tiles_per_batch * (double)avail / (double)need);
            tiles_per_batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(guess, tiles_per_batch - 1));
        }
    }
    for (; !fixed_mem;) {
        if (!p->batch_paths && owned > 0) {   // equal batches
            const uint32_t nb = (owned + tiles_per_batch - 1) / tiles_per_batch;
            tiles_per_batch = (owned + nb - 1) / nb;
        }
        if ((uint64_t)tiles_per_batch * per_tile > 0x7FFFFFFFull) { ctx->error = "rt_render: one tile x num_samples exceeds 2^31 paths"; return RT_ERR_UNSUPPORTED; }
        P = (uint32_t)(tiles_per_batch * per_tile);
        if ((uint64_t)P * nl > 0xFFFFFFFFull || (uint64_t)P * (n2 ? n2 : 1) >= (1ull << 31)) {   // 31-bit ray ids in the BVH queue items
            ctx->error = "rt_render: batch too large for 32-bit queue entries; lower batch_paths"; return RT_ERR_UNSUPPORTED; }
        const size_t need = carve_batch(B, nullptr, P, levels_alloc, nl, n2);
        if (need <= ctx->batch_bytes) break;
        if (!p->batch_paths && tiles_per_batch > 1) {   // a default batch leaves 15 % of what is free to the caller
            size_t mem_free = 0, mem_total = 0;
            if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && need > (mem_free + ctx->batch_bytes) / 100 * 85) {
                tiles_per_batch = (tiles_per_batch + 1) / 2;
                continue;
            }
        }
        HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->done_recorded) HIPCHECK(ctx, hipEventSynchronize(ctx->ev_done));   // a render still running on a caller's stream
        if (ctx->batch_mem) HIPCHECK(ctx, hipFree(ctx->batch_mem));
        ctx->batch_mem = nullptr; ctx->batch_bytes = 0;
        const hipError_t e = hipMalloc(&ctx->batch_mem, need);
        if (e == hipSuccess) { ctx->batch_bytes = need; break; }
        (void)hipGetLastError();
        ctx->batch_mem = nullptr;
        if (e != hipErrorOutOfMemory || p->batch_paths || tiles_per_batch == 1) {   // a default batch that does not fit is halved
            ctx->error = std::string("rt_render: hipMalloc of the batch buffers: ") + hipGetErrorString(e); return RT_ERR_HIP; }
        tiles_per_batch = (tiles_per_batch + 1) / 2;
    }
    carve_batch(B, fixed_mem ? (char*)fixed_mem + fixed_head : (char*)ctx->batch_mem, P, levels_alloc, nl, n2);
    if (S.debug & 8u) { B.bq_cap = std::min<uint32_t>(B.bq_cap, 3000u); B.cq_cap = std::min<uint32_t>(B.cq_cap, 3000u); }   // test switch: a BVH queue that overflows (inline traversal path)
    B.counters = fixed_mem ? (uint32_t*)fixed_mem : ctx->d_counters;
    B.stats = fixed_mem ? (unsigned long long*)(fixed_mem + align_up(sizeof(uint32_t) * C_TOTAL, 256)) : ctx->d_stats;

    GpuBackend be{ctx, S, B, stream, (p->flags & RT_FLAG_STATS) != 0, (p->flags & RT_FLAG_TIMING) != 0, levels_alloc};
    // the refilling gate walk pays off when a ray tests many gate nodes (config 5: ~100 of 7500; config 3: 2 of 8)
    be.walk_refill = ((S.num_gates > RT_LDS_GATES && !(S.debug & 32u)) || (S.debug & 16u)) && n2 > 0;
    // experiment switches: RT_DEBUG bit 256 forces the in-step walk on, bit 512 off
    be.wave_walk = !be.walk_refill && (((S.num_gates <= RT_LDS_GATES && S.num_gates - S.num_lead_planes >= RT_WAVE_WALK_MIN) && !(S.debug & 512u)) || (S.debug & 256u));
    be.ovf_bits_bytes = 4ull * (B.ovf_cap / 32u + 1u);
    be.stack_need = (S.debug & 1024u) ? (uint32_t)RT_STACK_DEPTH : RT_STACK_NEED(s->host.bvh_max_depth);   // RT_DEBUG bit 1024: the full-depth stack whatever the scene
    if ((S.debug & 4u) || (uint64_t)P * nl >= (1ull << 31)) be.S.shadow_defer = 0;   // experiment switch / the queue item's 31-bit destination
    if (!capturing) {   // the batch buffers, counters and statistics are the context's: wait for their previous user (any stream)
        const int rc = order_after_previous_work(ctx, stream);   // (a capturing stream may not wait on an event recorded outside the capture)
        if (rc) return rc;
    }
    HIPCHECK(ctx, hipMemsetAsync(B.stats, 0, sizeof(unsigned long long) * ST_TOTAL, stream));   // the array keeps the last call's contents
    if (!fixed_mem) HIPCHECK(ctx, hipEventRecord(ctx->ev_begin, stream));
    uint32_t batches = 0;
    if (owned > 0 && p->x1 > p->x0 && p->y1 > p->y0) {
        render_frame(be, A, owned, tiles_per_batch, bounces);
        batches = (owned + tiles_per_batch - 1) / tiles_per_batch;
    }
    if (!fixed_mem) HIPCHECK(ctx, hipEventRecord(ctx->ev_end, stream));
    if (!capturing) {   // (an event recorded inside a capture is no event later eager work could wait on)
        const int rc = mark_done(ctx, stream);
        if (rc) return rc;
    }
    if (be.err != hipSuccess) { ctx->error = std::string("rt_render: launch failed: ") + hipGetErrorString(be.err); return RT_ERR_HIP; }
    if (!wait && !stats) return RT_OK;
    HIPCHECK(ctx, hipStreamSynchronize(stream));
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        unsigned long long h[ST_TOTAL];
        HIPCHECK(ctx, hipMemcpy(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost));
        // primary rays: valid pixels of the owned tiles x spp
        uint64_t prim = 0;
        for (uint32_t k = 0; k < owned; ++k) {
            uint32_t tile = (uint32_t)p->tile_rank + k * (uint32_t)p->tile_world;
            uint32_t tx = tile % A.tiles_x, ty = tile / A.tiles_x;
            int w = std::min(RT_TILE, p->x1 - p->x0 - (int)tx * RT_TILE), hh = std::min(RT_TILE, p->y1 - p->y0 - (int)ty * RT_TILE);
            if (w > 0 && hh > 0) prim += (uint64_t)w * hh;
        }
        stats->rays_primary = prim * (uint64_t)A.spp;
        stats->rays_shadow = h[ST_SHADOW]; stats->rays_ao = h[ST_AO]; stats->rays_bounce = h[ST_BOUNCE];
        stats->rays_skipped = h[ST_SKIPPED];
        if (getenv("RT_BVH_DIAG") && h[ST_DIAG + 0] + h[ST_DIAG + 2] > 0)
            std::fprintf(stderr, "[k_bvh<AO>] node steps %llu (%.1f lanes each), triangle steps %llu (%.1f lanes each), votes %llu (%.1f active lanes), refills %llu (%.1f rays each)\n",
                         h[ST_DIAG], (double)h[ST_DIAG + 1] / std::max(1ull, h[ST_DIAG]), h[ST_DIAG + 2], (double)h[ST_DIAG + 3] / std::max(1ull, h[ST_DIAG + 2]),
                         h[ST_DIAG + 4], (double)h[ST_DIAG + 5] / std::max(1ull, h[ST_DIAG + 4]), h[ST_DIAG + 6], (double)h[ST_DIAG + 7] / std::max(1ull, h[ST_DIAG + 6]));
        if (getenv("RT_BVH_DIAG") && h[ST_DIAG + 0] + h[ST_DIAG + 2] > 0)
            std::fprintf(stderr, "[k_bvh<AO>] per node step: %.1f lanes without a ray, %.1f with parked triangles only, %.1f waiting for room to park a leaf; per triangle step: %.1f lanes could take a node step\n",
                         (double)h[ST_DIAG2] / std::max(1ull, h[ST_DIAG]), (double)h[ST_DIAG2 + 1] / std::max(1ull, h[ST_DIAG]), (double)h[ST_DIAG2 + 2] / std::max(1ull, h[ST_DIAG]),
                         (double)h[ST_DIAG2 + 3] / std::max(1ull, h[ST_DIAG + 2]));
        for (int k = 0; k < 3; ++k) { stats->nodes[k] = h[ST_NODES + k]; stats->tris[k] = h[ST_TRIS + k]; stats->gates[k] = h[ST_GATES + k]; stats->bvh_rays[k] = h[ST_BVHRAYS + k]; }
        float ms = 0.0f;
        HIPCHECK(ctx, hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
        stats->ms_total = ms;
        stats->batches = batches;
        for (auto& sp : be.spans) {
            float m = 0.0f;
            HIPCHECK(ctx, hipEventElapsedTime(&m, ctx->ev_pool[sp.e0], ctx->ev_pool[sp.e1]));
            stats->ms_kernel[sp.cls] += m;
            stats->launches[sp.cls]++;
        }
    }
    return RT_OK;
}

int rt_render_device(rt_scene* s, const rt_camera* cam, const rt_params* p, float* d_rgb, uint32_t* d_primid,
                     float* d_t, rt_stats* stats, void* stream)
{
    if (!s || !cam || !p || !d_rgb) return RT_ERR_INVALID;
    rt_ctx* ctx = s->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!s->committed) { ctx->error = "rt_render: scene not committed"; return RT_ERR_STATE; }
    int rc = check_params(ctx, p);
    if (rc) return rc;
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if (!stream) { rc = order_after_null_stream(ctx); if (rc) return rc; }
    return render_locked(s, cam, p, d_rgb, d_primid, d_t, stats, st, p->tile_world > 1, stream == nullptr);
}

int rt_render_device_noalloc(rt_scene* s, const rt_camera* cam, const rt_params* p, float* d_rgb, uint32_t* d_primid,
                             float* d_t, void* stream)
{
    if (!s || !cam || !p || !d_rgb) return RT_ERR_INVALID;
    rt_ctx* ctx = s->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!s->committed) { ctx->error = "rt_render: scene not committed"; return RT_ERR_STATE; }
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (p->flags & (RT_FLAG_STATS | RT_FLAG_TIMING)) {   // statistics are read back and timing creates events: both allocate
        ctx->error = "rt_render_device_noalloc: RT_FLAG_STATS / RT_FLAG_TIMING are not available here (use rt_render_device)";
        return RT_ERR_UNSUPPORTED;
    }
    if (!ctx->scratch_ptr) { ctx->error = "rt_render_device_noalloc: scratch address was not resolved at rt_init"; return RT_ERR_STATE; }
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    // Under stream capture the call only adds kernel and memset nodes to the caller's graph: no event is waited on or
    // recorded (a capturing stream may not wait on events recorded outside the capture, and an event recorded inside
    // it orders nothing afterwards); the stream's own order serialises the call's work, and the caller orders graph
    // replays against other users of this GPU's scratch.
    bool capturing = false;
    if (stream) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        HIPCHECK(ctx, hipStreamIsCapturing(st, &cs));
        capturing = cs != hipStreamCaptureStatusNone;
    }
    if (!stream) { rc = order_after_null_stream(ctx); if (rc) return rc; }
    if (!capturing) {   // the previous user of this GPU's scratch, whatever context it came through
        std::lock_guard<std::mutex> sl(g_scratch_mu);
        if (g_scratch_ev_recorded[ctx->device]) HIPCHECK(ctx, hipStreamWaitEvent(st, g_scratch_ev[ctx->device], 0));
    }
    rc = render_locked(s, cam, p, d_rgb, d_primid, d_t, nullptr, st, p->tile_world > 1, false, ctx->scratch_ptr, (size_t)RT_NOALLOC_SCRATCH_BYTES, capturing);
    if (!capturing) {   // also after a failed call: whatever it queued before failing still uses the array
        std::lock_guard<std::mutex> sl(g_scratch_mu);
        if (hipEventRecord(g_scratch_ev[ctx->device], st) == hipSuccess) g_scratch_ev_recorded[ctx->device] = true;
    }
    if (rc) return rc;
    if (!stream) HIPCHECK(ctx, hipStreamSynchronize(st));
    return RT_OK;
}

int rt_render(rt_scene* s, const rt_camera* cam, const rt_params* p, float* rgb, uint32_t* primid, float* t, rt_stats* stats)
{
    if (!s || !cam || !p) return RT_ERR_INVALID;
    rt_ctx* ctx = s->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!s->committed) { ctx->error = "rt_render: scene not committed"; return RT_ERR_STATE; }
    int rc = check_params(ctx, p);
    if (rc) return rc;
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)p->width * p->height;
    if (npix > ctx->out_pixels) {
        HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_rgb); (void)hipFree(ctx->d_prim); (void)hipFree(ctx->d_t);
        ctx->d_rgb = nullptr; ctx->d_prim = nullptr; ctx->d_t = nullptr; ctx->out_pixels = 0;
        HIPCHECK(ctx, hipMalloc((void**)&ctx->d_rgb, npix * 12));
        HIPCHECK(ctx, hipMalloc((void**)&ctx->d_prim, npix * 4));
        HIPCHECK(ctx, hipMalloc((void**)&ctx->d_t, npix * 4));
        ctx->out_pixels = npix;
    }
    const bool rect = p->x0 != 0 || p->y0 != 0 || p->x1 != p->width || p->y1 != p->height;
    const bool split = p->tile_world > 1;
    if (split) {   // keep what the caller has in the pixels other ranks own
        if (rgb) HIPCHECK(ctx, hipMemcpyAsync(ctx->d_rgb, rgb, npix * 12, hipMemcpyHostToDevice, ctx->stream));
        if (primid) HIPCHECK(ctx, hipMemcpyAsync(ctx->d_prim, primid, npix * 4, hipMemcpyHostToDevice, ctx->stream));
        if (t) HIPCHECK(ctx, hipMemcpyAsync(ctx->d_t, t, npix * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = render_locked(s, cam, p, ctx->d_rgb, primid ? ctx->d_prim : nullptr, t ? ctx->d_t : nullptr, stats, ctx->stream, false, true);
    if (rc) return rc;
    if (rect && !split) {
        // raytrace(beginY,endY,beginX,endX): only the rectangle travels back, pixels outside it stay the caller's
        const size_t w = (size_t)(p->x1 - p->x0), h = (size_t)(p->y1 - p->y0), W = (size_t)p->width;
        const size_t first = (size_t)p->y0 * W + (size_t)p->x0;
        if (w && h) {
            if (rgb) HIPCHECK(ctx, hipMemcpy2D(rgb + first * 3, W * 12, ctx->d_rgb + first * 3, W * 12, w * 12, h, hipMemcpyDeviceToHost));
            if (primid) HIPCHECK(ctx, hipMemcpy2D(primid + first, W * 4, ctx->d_prim + first, W * 4, w * 4, h, hipMemcpyDeviceToHost));
            if (t) HIPCHECK(ctx, hipMemcpy2D(t + first, W * 4, ctx->d_t + first, W * 4, w * 4, h, hipMemcpyDeviceToHost));
        }
        return RT_OK;
    }
    if (rgb) HIPCHECK(ctx, hipMemcpy(rgb, ctx->d_rgb, npix * 12, hipMemcpyDeviceToHost));
    if (primid) HIPCHECK(ctx, hipMemcpy(primid, ctx->d_prim, npix * 4, hipMemcpyDeviceToHost));
    if (t) HIPCHECK(ctx, hipMemcpy(t, ctx->d_t, npix * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

// grows one of a context's cached device buffers (the context's device is current, its mutex held)
static int ensure_floats(rt_ctx* ctx, float** buf, size_t* have, size_t need)
{
    if (need <= *have) return RT_OK;
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (*buf) HIPCHECK(ctx, hipFree(*buf));
    *buf = nullptr; *have = 0;
    HIPCHECK(ctx, hipMalloc((void**)buf, need * sizeof(float)));
    *have = need;
    return RT_OK;
}

int rt_render_multi(rt_scene* const* scenes, int num_scenes, const rt_camera* cam, const rt_params* p, float* rgb, rt_stats* stats)
{
    if (!scenes || num_scenes < 1 || !cam || !p || !rgb) return RT_ERR_INVALID;
    for (int i = 0; i < num_scenes; ++i) if (!scenes[i]) return RT_ERR_INVALID;
    rt_ctx* ctx0 = scenes[0]->ctx;
    if (num_scenes == 1) {
        rt_params q = *p;
        q.tile_rank = 0; q.tile_world = 1;
        return rt_render(scenes[0], cam, &q, rgb, nullptr, nullptr, stats);
    }
    for (int i = 0; i < num_scenes; ++i)
        for (int j = 0; j < i; ++j)
            if (scenes[i]->ctx == scenes[j]->ctx) { ctx0->error = "rt_render_multi: every scene needs a context of its own"; return RT_ERR_INVALID; }
    rt_params base = *p;
    base.tile_rank = 0; base.tile_world = num_scenes;
    {
        std::lock_guard<std::mutex> lk(ctx0->mu);
        int rc = check_params(ctx0, &base);
        if (rc) return rc;
    }
    // slab layout of every rank: [owned tiles][RT_TILE*RT_TILE][3]; rank 0 owns the most tiles
    uint32_t max_owned = 0;
    rt_owned_tiles(&base, 0, num_scenes, &max_owned);
    const size_t slab_floats = (size_t)max_owned * RT_TILE * RT_TILE * 3;
    const size_t npix = (size_t)p->width * p->height;
    {
        std::lock_guard<std::mutex> lk(ctx0->mu);
        HIPCHECK(ctx0, hipSetDevice(ctx0->device));
        int rc = ensure_floats(ctx0, &ctx0->d_gather, &ctx0->gather_floats, slab_floats * (size_t)num_scenes);
        if (rc) return rc;
        if (npix > ctx0->out_pixels) {
            HIPCHECK(ctx0, hipStreamSynchronize(ctx0->stream));
            (void)hipFree(ctx0->d_rgb); (void)hipFree(ctx0->d_prim); (void)hipFree(ctx0->d_t);
            ctx0->d_rgb = nullptr; ctx0->d_prim = nullptr; ctx0->d_t = nullptr; ctx0->out_pixels = 0;
            HIPCHECK(ctx0, hipMalloc((void**)&ctx0->d_rgb, npix * 12));
            HIPCHECK(ctx0, hipMalloc((void**)&ctx0->d_prim, npix * 4));
            HIPCHECK(ctx0, hipMalloc((void**)&ctx0->d_t, npix * 4));
            ctx0->out_pixels = npix;
        }
    }
    // one host thread + the context's own stream per rank: render the rank's tiles, send the slab to the gathering GPU
    std::vector<int> rcs((size_t)num_scenes, RT_OK);
    auto rank_work = [&](int i) -> int {
        rt_scene* s = scenes[i];
        rt_ctx* ctx = s->ctx;
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (!s->committed) { ctx->error = "rt_render_multi: scene not committed"; return RT_ERR_STATE; }
        HIPCHECK(ctx, hipSetDevice(ctx->device));
        uint32_t owned = 0;
        rt_owned_tiles(&base, i, num_scenes, &owned);
        if (owned == 0) { if (stats) std::memset(&stats[i], 0, sizeof(rt_stats)); return RT_OK; }
        int rc = ensure_floats(ctx, &ctx->d_slab, &ctx->slab_floats, slab_floats);
        if (rc) return rc;
        rt_params q = base;
        q.tile_rank = i;
        rc = render_locked(s, cam, &q, ctx->d_slab, nullptr, nullptr, stats ? &stats[i] : nullptr, ctx->stream, true, false);
        if (rc) return rc;
        const size_t bytes = (size_t)owned * RT_TILE * RT_TILE * 3 * sizeof(float);
        float* dst = ctx0->d_gather + slab_floats * (size_t)i;
        if (ctx->device == ctx0->device) HIPCHECK(ctx, hipMemcpyAsync(dst, ctx->d_slab, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        else {
            // direct access to the gathering GPU's memory, enabled once per pair: without it the peer copy is staged
            // through the host instead of crossing this GPU's own xGMI link to GPU 0
            if (std::find(ctx->peers_enabled.begin(), ctx->peers_enabled.end(), ctx0->device) == ctx->peers_enabled.end()) {
                int can = 0;
                HIPCHECK(ctx, hipDeviceCanAccessPeer(&can, ctx->device, ctx0->device));
                if (can) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(ctx0->device, 0);
                    if (pe == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                    else if (pe != hipSuccess) { ctx->error = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe); return RT_ERR_HIP; }
                } else {
                    std::fprintf(stderr, "rt_render_multi: GPU %d cannot access GPU %d directly (hipDeviceCanAccessPeer = 0): "
                                         "its tile slab is staged through the host\n", ctx->device, ctx0->device);
                }
                ctx->peers_enabled.push_back(ctx0->device);
            }
            HIPCHECK(ctx, hipMemcpyPeerAsync(dst, ctx0->device, ctx->d_slab, ctx->device, bytes, ctx->stream));   // one xGMI link per peer, all ranks at once
        }
        HIPCHECK(ctx, hipEventRecord(ctx->ev_slab, ctx->stream));   // the gathering GPU unpacks this slab as soon as it has landed
        return RT_OK;
    };
    {
        std::vector<std::thread> threads;
        for (int i = 1; i < num_scenes; ++i) threads.emplace_back([&, i] { rcs[(size_t)i] = rank_work(i); });
        rcs[0] = rank_work(0);
        for (auto& t : threads) t.join();
    }
    for (int i = 0; i < num_scenes; ++i)
        if (rcs[(size_t)i] != RT_OK) {
            if (i != 0) ctx0->error = "rt_render_multi: rank " + std::to_string(i) + ": " + scenes[i]->ctx->error;
            return rcs[(size_t)i];
        }
    // scatter the slabs into the frame on the gathering GPU and hand the frame to the caller
    std::lock_guard<std::mutex> lk(ctx0->mu);
    HIPCHECK(ctx0, hipSetDevice(ctx0->device));
    const bool rect = p->x0 != 0 || p->y0 != 0 || p->x1 != p->width || p->y1 != p->height;
    uint32_t tx, ty;
    tile_grid(&base, tx, ty);
    for (int i = 0; i < num_scenes; ++i) {
        uint32_t owned = 0;
        rt_owned_tiles(&base, i, num_scenes, &owned);
        if (owned == 0) continue;
        // per-rank order: the unpack of rank i waits (on the device) for rank i's copy only, not for every rank
        if (i != 0) HIPCHECK(ctx0, hipStreamWaitEvent(ctx0->stream, scenes[i]->ctx->ev_slab, 0));
        hipLaunchKernelGGL(k_unpack, dim3(grid_for((uint64_t)owned * RT_TILE * RT_TILE)), dim3(RT_BLOCK), 0, ctx0->stream,
                           ctx0->d_gather + slab_floats * (size_t)i, ctx0->d_rgb, p->width, p->x0, p->y0, p->x1, p->y1, tx, ty, i, num_scenes, owned);
    }
    HIPCHECK(ctx0, hipGetLastError());
    HIPCHECK(ctx0, hipStreamSynchronize(ctx0->stream));
    if (rect) {   // raytrace(beginY,endY,beginX,endX): only the rectangle travels back
        const size_t w = (size_t)(p->x1 - p->x0), h = (size_t)(p->y1 - p->y0), W = (size_t)p->width;
        const size_t first = (size_t)p->y0 * W + (size_t)p->x0;
        if (w && h) HIPCHECK(ctx0, hipMemcpy2D(rgb + first * 3, W * 12, ctx0->d_rgb + first * 3, W * 12, w * 12, h, hipMemcpyDeviceToHost));
        return RT_OK;
    }
    HIPCHECK(ctx0, hipMemcpy(rgb, ctx0->d_rgb, npix * 12, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_unpack_tiles(rt_ctx* ctx, const float* d_slab, const rt_params* p, int rank, int world, float* d_frame, void* stream)
{
    if (!ctx || !d_slab || !p || !d_frame) return RT_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    uint32_t owned = 0;
    if (rt_owned_tiles(p, rank, world, &owned) != RT_OK) { ctx->error = "rt_unpack_tiles: bad rank/world"; return RT_ERR_INVALID; }
    if (owned == 0) return RT_OK;
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    uint32_t tx, ty;
    tile_grid(p, tx, ty);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if (!stream) { const int rc = order_after_null_stream(ctx); if (rc) return rc; }
    hipLaunchKernelGGL(k_unpack, dim3(grid_for((uint64_t)owned * RT_TILE * RT_TILE)), dim3(RT_BLOCK), 0, st,
                       d_slab, d_frame, p->width, p->x0, p->y0, p->x1, p->y1, tx, ty, rank, world, owned);
    HIPCHECK(ctx, hipGetLastError());
    if (!stream) HIPCHECK(ctx, hipStreamSynchronize(st));
    return RT_OK;
}

int rt_quantize_rgb8(rt_ctx* ctx, const float* d_rgb, size_t num_pixels, uint8_t* d_rgb8, void* stream)
{
    if (!ctx || !d_rgb || !d_rgb8) return RT_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if (!stream) { const int rc = order_after_null_stream(ctx); if (rc) return rc; }
    if (num_pixels) hipLaunchKernelGGL(k_quantize, dim3(grid_for(num_pixels * 3)), dim3(RT_BLOCK), 0, st, d_rgb, d_rgb8, num_pixels * 3);
    HIPCHECK(ctx, hipGetLastError());
    if (!stream) HIPCHECK(ctx, hipStreamSynchronize(st));
    return RT_OK;
}

int rt_selftest_device_math(rt_ctx* ctx, uint64_t n, uint64_t seed, uint64_t mismatches[4])
{
    if (!ctx || !mismatches) return RT_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    static_assert(ST_TOTAL >= 4, "the statistics array doubles as the self-test's counters");
    HIPCHECK(ctx, hipMemsetAsync(ctx->d_stats, 0, sizeof(unsigned long long) * 4, ctx->stream));
    if (n) hipLaunchKernelGGL(k_selftest_math, dim3(grid_for(n > (1ull << 24) ? (1ull << 24) : n)), dim3(RT_BLOCK), 0, ctx->stream, (unsigned long long)n, (unsigned long long)seed, ctx->d_stats);
    HIPCHECK(ctx, hipGetLastError());
    unsigned long long h[4];
    HIPCHECK(ctx, hipMemcpyAsync(h, ctx->d_stats, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 4; ++k) mismatches[k] = h[k];
    return RT_OK;
}

size_t rt_bmp24_bytes(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    const size_t row = ((size_t)width * 3 + 3) & ~(size_t)3;
    return 54 + row * (size_t)height;
}

int rt_encode_bmp24(rt_ctx* ctx, const float* d_rgb, int width, int height, uint8_t* d_file, void* stream)
{
    if (!ctx) return RT_ERR_INVALID;
    if (!d_rgb || !d_file || width <= 0 || height <= 0) { ctx->error = "rt_encode_bmp24: null buffer or empty image"; return RT_ERR_INVALID; }
    const size_t total = rt_bmp24_bytes(width, height);
    if (total > 0xFFFFFFFFull) { ctx->error = "rt_encode_bmp24: image too large for a BMP"; return RT_ERR_UNSUPPORTED; }
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIPCHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t row = (uint32_t)(((size_t)width * 3 + 3) & ~(size_t)3);
    BmpHeader h{};
    auto put32 = [&](int at, uint32_t v) { for (int k = 0; k < 4; ++k) h.b[at + k] = (uint8_t)(v >> (8 * k)); };
    h.b[0] = 'B'; h.b[1] = 'M';
    put32(2, (uint32_t)total); put32(10, 54);
    put32(14, 40); put32(18, (uint32_t)width); put32(22, (uint32_t)height);
    h.b[26] = 1; h.b[28] = 24;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if (!stream) { const int rc = order_after_null_stream(ctx); if (rc) return rc; }
    hipLaunchKernelGGL(k_bmp24, dim3(grid_for((size_t)width * height)), dim3(RT_BLOCK), 0, st, d_rgb, d_file, width, height, row, h);
    HIPCHECK(ctx, hipGetLastError());
    if (!stream) HIPCHECK(ctx, hipStreamSynchronize(st));
    return RT_OK;
}

} // extern "C"
