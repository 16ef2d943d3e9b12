// rt_bvh.hip — k_bvh, the persistent triangle-BVH traversal kernel (the dominant kernel of the pipeline), in a
// translation unit of its own: it is built with -mllvm -amdgpu-sched-strategy=iterative-ilp (see the Makefile),
// which shortens this VALU-issue-bound kernel by 5 % (k_bvh<AO> 121.1 -> 115.1 ms on config 3) and lengthens the
// gate-walking kernels of rt_gpu.hip by 1-2 %, so those keep the default strategy.
#include "rt_kernels.hpp"

using namespace rt;

// Compacted triangle-BVH traversal (the dominant kernel).  Persistent waves with three ideas:
//  * refill: a lane that finishes its ray fetches the next one from the queue (the wave takes
//    RT_BVH_CHUNK consecutive items with one atomic; a single hot counter word serves only ~90
//    returning atomics per microsecond: MI355X_MICROARCH.md row 'dequeue'), so short rays do not
//    idle their lanes until the longest ray of the wave is done;
//  * decoupled node / triangle work: leaves met during the descent are parked in a small per-lane
//    LDS queue instead of being tested on the spot, and every iteration the wave votes (ballot +
//    popcount) for ONE kind of step — a 64-byte four-wide node fetch + four box tests, or one 48-byte
//    triangle fetch + exact test — whichever more lanes are ready for;
//  * the exact Moller-Trumbore test is evaluated branch-free (same operations, same order as
//    tri_t(); only the early returns are folded into one predicate).
// Results are identical to bvh_item()/bvh_drain() (rt_pipeline.hpp / rt_core.hpp), which the CPU
// emulation runs: the closest hit does not depend on visiting order (ties resolved by rank).
#ifndef RT_BVH_CHUNK
#define RT_BVH_CHUNK 512
#endif
#ifndef RT_REFILL_IDLE
#define RT_REFILL_IDLE 20    // refill once this many lanes hold no ray (40 until the vote below was re-weighted)
#endif
#ifndef RT_PEND
#define RT_PEND 1            // parked leaves per lane: 1 = in a register (round 3; 2 in LDS until the vote below was re-weighted: 92.0 -> 89.9 ms)
#endif
#ifndef RT_NODE_REPS
#define RT_NODE_REPS 3       // steps of one kind per vote (see the vote below)
#endif
#ifndef RT_TRI_REPS
#define RT_TRI_REPS 6
#endif
#ifndef RT_VOTE_NUM
// A node step wins the vote when n_node * RT_VOTE_DEN > n_tri * RT_VOTE_NUM.  Round 3: 3 : 1 instead of 1 : 1 — a triangle step
// costs half a node step and is what frees lanes (a ray whose stack is empty only waits for its parked triangles: 9.6 lanes per
// node step, RT_BVH_DIAG), so it pays to take it as soon as a third as many lanes want it; with that, refilling at 20 idle
// lanes instead of 40 and 3 repeated node steps instead of 4 became the best of their sweeps: k_bvh<AO> 97.7 -> 92.1 ms on
// config 3 (2 : 1 ... 3 : 1, 16 ... 24 idle lanes, 2 ... 4 steps all within 0.5 ms), 21.5 -> 21.05 ms on config 5.
#define RT_VOTE_NUM 3
#endif
#ifndef RT_VOTE_DEN
#define RT_VOTE_DEN 1
#endif
#ifndef RT_REP_NUM
#define RT_REP_NUM 1         // repeat while RT_REP_NUM/RT_REP_DEN of the voters are ready again
#define RT_REP_DEN 4
#endif

// MODE: BVH_CLOSEST (hit records), BVH_AO (distances only), BVH_SHADOW (any hit before the light ends the ray).
// Register budget for 6 waves per SIMD (80 VGPRs; the timed variants fit without scratch, the counting variants spill
// a little).  This kernel's time follows the waves that fit: with LDS padding that leaves 5 / 4 / 3 / 2 waves per SIMD
// k_bvh<AO> takes 103 / 117 / 143 / 199 ms on config 3 (T ~ 32 + 334 / waves: two thirds of it is memory latency that
// only more resident waves hide), see DESIGN.md 4.
// Scenes whose triangles and nodes exceed the Infinity Cache (config 5: 655 MB) are the exception: there the sixth wave
// costs more in cache misses than it hides (k_bvh<AO> 21.2 ms at 5 waves, 22.4 at 6), so both variants are built and
// launch_k_bvh picks by DScene::big_scene.
#ifndef RT_BVH_WAVES
#define RT_BVH_WAVES 6
#endif
#define RT_BVH_ATTR(W) __attribute__((amdgpu_waves_per_eu(W, W)))
template <int MODE, bool STATS, int WAVES>
__global__ void __launch_bounds__(RT_BVH_BLOCK) RT_BVH_ATTR(WAVES) k_bvh(DScene S, Batch B, int counter, int next_counter)
{
    constexpr bool AO = MODE != BVH_CLOSEST;   // the queue layout of the distance-only queries: no bq_c, bq_cap slots
    // Per-lane stack of {reference, entry distance of its box}, [entry][lane] (conflict-free).  The distance lets
    // a pop discard an entry whose box has fallen behind the closest hit found since it was pushed without
    // fetching the node (18 % of the node fetches of an AO ray otherwise).
    __shared__ uint2 s_stack[RT_LDS_STACK * RT_BVH_BLOCK];
#if RT_PEND > 1
    __shared__ uint32_t s_pend[RT_PEND * RT_BVH_BLOCK];
#endif
#ifdef RT_BVH_LDS_PAD   // experiment: fewer resident waves (round 3: is the kernel bound by issue or by latency? see DESIGN.md 4)
    __shared__ uint32_t s_pad[RT_BVH_LDS_PAD / 4];
    if (counter == 0x7FFFFFFF) s_pad[threadIdx.x] = 1u;
#endif
    // volatile: keeps the (rare) global spill accesses from being merged with the LDS ones into
    // generic-pointer (flat) instructions, which wait on both memory counters
    // (formed from the kernel argument + a 32-bit lane offset at each use: a 64-bit pointer held per lane would cost two
    // registers for the life of the kernel, and this kernel's time falls with every wave that fits — DESIGN.md 4)
    #define spill ((volatile uint2*)B.bq_spill + ((size_t)blockIdx.x * RT_BVH_BLOCK + threadIdx.x))
    const size_t spill_stride = (size_t)RT_BVH_GRID * RT_BVH_BLOCK;
    auto push = [&](uint32_t& sp_, uint32_t x, float tn) {
        if (sp_ < RT_LDS_STACK) s_stack[sp_ * RT_BVH_BLOCK + threadIdx.x] = make_uint2(x, f2u(tn));
        else if (sp_ < RT_STACK_DEPTH) { spill[(size_t)(sp_ - RT_LDS_STACK) * spill_stride].x = x; spill[(size_t)(sp_ - RT_LDS_STACK) * spill_stride].y = f2u(tn); }
        else return;
        ++sp_;
    };
    auto pop = [&](uint32_t& sp_) -> uint2 {
        --sp_;
        uint2 x = s_stack[(sp_ < RT_LDS_STACK ? sp_ : RT_LDS_STACK - 1) * RT_BVH_BLOCK + threadIdx.x];
        if (sp_ >= RT_LDS_STACK) { x.x = spill[(size_t)(sp_ - RT_LDS_STACK) * spill_stride].x; x.y = spill[(size_t)(sp_ - RT_LDS_STACK) * spill_stride].y; }
        return x;
    };
    uint32_t n = __builtin_amdgcn_readfirstlane(B.counters[counter]);   // wave-uniform values belong in scalar registers: as plain loads / shuffles the queue bookkeeping below ran on the vector unit
    if (n > (AO ? B.bq_cap : B.cq_cap)) n = AO ? B.bq_cap : B.cq_cap;
    // a wave takes at least 64 items, so waves beyond ceil(n/64) have nothing to fetch: they leave before touching
    // the queue counter (the launches of the deeper levels of a small frame hold a few hundred rays for 8192 waves)
    if ((uint64_t)(blockIdx.x * (RT_BVH_BLOCK / 64) + (threadIdx.x >> 6)) * 64u >= n) return;
    uint32_t* next = &B.counters[next_counter];
    Counters cnt{0, 0, 0, 0};
    const uint32_t lane = threadIdx.x & 63u;
    f3 o = F3(0, 0, 0), d = F3(0, 0, 1);
    RayCull4 rc = {};
    SelFar sf = {0u, 0u, 0u};   // the exit planes' byte selectors, held (74 VGPRs leave room at 6 waves per SIMD): three xors fewer per node step
    Hit best = {};
    uint32_t sp = 0, cur = REF_NONE, dst = 0;
    // the parked leaves of the lane
#if RT_PEND > 1
    uint32_t npend = 0;
    uint32_t* pend = s_pend + threadIdx.x;
    #define PEND_ANY() (npend > 0)
    #define PEND_ROOM() (npend < RT_PEND)
    #define PEND_PUSH(x) (pend[(npend++) * RT_BVH_BLOCK] = (x))
    #define PEND_POP() (pend[(--npend) * RT_BVH_BLOCK])
    #define PEND_CLEAR() (npend = 0)
#else
    uint32_t pend0 = REF_NONE;
    #define PEND_ANY() (pend0 != REF_NONE)
    #define PEND_ROOM() (pend0 == REF_NONE)
    #define PEND_PUSH(x) (pend0 = (x))
    #define PEND_POP() (pend0)
    #define PEND_CLEAR() (pend0 = REF_NONE)
#endif
    uint32_t tri = 0;   // triangles of the leaf being tested: next slot (bits 0..27) | triangles left (bits 28..31)
    bool active = false, exhausted = false;
    uint32_t chunk_next = 0, chunk_end = 0;   // wave-uniform
    uint32_t nrays = 0;
    uint32_t dg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // STATS: wave-uniform diagnostics (see ST_DIAG)
    for (;;) {
        // ---- retire finished rays -------------------------------------------------------------
        if (active && cur == REF_NONE && !PEND_ANY() && (tri >> 28) == 0u) {
            if (MODE == BVH_AO) key_min(&B.ao_key[dst], t_key(best.t));   // the ray's other items and k_ao's own hit meet here
            else if (MODE == BVH_SHADOW) { if (best.prim != H_MISS) B.sh_res[dst] = 0.0f; }   // k_shadow_defer stored what holds otherwise
            else { const q4 r = {best.t, u2f(best.prim), best.u, best.v}; B.hit[dst] = r; }
            active = false;
        }
        // ---- refill ---------------------------------------------------------------------------
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!active);
        const uint32_t nidle = (uint32_t)__popcll(idle);
        if (!exhausted && (nidle >= RT_REFILL_IDLE)) {
            if (chunk_next >= chunk_end) {
                // guided self-scheduling: full chunks while plenty of items remain, smaller ones near the
                // end of the queue so the launch does not end with a few waves holding hundreds of rays
                const uint32_t rem = chunk_end < n ? n - chunk_end : 0u;
                uint32_t take = (rem >> 14) << 6;
                take = take < 64u ? 64u : (take > RT_BVH_CHUNK ? (uint32_t)RT_BVH_CHUNK : take);
                if (chunk_end == 0) take = RT_BVH_CHUNK;   // first grab: position unknown
                uint32_t b0 = 0;
                if (lane == 0) b0 = atomicAdd(next, take);
                b0 = __builtin_amdgcn_readfirstlane(b0);   // every lane of the wave is here (the conditions above are wave-uniform)
                chunk_next = b0; chunk_end = b0 + take < n ? b0 + take : n;
                if (b0 >= n) { exhausted = true; chunk_next = chunk_end = 0; }
            }
            const uint32_t base = chunk_next, lim = chunk_end;
            chunk_next = base + nidle < lim ? base + nidle : lim;
            if (STATS && MODE == BVH_AO) { dg[6]++; dg[7] += chunk_next - base; }
            if (!active) {
                const uint32_t idx = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (idx < lim) {   // items at or beyond the chunk end are left for the next refill
                    const q4 a = B.bq_a[idx], b = B.bq_b[idx];
                    const uint32_t root0 = B.bq_roots[idx];   // REF_NONE marks a hole
                    o = xyz(a); d = xyz(b);
                    const bool multi = (f2u(b.w) & RT_BQ_MULTI) != 0u;
                    dst = f2u(b.w) & RT_BQ_DST_MASK;
                    best.t = a.w; best.prim = H_MISS; best.rank = 0; best.u = 0; best.v = 0;
                    if (!AO) { const q4 c = B.bq_c[idx]; best.prim = f2u(c.x); best.rank = f2u(c.y); best.u = c.z; best.v = c.w; }
                    rc = make_cull4(S, o, d);
                    sf = SelFar{f2u(rc.spx.y) ^ RT_PERM_FLIP, f2u(rc.spy.y) ^ RT_PERM_FLIP, f2u(rc.spz.y) ^ RT_PERM_FLIP};
                    sp = 0; PEND_CLEAR(); tri = 0;
                    if (root0 != REF_NONE) {
                        push(sp, root0, -INFINITY);
                        if (multi)   // rare: one Mesh is the rule; a list shorter than RT_DEFER_ROOTS ends with REF_NONE
                            for (uint32_t k = 1; k < RT_DEFER_ROOTS; ++k) {
                                const uint32_t rk = B.bq_roots[(size_t)k * B.bq_cap + idx];
                                if (rk == REF_NONE) break;
                                push(sp, rk, -INFINITY);
                            }
                    }
                    if (sp > 0) { cur = pop(sp).x; active = true; if (STATS) nrays++; }   // a slot without roots is a hole
                }
            }
        }
        // ---- vote for the kind of step ------------------------------------------------------------
        const bool can_node = active && cur != REF_NONE && (!(cur & REF_LEAF) || PEND_ROOM());
        const bool can_tri = active && ((tri >> 28) != 0u || PEND_ANY());
        const uint32_t n_node = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(can_node));
        const uint32_t n_tri = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(can_tri));
        if (n_node == 0 && n_tri == 0) {
            if (__builtin_amdgcn_ballot_w64(active) == 0ull && (exhausted || n == 0)) break;
            continue;   // only finished rays (retired at the top) or an empty refill: go round again
        }
        // The winner of the vote is repeated (up to RT_*_REPS steps) while RT_REP_NUM/RT_REP_DEN of its voters
        // are ready again: the retire / refill / vote code above is a third of an iteration.
        if (STATS && MODE == BVH_AO) { dg[4]++; dg[5] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(active)); }
        if (n_node * RT_VOTE_DEN > n_tri * RT_VOTE_NUM) {
            bool cn = can_node;
            for (int rep = 0;; ++rep) {
                if (STATS && MODE == BVH_AO) {
                    dg[0]++; dg[1] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(cn));
                    dg[8] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(!active));
                    dg[9] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(active && cur == REF_NONE));
                    dg[10] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(active && cur != REF_NONE && !cn));
                }
                if (cn) {
                    if (cur & REF_LEAF) {
                        PEND_PUSH(cur);   // a leaf that arrived as a root or waited for room (can_node: there is room now)
                    } else {
                        float t4[4];
                        uint32_t r4[4];
                        if (STATS) cnt.nodes++;
                        // one 64-byte fetch, four box tests, children sorted farthest first (missed ones last)
                        const uint32_t nh = node_children_desc(S, rc, cur, best.t, t4, r4, sf);
                        if (sp + 4u <= RT_LDS_STACK) {   // all four pairs at fixed offsets, no predication: what lies beyond sp + nh is never read
                            uint2* top = s_stack + sp * RT_BVH_BLOCK + threadIdx.x;
                            top[0 * RT_BVH_BLOCK] = make_uint2(r4[0], f2u(t4[0]));
                            top[1 * RT_BVH_BLOCK] = make_uint2(r4[1], f2u(t4[1]));
                            top[2 * RT_BVH_BLOCK] = make_uint2(r4[2], f2u(t4[2]));
                            top[3 * RT_BVH_BLOCK] = make_uint2(r4[3], f2u(t4[3]));
                            sp += nh;
                        } else {
                            for (uint32_t k = 0; k < 4; ++k) if (k < nh) push(sp, r4[k], t4[k]);
                        }
                    }
                    // next reference: the nearest child just stored, or what waited below it; entries whose box lies
                    // behind the closest hit by now are dropped without a fetch, leaves are parked for the triangle steps
                    uint32_t x = REF_NONE;
                    while (x == REF_NONE && sp > 0) {
                        const uint2 y = pop(sp);
                        if (u2f(y.y) > best.t) continue;
                        if ((y.x & REF_LEAF) && PEND_ROOM()) PEND_PUSH(y.x);
                        else x = y.x;
                    }
                    cur = x;
                }
                if (rep == RT_NODE_REPS - 1) break;
                cn = active && cur != REF_NONE && (!(cur & REF_LEAF) || PEND_ROOM());
                if ((uint32_t)__popcll(__builtin_amdgcn_ballot_w64(cn)) * RT_REP_DEN < n_node * RT_REP_NUM) break;
            }
        } else {
            bool ct = can_tri;
            for (int rep = 0;; ++rep) {
                if (STATS && MODE == BVH_AO) {
                    dg[2]++; dg[3] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(ct));
                    dg[11] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(active && cur != REF_NONE && (!(cur & REF_LEAF) || PEND_ROOM())));
                }
                if (ct) {
                    if ((tri >> 28) == 0u) {
                        const uint32_t ref = PEND_POP();
#if RT_PEND == 1
                        pend0 = REF_NONE;
#endif
                        tri = (ref & 0x7FFFFFFFu) + 0x10000000u;   // leaf reference: count-1 in bits 28..30 -> count in bits 28..31
                    }
                    const uint32_t tri_next = tri & 0x0FFFFFFFu;
                    const q4* T = S.tris + (size_t)tri_next * 3;
                    const q4 a = T[0], b = T[1], e = T[2];
                    if (STATS) cnt.tris++;
                    // Triangle::intersectRay (reference main.cpp:713-739), branch-free
                    const f3 v0v1 = xyz(b), v0v2 = xyz(e);
                    const f3 pvec = cross(d, v0v2);
                    const float det = dot(v0v1, pvec);
                    const float invDet = 1.0f / det;
                    const f3 tvec = o - xyz(a);
                    const float u = dot(tvec, pvec) * invDet;
                    const f3 qvec = cross(tvec, v0v1);
                    const float v = dot(d, qvec) * invDet;
                    const float t = dot(v0v2, qvec) * invDet;
                    const bool ok = !(det < 0.000001f) && !(u < 0 || u > 1) && !(v < 0 || u + v > 1);
                    const uint32_t rank = f2u(e.w);
                    if (MODE == BVH_SHADOW) {   // all-opaque meshes: the first triangle before the light decides, the ray is done
                        if (ok && t < best.t) { best.prim = H_TRI; sp = 0; PEND_CLEAR(); tri = (tri & 0x0FFFFFFFu) | 0x10000000u; cur = REF_NONE; }
                    } else if (MODE == BVH_AO) {   // only the distance leaves this kernel: the reference's first-wins rule on equal t cannot change it
                        if (ok && t < best.t) best.t = t;
                    } else if (ok && closer(t, rank, best)) { best.t = t; best.prim = H_TRI | tri_next; best.u = u; best.v = v; best.rank = rank; }
                    tri += 0xF0000001u;   // next slot, one triangle fewer
                }
                if (rep == RT_TRI_REPS - 1) break;
                ct = active && ((tri >> 28) != 0u || PEND_ANY());
                if ((uint32_t)__popcll(__builtin_amdgcn_ballot_w64(ct)) * RT_REP_DEN < n_tri * RT_REP_NUM) break;
            }
        }
    }
    #undef spill
    #undef PEND_ANY
    #undef PEND_ROOM
    #undef PEND_PUSH
    #undef PEND_POP
    #undef PEND_CLEAR
    constexpr int cls = MODE == BVH_AO ? 2 : MODE == BVH_SHADOW ? 1 : 0;
    flush_counters<STATS>(cnt, B.stats, cls);
    if (STATS) {
        for (int off = 32; off > 0; off >>= 1) nrays += __shfl_xor(nrays, off);
        if (lane == 0) atomicAdd(&B.stats[ST_BVHRAYS + cls], (unsigned long long)nrays);
        if (MODE == BVH_AO && lane == 0) for (int k = 0; k < 8; ++k) atomicAdd(&B.stats[ST_DIAG + k], (unsigned long long)dg[k]);
        if (MODE == BVH_AO && lane == 0) for (int k = 0; k < 4; ++k) atomicAdd(&B.stats[ST_DIAG2 + k], (unsigned long long)dg[8 + k]);
    }
}


namespace rt {
void launch_k_bvh(int mode, bool stats, hipStream_t stream, const DScene& S, const Batch& B, int counter, int next_counter)
{
    const dim3 grid(RT_BVH_GRID), block(RT_BVH_BLOCK);
#define RT_BVH_LAUNCH(M, ST, W) hipLaunchKernelGGL((k_bvh<M, ST, W>), grid, block, 0, stream, S, B, counter, next_counter)
#define RT_BVH_PICK(M) do { \
        if (S.big_scene) { if (stats) RT_BVH_LAUNCH(M, true, 5); else RT_BVH_LAUNCH(M, false, 5); } \
        else { if (stats) RT_BVH_LAUNCH(M, true, RT_BVH_WAVES); else RT_BVH_LAUNCH(M, false, RT_BVH_WAVES); } \
    } while (0)
    if (mode == BVH_AO) RT_BVH_PICK(BVH_AO);
    else if (mode == BVH_SHADOW) RT_BVH_PICK(BVH_SHADOW);
    else RT_BVH_PICK(BVH_CLOSEST);
#undef RT_BVH_PICK
#undef RT_BVH_LAUNCH
}
} // namespace rt
