// rt_pipeline.hpp — the wavefront pipeline: per-item bodies of the seven kernels and the
// batch/depth loop that sequences them.  The item bodies are host/device functions (see
// rt_core.hpp for why); `render_frame<Backend>` is instantiated with the HIP launcher in
// rt_gpu.hip and with a sequential loop backend in the test-only emulation (tests/emul/).
//
// Pipeline per batch of P = pixels*spp paths (reference call stack in brackets):
//   raygen            [raytrace(): sampleUnitSquare + Camera::calculateRay, main.cpp:1633-1641]
//   for depth d = 0 .. max_depth+1 over the per-depth path queue:
//     trace           [for obj in world->scene: obj->hit(), main.cpp:1645-1648 / 1457-1460 / 1548-1551]
//     bvh (trace)     [the Mesh::hit triangle loops of the rays whose gates passed, main.cpp:890-912, compacted]
//     trace_overflow  [rays with more Meshes than a queue item carries / a full queue, traversed in place]
//     prep            [hit attributes; Reflective::rho / Transparency::rho ray set-up, main.cpp:1437-1468, 1525-1592]
//     shadow          [Material::shadowed, main.cpp:40-76]
//     ao              [Ambient::L's n x n rays, main.cpp:1361-1404]
//     bvh (ao)        [their Mesh::hit triangle loops, compacted]
//     ao_overflow     [the AO rays of waves that found the work queue full, traversed in place; normally none]
//     shade           [Matte/Specular/Textured/Mirror/Refraction::shade, main.cpp:1143-1299, 1487-1511, 1607-1616]
//   resolve           [per-level `L = clamp(local + child)` unwound deepest first; pixel average main.cpp:1656-1658]
#pragma once
#include "rt_core.hpp"

namespace rt {

enum : uint8_t { LK_NONE = 0, LK_TERMINAL = 1, LK_MIRROR = 2, LK_REFRACT = 3 };
// (C_AO counts the entries of Batch::ao_ovf)
enum { C_QUEUE = 0, C_AO = 64, C_SH = 128, C_BQ = 192, C_BQA = 256, C_BQ_NEXT = 320, C_BQA_NEXT = 384, C_TR_OVF = 448 /* entries of Batch::ao_ovf written by k_trace */,
       C_BQS = 512, C_BQS_NEXT = 576 /* the work queue as the shadow pass uses it */, C_SH_OVF = 640 /* entries of Batch::ao_ovf written by k_shadow_defer */,
       C_AOW = 704 /* k_ao_walk: next unclaimed AO item */, C_TOTAL = 768 };   // per-batch u32 counters, one per depth
enum { ST_SHADOW = 0, ST_AO = 1, ST_BOUNCE = 2, ST_NODES = 3, ST_TRIS = 6, ST_GATES = 9, ST_BVHRAYS = 12,
       ST_DIAG = 15 /* k_bvh<AO> lane-packing diagnostics of the counting variant: node steps, lanes in them, triangle steps, lanes in them, votes, active lanes at votes, refills, lanes refilled */,
       ST_SKIPPED = 23 /* closest-hit queries the reference traces and this path proves unnecessary (prep_item) */,
       ST_DIAG2 = 24 /* more of the same, summed over node steps: lanes without a ray, lanes whose ray has only parked triangles left, lanes waiting with a leaf for room in the parked list; summed over triangle steps: lanes that could take a node step */,
       ST_TOTAL = 28 }; // u64
#define RT_MAX_LEVELS 62

struct Batch {
    uint32_t P;            // path capacity of the buffers
    q4* ray_o[2];          // {origin, index of refraction of the medium (ShadeRec::indexOfRefraction)}
    q4* ray_d[2];          // {direction, 0}
    q4* hit;               // {t, bits(prim), u, v}  (u,v: barycentric after trace, texture uv after prep)
    q4* nrm;               // {normal, bits(material)}
    uint32_t* pstate;      // bits 0..7: Ambient::L evaluations so far; bits 8..15: index of the current one
    uint32_t* queue[2];    // paths alive at depth d (d >= 1), ping-pong
    // AO work is positional: slot = position of the path in this depth's path list.  Per slot, written by
    // k_prep: {origin, bits(RNG prefix of the primary sample)} {u, 0} {w, bits(AO call index), 0xFFFFFFFF = no AO rays}
    q4* ao_o; q4* ao_u; q4* ao_w;
    uint32_t* ao_slot;     // per path: its slot, 0xFFFFFFFF = none
    uint32_t* sh_queue;    // p * num_lights + light
    uint32_t* ao_key;      // [slot * n*n + k]  closest distance as an ordered key (t_key; occlusion_dist when nothing was hit):
                           // k_ao stores the closest of the primitives it tests in place, k_bvh / k_ao_overflow take minima
    uint32_t* ao_ovf;      // waves of k_ao (first item index) that found the BVH work queue full: redone by k_ao_overflow
    uint32_t ovf_cap;
    uint32_t* ovf_bits;    // GPU, k_ao_walk: one bit per 64 AO items, set when the group was entered in ao_ovf (its lanes hold items of many groups)
    float* sh_res;         // [p * num_lights + light]
    q4* lvl;               // [d*P + p]  TERMINAL: final colour; MIRROR: {ambient term, phong of light 0}
    float* lvl_ph;         // [(d*P + p)*(num_lights-1) + (l-1)] phong of lights 1..
    uint8_t* lvl_kind;     // [d*P + p]
    // compacted BVH work queue, structure-of-arrays so a wave's appends and fetches are contiguous
    q4* bq_a;              // {origin, closest t among the non-mesh primitives}
    q4* bq_b;              // {direction, bits(dst)}
    q4* bq_c;              // closest-hit queries only: {bits(prim), bits(rank), u, v} of that closest hit
    uint32_t* bq_roots;    // [k * bq_cap + item]: mesh-BVH roots (their count rides in bq_b.w's top 3 bits); [item] == REF_NONE: hole
    uint32_t bq_cap;       // items the queue can hold (rays beyond it are traversed inline)
    uint32_t sh_cap;       // entries sh_queue can hold (holes of the chunked appends included)
    uint32_t cq_cap;       // slots of the work queue the closest-hit pass may use (bq_c has this many; <= bq_cap)
    uint32_t* bq_spill;    // GPU: k_bvh's stack entries beyond its short LDS stack, [entry][global lane]
    uint32_t* counters;    // C_TOTAL
    unsigned long long* stats; // ST_TOTAL, accumulated over the whole render
};

struct RenderArgs {
    DCamera cam;
    int32_t width, height, x0, y0, x1, y1;
    int32_t spp, sampler, max_depth, levels;   // levels = max_depth + 2
    float light_atten;
    uint32_t seed;
    uint32_t tiles_x, tiles_y;                 // RT_TILE tiles covering the render rectangle
    int32_t tile_rank, tile_world;
    uint32_t first_tile, num_tiles;            // this batch: owned-tile ordinals [first_tile, first_tile+num_tiles)
    uint32_t npx;                              // num_tiles * RT_TILE*RT_TILE
    float* out_rgb; uint32_t* out_prim; float* out_t;
    int32_t slab;                              // 1: outputs are compact tile slabs, 0: full frames
};

// path p -> (pixel, sample); paths are sample-major so a wave covers an 8x8 pixel block
struct PathPix { int x, y; uint32_t j, q; bool valid; uint32_t key; size_t out; };

RT_FN PathPix pix_of(const RenderArgs& A, uint32_t q)
{
    PathPix r; r.q = q; r.j = 0;
    uint32_t slot = q / (RT_TILE * RT_TILE), local = q % (RT_TILE * RT_TILE);
    uint32_t tile = (uint32_t)A.tile_rank + (A.first_tile + slot) * (uint32_t)A.tile_world;
    uint32_t tx = tile % A.tiles_x, ty = tile / A.tiles_x;
    uint32_t mt = local / 64, l = local % 64;
    uint32_t lx = (mt % 8) * 8 + (l % 8), ly = (mt / 8) * 8 + (l / 8);
    r.x = A.x0 + (int)(tx * RT_TILE + lx);
    r.y = A.y0 + (int)(ty * RT_TILE + ly);
    r.valid = ty < A.tiles_y && r.x < A.x1 && r.y < A.y1;
    r.key = (uint32_t)(r.y * A.width + r.x);
    r.out = A.slab ? ((size_t)(A.first_tile + slot) * (RT_TILE * RT_TILE) + ly * RT_TILE + lx)
                   : ((size_t)r.y * (size_t)A.width + (size_t)r.x);
    return r;
}
RT_FN PathPix path_of(const RenderArgs& A, uint32_t p)
{
    PathPix r = pix_of(A, p % A.npx);
    r.j = p / A.npx;
    return r;
}

// queue append; on the GPU this is specialised with a wave ballot + one atomic per wave
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline uint32_t queue_slot(uint32_t* counter)
{
    unsigned long long mask = __builtin_amdgcn_ballot_w64(1);
    uint32_t lane = __lane_id();
    uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __builtin_amdgcn_readlane(base, leader);   // wave-uniform: a scalar register, not a vector one
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
#else
inline uint32_t queue_slot(uint32_t* counter) { return (*counter)++; }
#endif

// Ordered key of a ray parameter: unsigned order of the keys = order of the floats (t may be negative),
// so that several work items of one occlusion ray can combine their distances with an integer atomic minimum.
RT_FN uint32_t t_key(float t) { const uint32_t b = f2u(t); return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
RT_FN float key_t(uint32_t k) { return u2f(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }
RT_FN void key_min(uint32_t* p, uint32_t k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, k);
#else
    if (k < *p) *p = k;
#endif
}
RT_FN bool wave_any(bool x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(x) != 0ull;
#else
    return x;
#endif
}

// ---- kernel bodies --------------------------------------------------------------------------------
RT_FN void raygen_item(const DScene& S, const RenderArgs& A, const Batch& B, uint32_t p)
{
    PathPix px = path_of(A, p);
    B.pstate[p] = 0;
    if (!px.valid) { B.lvl_kind[p] = LK_NONE; return; }
    float sx, sy;
    sample_point(A.sampler, A.spp, A.seed, px.key, (int)px.j, sx, sy);
    float fx = ((float)px.x - (0.5f * (float)A.width)) + sx;   // main.cpp:1639-1640
    float fy = ((float)px.y - (0.5f * (float)A.height)) + sy;
    f3 d = camera_dir(A.cam, fx, fy);
    q4 o4 = {A.cam.eye.x, A.cam.eye.y, A.cam.eye.z, 1.0f};
    q4 d4 = {d.x, d.y, d.z, 0.0f};
    B.ray_o[0][p] = o4; B.ray_d[0][p] = d4;
}

RT_FN uint32_t item_path(const RenderArgs& A, const Batch& B, int depth, uint32_t idx)
{
    return depth == 0 ? idx : B.queue[depth & 1][idx];
}

// Chunked queue append for the GPU's hot queues.  A single counter word serves only ~90 returning
// atomics per microsecond, so a wave reserves CHUNK consecutive slots with ONE atomic and hands
// them out by ballot + prefix sum from a wave-uniform cursor.  Slots a wave reserves but never
// fills (the tail of a chunk) are marked as holes (`hole_value` in `holes[slot]`); the consumer
// skips them.  Must be called by all 64 lanes of the wave (`want` selects).  Returns the slot of a
// lane that wants one; a slot >= cap must not be written.
struct QCursor { uint32_t next, end, chunk = 0; };   // chunk 0 = the queue's largest chunk

// Chunk size of one launch (wave-uniform): every wave leaves up to one chunk of holes behind, which the consumer
// has to step over, so a launch whose waves append few items each (small frames, deep levels) takes small chunks:
// about a quarter of a wave's expected appends, a power of two in [64, largest].
RT_FN uint32_t queue_chunk(uint64_t appends, uint32_t waves, uint32_t largest)
{
    const uint64_t per_wave = appends / (waves ? waves : 1u) / 4u;
    uint32_t c = 64u;
    while (c < largest && c * 2u <= per_wave) c *= 2u;
    return c;
}
#ifndef RT_QCHUNK
#define RT_QCHUNK 1024u    // BVH work queue (k_trace, k_ao)
#endif
#ifndef RT_SHCHUNK
#define RT_SHCHUNK 512u    // shadow queue (k_prep); 256: k_prep +1.2 ms, 1024: k_shadow +0.5 ms (more holes)
#endif
#ifndef RT_AO_BLOCK
#define RT_AO_BLOCK 4096u   // slots per block of the AO kernel's blocked stratum-major order (power of two)
#endif

// CHUNK is the largest chunk; qc.chunk (queue_chunk() below) is what the launch uses.
template <uint32_t CHUNK>
RT_FN uint32_t chunk_append(uint32_t* counter, bool want, QCursor& qc, uint32_t* holes, uint32_t hole_value, uint32_t cap)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
    const uint32_t nw = (uint32_t)__popcll(mask);
    if (nw == 0) return 0xFFFFFFFFu;
    const uint32_t lane = __lane_id();
    if (qc.next + nw > qc.end) {
        const uint32_t rem = qc.end - qc.next;   // < 64
        if (lane < rem && qc.next + lane < cap) holes[qc.next + lane] = hole_value;
        uint32_t b0 = 0;
        const uint32_t chunk = qc.chunk ? qc.chunk : CHUNK;
        if (lane == 0) b0 = atomicAdd(counter, chunk);
        b0 = __shfl(b0, 0);   // (as a scalar broadcast, readfirstlane, the gate-walk kernels measured 1 % slower: the cursor then competes for their scalar registers)
        qc.next = b0; qc.end = b0 + chunk;
    }
    const uint32_t slot = qc.next + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    qc.next += nw;
    return slot;
#else
    (void)qc; (void)holes; (void)hole_value; (void)cap;
    return want ? (*counter)++ : 0xFFFFFFFFu;
#endif
}

// marks the unused rest of the wave's last chunk as holes (GPU; all lanes of the wave)
RT_FN void close_cursor(const QCursor& qc, uint32_t* holes, uint32_t hole_value, uint32_t cap)
{
#if defined(__HIP_DEVICE_COMPILE__)
    for (uint32_t s = qc.next + __lane_id(); s < qc.end && s < cap; s += 64u) holes[s] = hole_value;
#else
    (void)qc; (void)holes; (void)hole_value; (void)cap;
#endif
}

// Queue item word b.w: the ray's destination index (31 bits) | bit 31 when the item carries more than one mesh root.
// The roots of an item lie in bq_roots[k * bq_cap + slot], k = 0 .. sp-1; an item with 1 < sp < RT_DEFER_ROOTS roots
// ends its list with REF_NONE (round 3: the count used to ride in three bits of b.w, which left 29 bits for the
// destination and forced a 4K / 16 spp / 4x4 AO frame into four batches).
#define RT_BQ_MULTI 0x80000000u
#define RT_BQ_DST_MASK 0x7FFFFFFFu
RT_FN uint32_t bq_word(uint32_t dst, uint32_t sp) { return dst | (sp > 1u ? RT_BQ_MULTI : 0u); }
RT_FN void bq_store_roots(const Batch& B, uint32_t slot, Stack st, uint32_t sp)
{
    for (uint32_t k = 0; k < sp; ++k) B.bq_roots[(size_t)k * B.bq_cap + slot] = st.p[k * st.stride];
    if (sp > 1u && sp < RT_DEFER_ROOTS) B.bq_roots[(size_t)sp * B.bq_cap + slot] = REF_NONE;
}

// One closest-hit query (the scene loop of raytrace(), Reflective::rho, Transparency::rho).  Like ao_item the hot
// kernel traverses nothing: planes / spheres / free triangles are tested in place, the roots of the Meshes whose
// gates pass ride in ONE queue item for k_bvh (a closest-hit record cannot be combined from several items the way
// a distance can).  A ray with more than RT_DEFER_ROOTS such Meshes, or one that finds the queue full, sets
// `overflow`: its wave's rays are redone, traversed in place, by trace_overflow_item after k_bvh has run.
// GPU: called by all 64 lanes of a wave (`valid` = false for lanes without an item): the queue cursor votes
template <bool STATS>
RT_FN void trace_item(const DScene& S, const RenderArgs& A, const Batch& B, int depth, uint32_t idx, bool valid, Stack st, Counters& cnt, QCursor& qc, bool& overflow)
{
    uint32_t p = 0, sp = 0;
    f3 o = F3(0, 0, 0), d = F3(0, 0, 1);
    Hit h; h.t = FLT_MAX; h.prim = H_MISS; h.u = 0; h.v = 0; h.rank = 0;
    if (valid) {
        p = item_path(A, B, depth, idx);
        valid = !(depth == 0 && !path_of(A, p).valid);
    }
    if (valid) {
        const q4 o4 = B.ray_o[depth & 1][p], d4 = B.ray_d[depth & 1][p];
        o = xyz(o4); d = xyz(d4);
        if (!(S.debug & 2u)) {
            const RayGate rg = make_gate(S.cull_pad, S.extent, o, d);
            uint32_t i = lead_planes_hit(S, o, d, rg, h);
            i = walk_gates<STATS>(S, o, d, rg, i, h, st, sp, cnt);
            if (i < S.num_gates) { overflow = true; sp = 0; }   // more Meshes than one item carries
        }
    }
    const bool want = sp > 0;
    const uint32_t slot = chunk_append<RT_QCHUNK>(&B.counters[C_BQ + depth], want, qc, B.bq_roots, (uint32_t)REF_NONE, B.cq_cap);
    if (want && slot < B.cq_cap) {
        const q4 a = {o.x, o.y, o.z, h.t};
        const q4 b = {d.x, d.y, d.z, u2f(bq_word(p, sp))};   // p < 2^31 (checked where the batch is sized); 1 <= sp <= RT_DEFER_ROOTS
        const q4 c = {u2f(h.prim), u2f(h.rank), h.u, h.v};
        B.bq_a[slot] = a; B.bq_b[slot] = b; B.bq_c[slot] = c;
        bq_store_roots(B, slot, st, sp);
        return;   // k_bvh writes hit[p]
    }
    if (want) overflow = true;   // queue full
    if (!valid) return;
    q4 r = {h.t, u2f(h.prim), h.u, h.v};
    B.hit[p] = r;
}

// The rays of a wave of trace_item that set `overflow`, each traversed in place.  `entry` = the wave's first item.
template <bool STATS>
RT_FN void trace_overflow_item(const DScene& S, const RenderArgs& A, const Batch& B, int depth, uint32_t entry, uint32_t lane, uint32_t n, Stack st, Counters& cnt)
{
    const uint32_t idx = entry + lane;
    if (idx >= n) return;
    const uint32_t p = item_path(A, B, depth, idx);
    if (depth == 0 && !path_of(A, p).valid) return;
    const q4 o4 = B.ray_o[depth & 1][p], d4 = B.ray_d[depth & 1][p];
    Hit h; float sh;
    trace<0, STATS, false>(S, xyz(o4), xyz(d4), FLT_MAX, h, sh, st, cnt);
    q4 r = {h.t, u2f(h.prim), h.u, h.v};
    B.hit[p] = r;
}

// One item of the compacted BVH kernel: every lane holds a ray that has triangle BVHs to search.
// MODE 0: closest-hit query, result -> hit[dst];  MODE 1: occlusion distance -> minimum into ao_key[dst];
// MODE 2: shadow query over all-opaque meshes: any triangle closer than the light -> sh_res[dst] = 0.
enum { BVH_CLOSEST = 0, BVH_AO = 1, BVH_SHADOW = 2 };
template <int MODE, bool STATS>
RT_FN void bvh_item(const DScene& S, const Batch& B, uint32_t idx, Stack st, Counters& cnt)
{
    q4 a = B.bq_a[idx], b = B.bq_b[idx];
    f3 o = xyz(a), d = xyz(b);
    Hit best; best.t = a.w; best.prim = H_MISS; best.rank = 0; best.u = 0; best.v = 0;
    if (MODE == BVH_CLOSEST) { q4 c = B.bq_c[idx]; best.prim = f2u(c.x); best.rank = f2u(c.y); best.u = c.z; best.v = c.w; }
    RayCull4 rc = make_cull4(S, o, d);
    const uint32_t dst = f2u(b.w) & RT_BQ_DST_MASK;
    const uint32_t maxroots = (f2u(b.w) & RT_BQ_MULTI) ? (uint32_t)RT_DEFER_ROOTS : 1u;
    if (B.bq_roots[idx] == REF_NONE) return;   // hole
    for (uint32_t k = 0; k < maxroots; ++k) {
        uint32_t sp = 1;
        st.p[0] = B.bq_roots[(size_t)k * B.bq_cap + idx];
        if (st.p[0] == REF_NONE) break;   // end of a multi-root list
        if (MODE == BVH_SHADOW) {
            if (bvh_drain<STATS, true>(S, o, d, rc, best, st, sp, cnt)) { B.sh_res[dst] = 0.0f; return; }
        } else bvh_drain<STATS, false>(S, o, d, rc, best, st, sp, cnt);
    }
    if (MODE == BVH_AO) key_min(&B.ao_key[dst], t_key(best.t));
    else if (MODE == BVH_CLOSEST) { q4 r = {best.t, u2f(best.prim), best.u, best.v}; B.hit[dst] = r; }
}

// Returns true when the path's material consumes shadow rays (Specular / Textured); P and N are then
// the hit point and shading normal.  The AO "queue" is positional: slot = idx, no append.
// `skipped` counts the queries of the reference that are not traced here: Ambient::L's n*n rays of a surface whose
// ka is 0 (the reference traces them and multiplies the factor away, main.cpp:1497 with main.cpp:1714's ka = 0).
RT_FN bool prep_body(const DScene& S, const RenderArgs& A, const Batch& B, int depth, uint32_t idx, uint32_t& p_out, f3& P_out, f3& N_out, uint32_t& skipped)
{
    { q4 none = {0.0f, 0.0f, 0.0f, u2f(0xFFFFFFFFu)}; B.ao_w[idx] = none; }
    uint32_t p = item_path(A, B, depth, idx);
    PathPix px = path_of(A, p);
    p_out = p;
    if (depth == 0 && !px.valid) return false;
    // the records the path index addresses are requested together, before the first test (see shade_item)
    q4 h4 = B.hit[p];
    q4 o4 = B.ray_o[depth & 1][p], d4 = B.ray_d[depth & 1][p];
    uint32_t ps = B.pstate[p];
    Hit h; h.t = h4.x; h.prim = f2u(h4.y); h.u = h4.z; h.v = h4.w;
    size_t li = (size_t)depth * B.P + p;
    f3 o = xyz(o4), d = xyz(d4);
    if (h.prim == H_MISS) {
        B.lvl_kind[li] = LK_NONE;
        if (depth == 0 && px.j == 0) {
            if (A.out_prim) A.out_prim[px.out] = 0xFFFFFFFFu;
            if (A.out_t) A.out_t[px.out] = h.t;
        }
        return false;
    }
    Surface sf = surface_of(S, h, o, d);
    if (depth == 0 && px.j == 0) {
        if (A.out_prim) A.out_prim[px.out] = sf.prim_id;
        if (A.out_t) A.out_t[px.out] = h.t;
    }
    q4 n4 = {sf.normal.x, sf.normal.y, sf.normal.z, u2f((uint32_t)sf.material)};
    B.nrm[p] = n4;
    h4.z = sf.u; h4.w = sf.v;
    B.hit[p] = h4;
    const DMaterial M = S.materials[sf.material];
    f3 P = o + d * h.t;   // sr.ray(sr.t) = o + t*d

    // ambient term: every class but Refraction evaluates world->ambient->L(sr)
    uint32_t slot = 0xFFFFFFFFu;
    if (M.kind != RT_MAT_REFRACTION && S.ambient.occlusion_samples > 0) {
        uint32_t a = ps & 0xFFu;
        ps = ((a + 1u) & 0xFFu) | (a << 8);
        B.pstate[p] = ps;
        if (M.ka == 0.0f) skipped += (uint32_t)(S.ambient.occlusion_samples * S.ambient.occlusion_samples);
        if (M.ka != 0.0f) {   // ka == 0 multiplies the (finite) occlusion factor away: no rays needed
            slot = idx;
            f3 ao, au, aw;
            ao_frame(P, sf.normal, ao, au, aw);
            q4 r0 = {ao.x, ao.y, ao.z, u2f(rt_rand31_prefix(A.seed, px.key, px.j))}, r1 = {au.x, au.y, au.z, 0.0f}, r2 = {aw.x, aw.y, aw.z, u2f(a)};
            B.ao_o[slot] = r0; B.ao_u[slot] = r1; B.ao_w[slot] = r2;
        }
    }
    B.ao_slot[p] = slot;

    P_out = P; N_out = sf.normal;

    if (M.kind == RT_MAT_MIRROR || M.kind == RT_MAT_REFRACTION) {
        B.lvl_kind[li] = M.kind == RT_MAT_MIRROR ? LK_MIRROR : LK_REFRACT;
        if (depth + 1 < A.levels) B.lvl_kind[li + B.P] = LK_NONE;
        if (!((unsigned)depth > (unsigned)A.max_depth)) {   // `if (sr.depth > maxBounceDepth) return sr.color`
            f3 r; float ior = 1.0f;
            if (M.kind == RT_MAT_MIRROR) {
                f3 wo = normalize(-d);
                r = reflect(-wo, sf.normal);
            } else {
                float enter = M.ior;
                r = refraction_dir(d, sf.normal, o4.w, enter);
                ior = enter;
            }
            f3 no = (o + d * h.t) + r * 0.1f;
            q4 no4 = {no.x, no.y, no.z, ior};
            q4 nd4 = {r.x, r.y, r.z, 0.0f};
            B.ray_o[(depth + 1) & 1][p] = no4; B.ray_d[(depth + 1) & 1][p] = nd4;
            uint32_t e = queue_slot(&B.counters[C_QUEUE + depth + 1]);
            B.queue[(depth + 1) & 1][e] = p;
        }
    } else {
        B.lvl_kind[li] = LK_TERMINAL;
    }
    return M.kind == RT_MAT_SPECULAR || M.kind == RT_MAT_TEXTURED;
}

// `valid` = idx addresses a path; all 64 lanes of a wave call this together (GPU): the shadow-ray
// appends go through the wave's chunked cursor.
// `skipped` += the reference's queries that this path does not trace (rt_stats.rays_skipped): the shadow ray of a
// light the surface faces away from (the reference calls shadowed() before it tests n.wi, main.cpp:1227, 1281) and
// the AO rays of a surface with ka = 0 (prep_body).
RT_FN void prep_item(const DScene& S, const RenderArgs& A, const Batch& B, int depth, uint32_t idx, bool valid, QCursor& qsh, uint32_t& skipped)
{
    uint32_t p = 0;
    f3 P = F3(0, 0, 0), N = F3(0, 0, 1);
    const bool lit = valid && prep_body(S, A, B, depth, idx, p, P, N, skipped);
    // the shadow ray is only consumed when n.wi > 0 (main.cpp:1227-1237, 1281-1293)
    for (uint32_t l = 0; l < S.num_lights; ++l) {
        bool want = false;
        if (lit) {
            f3 wi = normalize(light_direction(S.lights[l], P));
            want = dot(N, wi) > 0.0f;
            if (!want) skipped++;
        }
        const uint32_t e = chunk_append<RT_SHCHUNK>(&B.counters[C_SH + depth], want, qsh, B.sh_queue, 0xFFFFFFFFu, B.sh_cap);
        if (want && e < B.sh_cap) B.sh_queue[e] = p * S.num_lights + l;   // sh_cap covers the worst case (carve_batch)
    }
}

template <bool STATS, int DEPTH = RT_STACK_DEPTH>
RT_FN void shadow_item(const DScene& S, const RenderArgs& A, const Batch& B, int depth, uint32_t idx, Stack st, Counters& cnt)
{
    uint32_t e = B.sh_queue[idx];
    if (e == 0xFFFFFFFFu) return;   // hole left by the chunked appends
    cnt.rays++;
    uint32_t p = e / S.num_lights, l = e % S.num_lights;
    q4 o4 = B.ray_o[depth & 1][p], d4 = B.ray_d[depth & 1][p];
    f3 P = xyz(o4) + xyz(d4) * B.hit[p].x;
    f3 src = light_source(S.lights[l], P);
    f3 dir = normalize(P - src);
    float dist = length(P - src) - 0.01f;
    Hit h; float sh = 1.0f;
    trace<1, STATS, false, DEPTH>(S, src, dir, dist, h, sh, st, cnt);
    B.sh_res[e] = sh;
}

// The same query on a scene whose Meshes all lie in all-opaque world->scene entries (DScene::shadow_defer): the hot
// kernel traverses nothing.  What the primitives tested in place decide is stored at once (1, 0.8 or 0); unless that
// is already 0, the roots of the Meshes whose gates pass ride in one queue item for k_bvh<BVH_SHADOW>, which stores 0
// when any of their triangles lies before the light.  More than RT_DEFER_ROOTS such Meshes, or a full queue, set
// `overflow`: the wave's rays are redone in place by shadow_overflow_item after k_bvh has run.
// GPU: called by all 64 lanes of a wave (`valid` = false for lanes without an item): the queue cursor votes.
template <bool STATS>
RT_FN void shadow_defer_item(const DScene& S, const RenderArgs& A, const Batch& B, int depth, uint32_t idx, bool valid, Stack st, Counters& cnt,
                             QCursor& qc, bool& overflow)
{
    uint32_t e = valid ? B.sh_queue[idx] : 0xFFFFFFFFu, sp = 0;
    valid = e != 0xFFFFFFFFu;   // hole left by the chunked appends
    f3 src = F3(0, 0, 0), dir = F3(0, 0, 1);
    float dist = 0.0f;
    if (valid) {
        cnt.rays++;
        const uint32_t p = e / S.num_lights, l = e % S.num_lights;
        const q4 o4 = B.ray_o[depth & 1][p], d4 = B.ray_d[depth & 1][p];
        const f3 P = xyz(o4) + xyz(d4) * B.hit[p].x;
        src = light_source(S.lights[l], P);
        dir = normalize(P - src);
        dist = length(P - src) - 0.01f;
        Hit h; float sh = 1.0f;
        sp = trace<1, STATS, true>(S, src, dir, dist, h, sh, st, cnt);
        if (sp > RT_DEFER_ROOTS) { overflow = true; sp = 0; }
        if (sh == 0.0f) sp = 0;   // decided
        B.sh_res[e] = sh;
    }
    const bool want = sp > 0;
    const uint32_t slot = chunk_append<RT_QCHUNK>(&B.counters[C_BQS + depth], want, qc, B.bq_roots, (uint32_t)REF_NONE, B.bq_cap);
    if (want) {
        if (slot < B.bq_cap) {
            const q4 a = {src.x, src.y, src.z, dist};
            const q4 b = {dir.x, dir.y, dir.z, u2f(bq_word(e, sp))};   // e < 2^31 (checked where shadow_defer is decided)
            B.bq_a[slot] = a; B.bq_b[slot] = b;
            bq_store_roots(B, slot, st, sp);
        } else overflow = true;   // queue full
    }
}

// The rays of a wave of shadow_defer_item that set `overflow`, each traversed in place.  `entry` = the wave's first item.
template <bool STATS>
RT_FN void shadow_overflow_item(const DScene& S, const RenderArgs& A, const Batch& B, int depth, uint32_t entry, uint32_t lane, uint32_t n, Stack st, Counters& cnt)
{
    if (entry + lane < n) shadow_item<STATS>(S, A, B, depth, entry + lane, st, cnt);
}

// AO work item -> (slot, stratum k) for the refilling walk kernel (k_ao_walk, GPU), whose items are single rays.
// Blocked stratum-major order: RT_AO_BLOCK consecutive slots x all n*n
// strata, stratum by stratum, so a wave traces the same stratum (x,y) of 64 neighbouring hits, whose rays leave
// nearby points in similar directions, and the 16 passes over a block's hit records (~70 B each) are served by L2.
// item = base + off with base a multiple of 256 that is the same for the whole workgroup (GPU) and off < 256:
// block and stratum come from `base` alone, i.e. from scalar registers.
RT_FN void ao_coords(const DScene& S, uint32_t nslots, uint32_t base, uint32_t off, uint32_t& slot, uint32_t& k)
{
    const uint32_t n2b = (uint32_t)(S.ambient.occlusion_samples * S.ambient.occlusion_samples);
    const uint32_t nfull = nslots & ~(RT_AO_BLOCK - 1u);
    if (base < nfull * n2b) {
        const uint32_t b = base / (RT_AO_BLOCK * n2b), r = base % (RT_AO_BLOCK * n2b);
        k = r / RT_AO_BLOCK; slot = b * RT_AO_BLOCK + (r & (RT_AO_BLOCK - 1u)) + off;
    } else {
        const uint32_t r = base + off - nfull * n2b, tail = nslots - nfull;
        k = r / tail; slot = nfull + r % tail;
    }
}

// ao_key index of the distance of stratum k of the path in AO slot `slot`: stratum-major, so that the 64 lanes of a
// wave (64 consecutive slots, one stratum at a time) store and k_shade loads contiguous words.
RT_FN size_t ao_index(const Batch& B, uint32_t slot, uint32_t k) { return (size_t)k * B.P + slot; }

// The n*n AO rays of one hit (Ambient::L's loop, main.cpp:1372-1398).  One lane = one AO slot, all its strata in
// turn: the AO frame is read once per hit, and at any time the lanes of a wave trace the SAME stratum of 64
// neighbouring hits, i.e. similar rays.  The hot kernel traverses nothing: it walks the gate nodes, tests planes /
// spheres / free triangles in place and queues the triangle-BVH roots of the Meshes whose gates pass for k_bvh —
// one queue item per RT_DEFER_ROOTS roots, as many items as the ray needs.  All of a ray's distances (its own and
// its items') meet in ao_key by minimum.  A wave that finds the queue full sets `overflow`: all rays of its 64
// slots are redone by ao_overflow_item.
// GPU: called by all 64 lanes of a wave (`valid` = false for lanes without an item): the queue cursor votes.
// WAVE (GPU only): the gate walk is made by the wave in step (walk_gates_wave, rt_core.hpp) instead of lane by lane.
template <bool STATS, bool WAVE = false>
RT_FN void ao_item(const DScene& S, const RenderArgs& A, const Batch& B, int depth, uint32_t slot, bool valid, Stack st,
                   Counters& cnt, QCursor& qc, bool& overflow)
{
    const uint32_t n2 = (uint32_t)(S.ambient.occlusion_samples * S.ambient.occlusion_samples);
    // slot = position of the path in this depth's path list; paths without AO rays hold no path id
    // The frame is read once per hit and held across the walks of its n*n rays (round 3: the gate-walk kernels have the
    // registers for it since they are built without SLP vectorisation; until then it was re-read for every stratum).
#ifndef RT_AO_REREAD
    q4 ro = {0, 0, 0, 0}, ru = ro, rw = ro;
    if (valid) { rw = B.ao_w[slot]; valid = f2u(rw.w) != 0xFFFFFFFFu; }
    if (valid) { ro = B.ao_o[slot]; ru = B.ao_u[slot]; }
    const f3 rv = cross(xyz(rw), xyz(ru));
#else
    if (valid) valid = f2u(B.ao_w[slot].w) != 0xFFFFFFFFu;
#endif
    for (uint32_t k = 0; k < n2; ++k) {   // wave-uniform
        f3 o = F3(0, 0, 0), dir = F3(0, 0, 1);
        Hit h; h.t = S.ambient.occlusion_dist; h.prim = H_MISS; h.u = 0; h.v = 0; h.rank = 0;
        uint32_t sp = 0, i = S.num_gates;
        const uint32_t dst = (uint32_t)ao_index(B, slot, k);
        RayGate rg = {};
        if (valid) {
#ifdef RT_AO_REREAD
            const q4 ro = B.ao_o[slot], ru = B.ao_u[slot], rw = B.ao_w[slot];
            const f3 rv = cross(xyz(rw), xyz(ru));
#endif
            cnt.rays++;
            o = xyz(ro);
            dir = ao_dir(S.ambient, xyz(ru), xyz(rw), rv, f2u(ro.w), f2u(rw.w), k);
            rg = make_gate(S.cull_pad, S.extent, o, dir);
            i = (S.debug & 2u) ? S.num_gates : lead_planes_hit<true>(S, o, dir, rg, h);   // DIST: only h.t leaves this function
        }
        for (;;) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (WAVE) i = walk_gates_wave<STATS>(S, o, dir, rg, i, h.t, st, sp, cnt);
            else
#endif
            if (i < S.num_gates) i = walk_gates<STATS, true>(S, o, dir, rg, i, h, st, sp, cnt);
            const bool want = sp > 0;
            const uint32_t item = chunk_append<RT_QCHUNK>(&B.counters[C_BQA + depth], want, qc, B.bq_roots, (uint32_t)REF_NONE, B.bq_cap);
            if (want) {
                if (item < B.bq_cap) {
                    const q4 a = {o.x, o.y, o.z, h.t};   // an upper bound of the ray's distance: the minimum is taken at the destination
                    const q4 b = {dir.x, dir.y, dir.z, u2f(bq_word(dst, sp))};   // dst < 2^31 (checked where the batch is sized); 1 <= sp <= RT_DEFER_ROOTS
                    B.bq_a[item] = a; B.bq_b[item] = b;
                    bq_store_roots(B, item, st, sp);
                } else overflow = true;
                sp = 0;
            }
            if (!wave_any(i < S.num_gates)) break;
        }
        if (valid) B.ao_key[dst] = t_key(h.t);   // stays occlusion_dist when nothing closer was found
    }
}

// The rays of 64 AO slots whose wave found the queue full, each traversed in place, start to end (the queue is
// sized so that this does not happen with the reference's scenes; tests force it).  `entry` = the first slot.
template <bool STATS>
RT_FN void ao_overflow_item(const DScene& S, const RenderArgs& A, const Batch& B, uint32_t entry, uint32_t lane, uint32_t nslots, Stack st, Counters& cnt)
{
    const uint32_t n2 = (uint32_t)(S.ambient.occlusion_samples * S.ambient.occlusion_samples);
    const uint32_t slot = entry + lane;
    if (slot >= nslots) return;
    const q4 rw = B.ao_w[slot];
    if (f2u(rw.w) == 0xFFFFFFFFu) return;
    const q4 ro = B.ao_o[slot], ru = B.ao_u[slot];
    for (uint32_t k = 0; k < n2; ++k) {
        const f3 dir = ao_dir(S.ambient, xyz(ru), xyz(rw), f2u(ro.w), f2u(rw.w), k);
        Hit h; float sh;
        trace<0, STATS, false>(S, xyz(ro), dir, S.ambient.occlusion_dist, h, sh, st, cnt);
        key_min(&B.ao_key[ao_index(B, slot, k)], t_key(h.t));
    }
}

// world->ambient->L(sr): Light::L, or Ambient::L's reduction of the n*n distances in x,y order
// `slot` = B.ao_slot[p] (0xFFFFFFFF: the path shot no AO rays; not read when the ambient light has no occlusion samples)
RT_FN f3 ambient_L(const DScene& S, const Batch& B, uint32_t slot)
{
    const rt_ambient& A = S.ambient;
    f3 base = F3(A.colour[0], A.colour[1], A.colour[2]) * A.radiance;
    if (A.occlusion_samples <= 0) return base;
    if (slot == 0xFFFFFFFFu) return base;
    unsigned n = (unsigned)A.occlusion_samples;
    float dist = A.occlusion_dist;
    float maxSum = (dist * (float)n * (float)n);
    float occluded = maxSum;
    for (unsigned k = 0; k < n * n; ++k) occluded -= (dist - key_t(B.ao_key[ao_index(B, slot, k)]));   // x,y order = stratum order
    float f = occluded / maxSum;
    return base * f * f;
}

RT_FN void shade_item(const DScene& S, const RenderArgs& A, const Batch& B, int depth, uint32_t idx)
{
    uint32_t p = item_path(A, B, depth, idx);
    if (depth == 0 && !path_of(A, p).valid) return;
    const uint32_t nl = S.num_lights;
    // Everything the path index alone addresses is requested before the first test (round 3): this kernel is a chain of
    // dependent loads (hit -> normal -> material -> ray -> AO slot -> 16 distances -> texture -> texel: 91 % of its wave
    // cycles were waits), and a load behind an early return cannot be started before the branch is decided.  Stale
    // records of paths that missed are read (in range) and not used.
    q4 h4 = B.hit[p];
    q4 n4 = B.nrm[p];
    q4 o4 = B.ray_o[depth & 1][p], d4 = B.ray_d[depth & 1][p];
    const uint32_t slot = S.ambient.occlusion_samples > 0 ? B.ao_slot[p] : 0xFFFFFFFFu;
    const float shadow0 = nl > 0 ? B.sh_res[(size_t)p * nl] : 1.0f;
    if (f2u(h4.y) == H_MISS) return;
    size_t li = (size_t)depth * B.P + p;
    f3 N = xyz(n4);
    const DMaterial M = S.materials[f2u(n4.w)];
    const f3 amb = ambient_L(S, B, slot);   // a Refraction surface has no slot: no loads
    if (M.kind == RT_MAT_REFRACTION) return;   // Refraction::shade = clamp(child): nothing local
    f3 o = xyz(o4), d = xyz(d4);
    f3 P = o + d * h4.x;
    f3 colour = F3(M.colour[0], M.colour[1], M.colour[2]);
    q4 out;
    if (M.kind == RT_MAT_MATTE) {   // main.cpp:1143-1165
        f3 L = (colour * M.ka) * amb;
        for (uint32_t l = 0; l < nl; ++l) {
            f3 wi = light_direction(S.lights[l], P);
            float nDotWi = dot(N, wi);
            if (nDotWi > 0.0f) L = L + ((colour * M.kd) * 0.318309886183790671537767526745028724f) * light_L(S.lights[l]) * nDotWi;
        }
        out = {L.x, L.y, L.z, 0.0f};
    } else if (M.kind == RT_MAT_SPECULAR) {   // main.cpp:1215-1244
        f3 wo = normalize(-d);
        f3 L = (colour * M.ka) * amb;
        for (uint32_t l = 0; l < nl; ++l) {
            f3 wi = normalize(light_direction(S.lights[l], P));
            float nDotWi = dot(N, wi);
            if (nDotWi > 0.0f) {
                float shadow = l == 0 ? shadow0 : B.sh_res[p * nl + l];
                L = L + ((colour * M.kd) * 0.318309886183790671537767526745028724f) * light_L(S.lights[l]) * nDotWi * shadow;
                float sp = specular_fn(M.ks, M.spec_exp, N, wo, wi);
                L = L + F3(sp, sp, sp) * shadow;
            }
        }
        L = clamp01(L);
        out = {L.x, L.y, L.z, 0.0f};
    } else if (M.kind == RT_MAT_TEXTURED) {   // main.cpp:1269-1299
        f3 tex = texture_colour(S, M.texture, h4.z, h4.w);
        f3 wo = normalize(-d);
        f3 L = (F3(1.0f, 1.0f, 1.0f) * M.ka) * amb * tex;
        for (uint32_t l = 0; l < nl; ++l) {
            f3 wi = normalize(light_direction(S.lights[l], P));
            float nDotWi = dot(N, wi);
            if (nDotWi > 0.0f) {
                float shadow = l == 0 ? shadow0 : B.sh_res[p * nl + l];
                float ldist = length(light_source(S.lights[l], P) - P);
                float att = A.light_atten / (ldist * ldist);
                L = L + tex * light_L(S.lights[l]) * nDotWi * att * shadow;
                float sp = specular_fn(M.ks, M.spec_exp, N, wo, wi);
                L = L + F3(sp, sp, sp) * att * shadow;
            }
        }
        L = clamp01(L);
        out = {L.x, L.y, L.z, 0.0f};
    } else {   // MIRROR (main.cpp:1487-1511): store the ambient term and the per-light phong terms
        f3 wo = normalize(-d);
        f3 L = (colour * M.ka) * amb;
        out = {L.x, L.y, L.z, 0.0f};
        for (uint32_t l = 0; l < nl; ++l) {
            f3 wi = normalize(light_direction(S.lights[l], P));
            float sp = dot(N, wi) > 0.0f ? specular_fn(M.ks, M.spec_exp, N, wo, wi) : 0.0f;
            if (l == 0) out.w = sp; else B.lvl_ph[li * (nl - 1) + (l - 1)] = sp;
        }
    }
    B.lvl[li] = out;
}

// per pixel: unwind every sample's chain deepest level first, average, store
RT_FN void resolve_item(const DScene& S, const RenderArgs& A, const Batch& B, uint32_t q)
{
    PathPix px = pix_of(A, q);
    if (!px.valid) return;
    const uint32_t nl = S.num_lights;
    f3 sum = F3(0.0f, 0.0f, 0.0f);
    for (int j = 0; j < A.spp; ++j) {
        uint32_t p = (uint32_t)j * A.npx + q;
        int top = 0;
        while (top < A.levels) {
            uint8_t k = B.lvl_kind[(size_t)top * B.P + p];
            if (k == LK_MIRROR || k == LK_REFRACT) top++; else break;
        }
        f3 L = F3(0.0f, 0.0f, 0.0f);
        bool hit0 = B.lvl_kind[p] != LK_NONE;
        if (top < A.levels && B.lvl_kind[(size_t)top * B.P + p] == LK_TERMINAL) L = xyz(B.lvl[(size_t)top * B.P + p]);
        for (int dlv = top - 1; dlv >= 0; --dlv) {
            size_t li = (size_t)dlv * B.P + p;
            if (B.lvl_kind[li] == LK_MIRROR) {
                q4 m = B.lvl[li];
                L = xyz(m) + L;                       // L = ambient; L += rho
                if (nl > 0) L = L + F3(m.w, m.w, m.w);  // L += phong (light 0)
                for (uint32_t l = 1; l < nl; ++l) { float sp = B.lvl_ph[li * (nl - 1) + (l - 1)]; L = L + F3(sp, sp, sp); }
            }
            L = clamp01(L);
        }
        if (hit0) sum = sum + L;
    }
    float avg = 1.0f / (float)A.spp;
    float* o = A.out_rgb + px.out * 3;
    o[0] = sum.x * avg; o[1] = sum.y * avg; o[2] = sum.z * avg;
}

// ---- host-side sequencing ---------------------------------------------------------------------------
// Backend concept:
//   void begin_batch(const RenderArgs&)          zero the per-batch counters
//   void raygen(A, n) / trace(A,d,n,fixed) / prep / shadow / ao / shade / resolve(A, n)
// where for depth > 0 the item count lives in device memory (counter index given) and the
// backend launches a grid-stride kernel that reads it.
template <class Backend>
void render_frame(Backend& be, RenderArgs A, uint32_t owned_tiles, uint32_t tiles_per_batch, bool bounces)
{
    for (uint32_t first = 0; first < owned_tiles; first += tiles_per_batch) {
        A.first_tile = first;
        A.num_tiles = owned_tiles - first < tiles_per_batch ? owned_tiles - first : tiles_per_batch;
        A.npx = A.num_tiles * RT_TILE * RT_TILE;
        uint32_t paths = A.npx * (uint32_t)A.spp;
        be.begin_batch(A);
        be.raygen(A, paths);
        int levels = bounces ? A.levels : 1;
        for (int d = 0; d < levels; ++d) {
            be.trace(A, d, paths);
            be.bvh_trace(A, d);
            be.trace_overflow(A, d, paths);
            be.prep(A, d, paths);
            be.shadow(A, d);          // on scenes with shadow_defer: shadow_defer -> bvh<shadow> -> shadow_overflow
            be.ao(A, d, paths);
            be.bvh_ao(A, d);
            be.ao_overflow(A, d, paths);
            be.shade(A, d, paths);
        }
        be.resolve(A, A.npx);
    }
}

} // namespace rt
