// rt_lbvh.hip — SURVEY 8f N1: the triangle BVHs of a scene built on the device (LBVH: Morton codes, one
// radix sort, Karras' radix tree, bottom-up box fit), in the formats the traversal kernels read: 48-byte
// triangle records in leaf order and 64-byte quantised four-child nodes (collapsed from the two-child radix
// tree) with <= 2 triangles per leaf.
//
// The reference has no counterpart (its Mesh::hit scans every triangle, main.cpp:890-912); on this path a
// BVH only culls, so a frame rendered with these trees is bit-identical to one rendered with the host's
// binned-SAH trees (tests/test_gpu_parity.py::test_gpu_built_bvh_renders_the_same_frame).  LBVH trees cost
// more node visits per ray than SAH trees; the host builder stays the default and this one is chosen per scene
// (rt_scene_set_builder).  Trees deeper than the traversal stack allows are reported to the caller, which
// falls back to the host builder.
//
// All Mesh occurrences of the scene are built together: the sort key is (occurrence << 32 | 30-bit Morton
// code in the occurrence's box), so one sort groups the triangles by occurrence in slot order, and every
// occurrence gets its own radix tree over its slice of the sorted keys (inner node 0 = root).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <chrono>
#include <string>
#include <vector>

#include "rt_lbvh.hpp"

namespace rt {
namespace {

#define LB_BLOCK 256
struct DOcc { uint32_t first, n, rank_base, slot_off, node_off; float mn[3], inv[3]; };

__device__ inline uint32_t expand10(uint32_t v)
{
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// index of the last entry whose `field` (slot_off or node_off, ascending) is <= x
template <bool BY_NODE>
__device__ inline uint32_t find_occ(const DOcc* occ, uint32_t nocc, uint32_t x)
{
    uint32_t lo = 0, hi = nocc;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t v = BY_NODE ? occ[mid].node_off : occ[mid].slot_off;
        if (v <= x) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(LB_BLOCK) k_keys(const rt_triangle* tri, const DOcc* occ, uint32_t nocc, uint32_t nslots,
                                                   unsigned long long* keys, uint32_t* vals)
{
    const uint32_t s = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (s >= nslots) return;
    const uint32_t j = find_occ<false>(occ, nocc, s);
    const DOcc o = occ[j];
    const uint32_t gid = o.first + (s - o.slot_off);
    const rt_triangle t = tri[gid];
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float mn = fminf(t.v0[a], fminf(t.v1[a], t.v2[a])), mx = fmaxf(t.v0[a], fmaxf(t.v1[a], t.v2[a]));
        const float c = 0.5f * (mn + mx);
        const float f = (c - o.mn[a]) * o.inv[a];
        q[a] = (uint32_t)fminf(fmaxf(f, 0.0f), 1023.0f);
    }
    const uint32_t morton = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
    keys[s] = ((unsigned long long)j << 32) | morton;
    vals[s] = gid;
}

// {v0, bits(global tri)} {e1 = v1 - v0, bits(material)} {e2 = v2 - v0, bits(rank)}: HostScene::write_triangle_slot
__global__ void __launch_bounds__(LB_BLOCK) k_records(const rt_triangle* tri, const DOcc* occ, const unsigned long long* keys,
                                                      const uint32_t* vals, uint32_t nslots, q4* out)
{
    const uint32_t s = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (s >= nslots) return;
    const DOcc o = occ[(uint32_t)(keys[s] >> 32)];
    const uint32_t gid = vals[s];
    const rt_triangle t = tri[gid];
    const uint32_t rank = o.rank_base + (gid - o.first);
    out[(size_t)s * 3 + 0] = q4{t.v0[0], t.v0[1], t.v0[2], u2f(gid)};
    out[(size_t)s * 3 + 1] = q4{t.v1[0] - t.v0[0], t.v1[1] - t.v0[1], t.v1[2] - t.v0[2], u2f((uint32_t)t.material)};
    out[(size_t)s * 3 + 2] = q4{t.v2[0] - t.v0[0], t.v2[1] - t.v0[1], t.v2[2] - t.v0[2], u2f(rank)};
}

// per inner node: children (bit 31 = triangle, else inner node; local indices), key range, parent links
struct Tree {
    uint32_t* left; uint32_t* right; uint32_t* lo; uint32_t* hi;
    uint32_t* parent_node;   // [global node]  parent's global node | side << 31; 0x7FFFFFFF = root
    uint32_t* parent_prim;   // [slot]         same for triangles
    uint32_t* flag;          // arrivals of the bottom-up pass
    float* cbox;             // [global node][side][6]
};

// Karras 2012: the node's key range and split from common-prefix lengths; equal keys are told apart by index
// `shift` ignores that many low Morton bits: fewer radix levels; the triangles of one (coarser) cell are then split
// by index, which is balanced and, the keys being sorted by the full code, still follows the Morton curve.
__global__ void __launch_bounds__(LB_BLOCK) k_tree(const DOcc* mocc, uint32_t nmocc, const unsigned long long* keys, uint32_t nnodes, uint32_t shift, Tree T)
{
    const uint32_t g = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (g >= nnodes) return;
    const DOcc o = mocc[find_occ<true>(mocc, nmocc, g)];
    const int n = (int)o.n, i = (int)(g - o.node_off);
    T.flag[g] = 0;
    if (i == 0) T.parent_node[g] = 0x7FFFFFFFu;
    if (n <= 2) return;   // one root node over a single leaf: written by k_nodes
    const unsigned long long* K = keys + o.slot_off;
    const unsigned long long ki = K[i] >> shift;
    auto delta = [&](int b) -> int {
        if (b < 0 || b >= n) return -1;
        const unsigned long long kb = K[b] >> shift;
        return ki == kb ? 64 + __clz((uint32_t)(i ^ b)) : __clzll((long long)(ki ^ kb));
    };
    const int d = delta(i + 1) > delta(i - 1) ? 1 : -1;
    const int dmin = delta(i - d);
    int lmax = 2;
    while (delta(i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(j);
    int s = 0;
    for (int t = l;;) {
        t = (t + 1) >> 1;
        if (delta(i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const bool lprim = lo == gamma, rprim = hi == gamma + 1;
    T.left[g] = (lprim ? 0x80000000u : 0u) | (uint32_t)gamma;
    T.right[g] = (rprim ? 0x80000000u : 0u) | (uint32_t)(gamma + 1);
    T.lo[g] = (uint32_t)lo; T.hi[g] = (uint32_t)hi;
    if (lprim) T.parent_prim[o.slot_off + gamma] = g; else T.parent_node[o.node_off + gamma] = g;
    if (rprim) T.parent_prim[o.slot_off + gamma + 1] = g | 0x80000000u; else T.parent_node[o.node_off + gamma + 1] = g | 0x80000000u;
}

__device__ inline float load_coherent(const float* p)
{
    return u2f(__hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Bottom-up fit: every triangle carries its box to its parent; the second arrival at a node merges the two
// child boxes and carries the result further up.
__global__ void __launch_bounds__(LB_BLOCK) k_fit(const rt_triangle* tri, const DOcc* occ, uint32_t nocc, const uint32_t* vals, uint32_t nslots, Tree T)
{
    const uint32_t s = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (s >= nslots) return;
    const DOcc o = occ[find_occ<false>(occ, nocc, s)];
    if (o.node_off == 0xFFFFFFFFu || o.n <= 2) return;
    const rt_triangle t = tri[vals[s]];
    float b[6];
    for (int a = 0; a < 3; ++a) {
        b[a] = fminf(t.v0[a], fminf(t.v1[a], t.v2[a]));
        b[3 + a] = fmaxf(t.v0[a], fmaxf(t.v1[a], t.v2[a]));
    }
    uint32_t p = T.parent_prim[s];
    for (;;) {
        const uint32_t g = p & 0x7FFFFFFFu, side = p >> 31;
        float* mine = T.cbox + ((size_t)g * 2 + side) * 6;
        for (int k = 0; k < 6; ++k) mine[k] = b[k];
        __threadfence();
        if (atomicAdd(&T.flag[g], 1u) == 0u) return;   // the sibling subtree is not done yet: its thread continues
        __threadfence();
        const float* other = T.cbox + ((size_t)g * 2 + (side ^ 1u)) * 6;
        for (int a = 0; a < 3; ++a) {
            b[a] = fminf(b[a], load_coherent(other + a));
            b[3 + a] = fmaxf(b[3 + a], load_coherent(other + 3 + a));
        }
        p = T.parent_node[g];
        if (p == 0x7FFFFFFFu) return;   // g was the root
    }
}

// inner nodes of the final tree on a triangle's path (nodes over <= 2 keys become leaves): the stack bound
__global__ void __launch_bounds__(LB_BLOCK) k_depth(const DOcc* occ, uint32_t nocc, uint32_t nslots, Tree T, uint32_t* max_depth)
{
    const uint32_t s = blockIdx.x * LB_BLOCK + threadIdx.x;
    uint32_t depth = 0;
    if (s < nslots) {
        const DOcc o = occ[find_occ<false>(occ, nocc, s)];
        if (o.node_off != 0xFFFFFFFFu) {
            if (o.n <= 2) depth = 1;
            else
                for (uint32_t p = T.parent_prim[s]; p != 0x7FFFFFFFu; p = T.parent_node[p & 0x7FFFFFFFu]) {
                    const uint32_t g = p & 0x7FFFFFFFu;
                    if (T.hi[g] - T.lo[g] + 1u >= 3u) depth++;
                }
        }
    }
    for (int off = 32; off > 0; off >>= 1) depth = max(depth, (uint32_t)__shfl_xor((int)depth, off));
    if ((threadIdx.x & 63) == 0 && depth) atomicMax(max_depth, depth);
}

// outward 16-bit grid index of a coordinate (HostScene::quantize_nodes)
__device__ inline uint32_t quant(float v, float base, float scale, bool up)
{
    double x = ((double)v - (double)base) / (double)scale;
    x = up ? ceil(x) + 1.0 : floor(x) - 1.0;
    uint32_t q = (uint32_t)fmin(65535.0, fmax(0.0, x));
    if (up) { while (q < 65535u && base + (float)q * scale < v) ++q; }
    else { while (q > 0u && base + (float)q * scale > v) --q; }
    return q;
}

__global__ void __launch_bounds__(LB_BLOCK) k_nodes(const DOcc* mocc, uint32_t nmocc, uint32_t nnodes, Tree T, f3 qbase, f3 qscale, q4* nodes)
{
    const uint32_t g = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (g >= nnodes) return;
    const DOcc o = mocc[find_occ<true>(mocc, nmocc, g)];
    const uint32_t i = g - o.node_off;
    const float base[3] = {qbase.x, qbase.y, qbase.z}, scale[3] = {qscale.x, qscale.y, qscale.z};
    uint32_t q[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ref[2] = {0, 0};
    if (o.n <= 2) {
        // one root node over the leaf; no second child (rt_scene.cpp)
        if (i != 0) { nodes[(size_t)g * 2] = q4{0, 0, 0, 0}; nodes[(size_t)g * 2 + 1] = q4{0, 0, 0, 0}; return; }
        const float ext = 1023.0f;
        for (int a = 0; a < 3; ++a) {
            const float mx = o.inv[a] > 0.0f ? o.mn[a] + ext / o.inv[a] : o.mn[a];   // not used for culling quality: any superset will do
            q[a] = quant(o.mn[a], base[a], scale[a], false);
            q[3 + a] = quant(mx, base[a], scale[a], true);
        }
        ref[0] = REF_LEAF | ((o.n - 1u) << 28) | o.slot_off;
        ref[1] = REF_NONE;
    } else {
        if (T.hi[g] - T.lo[g] + 1u < 3u) { nodes[(size_t)g * 2] = q4{0, 0, 0, 0}; nodes[(size_t)g * 2 + 1] = q4{0, 0, 0, 0}; return; }   // became a leaf
        const uint32_t child[2] = {T.left[g], T.right[g]};
        for (int c = 0; c < 2; ++c) {
            const uint32_t x = child[c] & 0x7FFFFFFFu;
            if (child[c] & 0x80000000u) ref[c] = REF_LEAF | (o.slot_off + x);
            else {
                const uint32_t y = o.node_off + x, len = T.hi[y] - T.lo[y] + 1u;
                ref[c] = len == 2u ? (REF_LEAF | (1u << 28) | (o.slot_off + T.lo[y])) : y;
            }
            const float* b = T.cbox + ((size_t)g * 2 + c) * 6;
            for (int a = 0; a < 3; ++a) {
                q[c * 6 + a] = quant(b[a], base[a], scale[a], false);
                q[c * 6 + 3 + a] = quant(b[3 + a], base[a], scale[a], true);
            }
        }
    }
    uint32_t w[6];
    for (int k = 0; k < 6; ++k) w[k] = q[2 * k] | (q[2 * k + 1] << 16);
    nodes[(size_t)g * 2] = q4{u2f(w[0]), u2f(w[1]), u2f(w[2]), u2f(w[3])};
    nodes[(size_t)g * 2 + 1] = q4{u2f(w[4]), u2f(w[5]), u2f(ref[0]), u2f(ref[1])};
}

// two-wide quantised nodes -> the four-wide nodes the traversal fetches (collapse_node, rt_core.hpp)
__global__ void __launch_bounds__(LB_BLOCK) k_collapse(const q4* nodes2, uint32_t nnodes, uint32_t empty_ref, q4* nodes4)
{
    const uint32_t g = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (g >= nnodes) return;
    q4 out[4];
    collapse_node(nodes2, g, empty_ref, out);
    for (int k = 0; k < 4; ++k) nodes4[(size_t)g * 4 + k] = out[k];
}

struct Scratch {
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t get(T** p, size_t count)
    {
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, std::max<size_t>(16, count * sizeof(T)));
        if (e == hipSuccess) { ptrs.push_back(d); *p = (T*)d; }
        return e;
    }
};

} // namespace

#define LBCHECK(call)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) { err = std::string("lbvh_build: ") + #call + ": " + hipGetErrorString(e_); return RT_ERR_HIP; } \
    } while (0)

int lbvh_build(const HostScene& H, hipStream_t stream, uint32_t max_ok, q4** d_tris, q4** d_nodes, uint32_t* max_depth, double* seconds, std::string& err)
{
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t nslots = H.gpu_slots, nnodes = H.gpu_nodes;
    std::vector<DOcc> all, mesh;
    for (const HostScene::Occurrence& o : H.occurrences) {
        DOcc d{o.first_triangle, o.count, o.rank_base, o.slot_off, o.node_off, {o.mn[0], o.mn[1], o.mn[2]}, {0, 0, 0}};
        for (int a = 0; a < 3; ++a) { const float ext = o.mx[a] - o.mn[a]; d.inv[a] = ext > 0.0f ? 1023.0f / ext : 0.0f; }
        all.push_back(d);
        if (o.node_off != 0xFFFFFFFFu) mesh.push_back(d);
    }
    *d_tris = nullptr; *d_nodes = nullptr; *max_depth = 0;
    // results (owned by the caller on success)
    q4 *tris = nullptr, *nodes = nullptr;
    // one more slot than the layout has: the degenerate triangle that unused node slots refer to (collapse_node)
    LBCHECK(hipMalloc((void**)&tris, ((size_t)nslots + 1) * 3 * sizeof(q4)));
    LBCHECK(hipMemsetAsync(tris + (size_t)nslots * 3, 0, 3 * sizeof(q4), stream));
    if (hipMalloc((void**)&nodes, std::max<size_t>(16, (size_t)nnodes * 4 * sizeof(q4))) != hipSuccess) { (void)hipFree(tris); err = "lbvh_build: out of device memory"; return RT_ERR_HIP; }
    auto fail = [&](const std::string& m) { (void)hipFree(tris); (void)hipFree(nodes); err = m; return (int)RT_ERR_HIP; };
    if (nslots == 0) { *d_tris = tris; *d_nodes = nodes; if (seconds) *seconds = 0.0; return RT_OK; }

    Scratch S;
    rt_triangle* d_tri = nullptr; DOcc *d_all = nullptr, *d_mesh = nullptr;
    unsigned long long *k0 = nullptr, *k1 = nullptr; uint32_t *v0 = nullptr, *v1 = nullptr, *d_depth = nullptr;
    Tree T{};
    q4* nodes2 = nullptr;   // two-wide nodes, collapsed into `nodes` at the end
    hipError_t e = hipSuccess;
    if ((e = S.get(&nodes2, (size_t)nnodes * 2)) != hipSuccess) return fail(std::string("lbvh_build: hipMalloc: ") + hipGetErrorString(e));
    if ((e = S.get(&d_tri, H.triangles.size())) || (e = S.get(&d_all, all.size())) || (e = S.get(&d_mesh, mesh.size())) ||
        (e = S.get(&k0, nslots)) || (e = S.get(&k1, nslots)) || (e = S.get(&v0, nslots)) || (e = S.get(&v1, nslots)) || (e = S.get(&d_depth, 1)) ||
        (e = S.get(&T.left, nnodes)) || (e = S.get(&T.right, nnodes)) || (e = S.get(&T.lo, nnodes)) || (e = S.get(&T.hi, nnodes)) ||
        (e = S.get(&T.parent_node, nnodes)) || (e = S.get(&T.parent_prim, nslots)) || (e = S.get(&T.flag, nnodes)) || (e = S.get(&T.cbox, (size_t)nnodes * 12)))
        return fail(std::string("lbvh_build: hipMalloc: ") + hipGetErrorString(e));
    if ((e = hipMemcpyAsync(d_tri, H.triangles.data(), H.triangles.size() * sizeof(rt_triangle), hipMemcpyHostToDevice, stream)) ||
        (e = hipMemcpyAsync(d_all, all.data(), all.size() * sizeof(DOcc), hipMemcpyHostToDevice, stream)) ||
        (e = hipMemcpyAsync(d_mesh, mesh.data(), mesh.size() * sizeof(DOcc), hipMemcpyHostToDevice, stream)) ||
        (e = hipMemsetAsync(d_depth, 0, 4, stream)))
        return fail(std::string("lbvh_build: upload: ") + hipGetErrorString(e));

    const uint32_t gs = (nslots + LB_BLOCK - 1) / LB_BLOCK, gn = (nnodes + LB_BLOCK - 1) / LB_BLOCK;
    size_t tmp_bytes = 0;
    if ((e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, (size_t)nslots, 0u, 64u, stream)) != hipSuccess)
        return fail(std::string("lbvh_build: radix sort (size query): ") + hipGetErrorString(e));
    char* tmp = nullptr;
    if ((e = S.get(&tmp, tmp_bytes)) != hipSuccess) return fail(std::string("lbvh_build: hipMalloc: ") + hipGetErrorString(e));
    // A radix tree is as deep as its keys are long (plus the index levels of equal keys); the traversal stack
    // allows max_ok inner levels.  Retry with shorter Morton codes before giving up (the caller then falls
    // back to the host builder).
    hipLaunchKernelGGL(k_keys, dim3(gs), dim3(LB_BLOCK), 0, stream, d_tri, d_all, (uint32_t)all.size(), nslots, k0, v0);
    if ((e = rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)nslots, 0u, 64u, stream)) != hipSuccess)
        return fail(std::string("lbvh_build: radix sort: ") + hipGetErrorString(e));
    static const uint32_t shifts[] = {0, 3, 6, 9, 12, 15};
    for (uint32_t attempt = 0; nnodes && attempt < sizeof shifts / sizeof shifts[0]; ++attempt) {
        if ((e = hipMemsetAsync(d_depth, 0, 4, stream)) != hipSuccess) return fail(std::string("lbvh_build: ") + hipGetErrorString(e));
        hipLaunchKernelGGL(k_tree, dim3(gn), dim3(LB_BLOCK), 0, stream, d_mesh, (uint32_t)mesh.size(), k1, nnodes, shifts[attempt], T);
        hipLaunchKernelGGL(k_depth, dim3(gs), dim3(LB_BLOCK), 0, stream, d_all, (uint32_t)all.size(), nslots, T, d_depth);
        if ((e = hipGetLastError()) != hipSuccess) return fail(std::string("lbvh_build: launch: ") + hipGetErrorString(e));
        if ((e = hipMemcpyAsync(max_depth, d_depth, 4, hipMemcpyDeviceToHost, stream)) || (e = hipStreamSynchronize(stream)))
            return fail(std::string("lbvh_build: ") + hipGetErrorString(e));
        if (*max_depth <= max_ok) break;
    }
    hipLaunchKernelGGL(k_records, dim3(gs), dim3(LB_BLOCK), 0, stream, d_tri, d_all, k1, v1, nslots, tris);
    if (nnodes) {
        hipLaunchKernelGGL(k_fit, dim3(gs), dim3(LB_BLOCK), 0, stream, d_tri, d_all, (uint32_t)all.size(), v1, nslots, T);
        hipLaunchKernelGGL(k_nodes, dim3(gn), dim3(LB_BLOCK), 0, stream, d_mesh, (uint32_t)mesh.size(), nnodes, T,
                           F3(H.qbase[0], H.qbase[1], H.qbase[2]), F3(H.qscale[0], H.qscale[1], H.qscale[2]), nodes2);
        hipLaunchKernelGGL(k_collapse, dim3(gn), dim3(LB_BLOCK), 0, stream, nodes2, nnodes, (uint32_t)REF_LEAF | nslots, nodes);
    }
    if ((e = hipGetLastError()) != hipSuccess) return fail(std::string("lbvh_build: launch: ") + hipGetErrorString(e));
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail(std::string("lbvh_build: ") + hipGetErrorString(e));
    *d_tris = tris; *d_nodes = nodes;
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

} // namespace rt
