// rt_scene.cpp — validation, reference-structure rebuild and triangle-BVH construction.
#include "rt_scene.hpp"
#include <atomic>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <numeric>

namespace rt {

namespace {

inline void box_start(Aabb& b, const float* p) { for (int k = 0; k < 3; ++k) b.mn[k] = b.mx[k] = p[k]; }
// BoundingVolumeBox::addVolumePoint (reference main.cpp:197-227): strict comparisons, min/max only
inline void box_add(Aabb& b, const float* p)
{
    for (int k = 0; k < 3; ++k) {
        if (p[k] < b.mn[k]) b.mn[k] = p[k];
        if (p[k] > b.mx[k]) b.mx[k] = p[k];
    }
}
inline void box_merge(Aabb& b, const Aabb& o) { box_add(b, o.mn); box_add(b, o.mx); }
// BoundingVolumeBox::updateMinMaxCentroid (main.cpp:240-242)
inline void centroid(const Aabb& b, float* c) { for (int k = 0; k < 3; ++k) c[k] = (b.mn[k] + b.mx[k]) / 2.0f; }

} // namespace

int HostScene::validate()
{
    auto bad = [&](const char* m) { error = m; return (int)RT_ERR_INVALID; };
    const uint32_t nm = (uint32_t)materials.size();
    for (auto& m : materials) {
        if (m.kind < RT_MAT_MATTE || m.kind > RT_MAT_REFRACTION) return bad("material: unknown kind");
        if (m.kind == RT_MAT_TEXTURED && m.texture >= (int32_t)textures.size()) return bad("material: texture index out of range");
    }
    auto mat_ok = [&](int32_t m) { return m >= 0 && (uint32_t)m < nm; };
    for (auto& p : planes) {
        if (!mat_ok(p.material)) return bad("plane: material index out of range");
        if (materials[p.material].kind == RT_MAT_TEXTURED) { error = "Textured material on a Plane reads a stale uv in the reference (SURVEY Q16): unsupported"; return RT_ERR_UNSUPPORTED; }
    }
    for (auto& s : spheres) {
        if (!mat_ok(s.material)) return bad("sphere: material index out of range");
        if (materials[s.material].kind == RT_MAT_TEXTURED) { error = "Textured material on a Sphere reads a stale uv in the reference (SURVEY Q16): unsupported"; return RT_ERR_UNSUPPORTED; }
    }
    for (auto& t : triangles) if (!mat_ok(t.material)) return bad("triangle: material index out of range");
    std::vector<uint8_t> tri_in_mesh(triangles.size(), 0);
    for (auto& m : meshes) {
        if (m.num_triangles == 0) return bad("mesh: empty");
        if ((uint64_t)m.first_triangle + m.num_triangles > triangles.size()) return bad("mesh: triangle run out of range");
        for (uint32_t k = 0; k < m.num_triangles; ++k) tri_in_mesh[m.first_triangle + k] = 1;
    }
    std::vector<uint8_t> mesh_in_mm(meshes.size(), 0);
    for (auto& mm : multimeshes) {
        if (mm.num_meshes == 0) return bad("multimesh: empty");
        if ((uint64_t)mm.first_mesh + mm.num_meshes > meshes.size()) return bad("multimesh: mesh run out of range");
        for (uint32_t k = 0; k < mm.num_meshes; ++k) mesh_in_mm[mm.first_mesh + k] = 1;
    }
    auto ref_ok = [&](const rt_shape_ref& r, bool in_bvh) -> int {
        switch (r.kind) {
        case RT_SHAPE_PLANE: if (in_bvh) { error = "Plane has no getBoundingBoxPoints(): cannot be a BVHAccel leaf"; return RT_ERR_INVALID; }
            return r.index < planes.size() ? RT_OK : bad("shape ref: plane index");
        case RT_SHAPE_SPHERE: return r.index < spheres.size() ? RT_OK : bad("shape ref: sphere index");
        case RT_SHAPE_TRIANGLE:
            if (r.index >= triangles.size()) return bad("shape ref: triangle index");
            if (tri_in_mesh[r.index]) return bad("shape ref: triangle belongs to a mesh");
            return RT_OK;
        case RT_SHAPE_MESH:
            if (r.index >= meshes.size()) return bad("shape ref: mesh index");
            if (mesh_in_mm[r.index]) return bad("shape ref: mesh belongs to a multimesh");
            return RT_OK;
        case RT_SHAPE_MULTIMESH: return r.index < multimeshes.size() ? RT_OK : bad("shape ref: multimesh index");
        case RT_SHAPE_BVH:
            if (in_bvh) { error = "nested BVHAccel is not supported"; return RT_ERR_UNSUPPORTED; }
            return r.index < bvhs.size() ? RT_OK : bad("shape ref: bvh index");
        }
        return bad("shape ref: unknown kind");
    };
    for (auto& b : bvhs) {
        if (b.num_leaves == 0) return bad("bvh: no leaves (the reference indexes mHeirarchy[size-1])");
        if ((uint64_t)b.first_leaf + b.num_leaves > bvh_leaves.size()) return bad("bvh: leaf run out of range");
        for (uint32_t k = 0; k < b.num_leaves; ++k) { int rc = ref_ok(bvh_leaves[b.first_leaf + k], true); if (rc) return rc; }
    }
    for (auto& r : scene) { int rc = ref_ok(r, false); if (rc) return rc; }
    for (auto& l : lights) if (l.kind != RT_LIGHT_POINT && l.kind != RT_LIGHT_DIRECTIONAL) return bad("light: unknown kind");
    if (ambient.occlusion_samples < 0 || ambient.occlusion_samples > 64) return bad("ambient: occlusion_samples out of range [0,64]");
    if (triangles.size() >= (1u << 28) - 1u) { error = "more than 2^28 - 2 triangles"; return RT_ERR_UNSUPPORTED; }   // one slot is the builders' own
    return RT_OK;
}

int HostScene::shape_box(const rt_shape_ref& r, Aabb& out)
{
    switch (r.kind) {
    case RT_SHAPE_SPHERE: {   // Sphere::getBoundingBoxPoints (main.cpp:571-576)
        const rt_sphere& s = spheres[r.index];
        float hi[3], lo[3];
        for (int k = 0; k < 3; ++k) { hi[k] = s.centre[k] + s.radius; lo[k] = s.centre[k] - s.radius; }
        box_start(out, hi); box_add(out, lo);
        return RT_OK;
    }
    case RT_SHAPE_TRIANGLE: { // Triangle::getBoundingBoxPoints (main.cpp:705-711)
        const rt_triangle& t = triangles[r.index];
        box_start(out, t.v0); box_add(out, t.v1); box_add(out, t.v2);
        return RT_OK;
    }
    case RT_SHAPE_MESH: out = mesh_box_[r.index]; return RT_OK;
    case RT_SHAPE_MULTIMESH: out = mm_box_[r.index]; return RT_OK;
    }
    error = "shape has no bounding box";
    return RT_ERR_INVALID;
}

uint32_t HostScene::prim_count(const rt_shape_ref& r)
{
    switch (r.kind) {
    case RT_SHAPE_MESH: return meshes[r.index].num_triangles;
    case RT_SHAPE_MULTIMESH: {
        uint32_t n = 0;
        for (uint32_t k = 0; k < multimeshes[r.index].num_meshes; ++k) n += meshes[multimeshes[r.index].first_mesh + k].num_triangles;
        return n;
    }
    case RT_SHAPE_BVH: {
        uint32_t n = 0;
        for (uint32_t k = 0; k < bvhs[r.index].num_leaves; ++k) n += prim_count(bvh_leaves[bvhs[r.index].first_leaf + k]);
        return n;
    }
    default: return 1;
    }
}

// BVHAccel::addShape + generateBVH (main.cpp:323-423): layer by layer, repeatedly join the two
// not-yet-joined nodes of the layer whose box centres are closest (scanning from the highest
// index down, first strict minimum wins); a left-over node is copied up unchanged.
int HostScene::reference_hierarchy(const rt_bvh& b, std::vector<RefNode>& H)
{
    H.clear();
    for (uint32_t k = 0; k < b.num_leaves; ++k) {
        RefNode n{};
        n.leaf = true; n.shape = bvh_leaves[b.first_leaf + k]; n.accessed = false;
        int rc = shape_box(n.shape, n.box);
        if (rc) return rc;
        H.push_back(n);
    }
    size_t layer = H.size();
    while (layer != 1) {
        const size_t next_layer = (size_t)std::ceil((float)layer / 2.0f);
        const int end = (int)H.size(), begin = (int)(H.size() - layer);
        auto join = [&](int bi, int bk) {
            RefNode n{};
            n.leaf = false; n.left = (uint32_t)bi; n.right = (uint32_t)bk; n.accessed = false;
            n.box = H[bi].box; box_merge(n.box, H[bk].box);
            H.push_back(n);
            H[bi].accessed = true; H[bk].accessed = true;
        };
        auto copy_up = [&] {
            for (int i = end - 1; i >= begin; --i)
                if (!H[i].accessed) { RefNode c = H[i]; H.push_back(c); H[i].accessed = true; break; }
        };
        size_t made = 0;
        if (layer <= 8192 && !force_reference_scan) {
            // Same result as the reference's scan (below) in O(n^2 log n): distances inside a layer never
            // change, so "closest remaining pair, highest i then highest k first among equals" is the order
            // of all pairs sorted by (distance, -i, -k), skipping pairs with a used end.
            struct Pair { float dist; int i, k; };
            std::vector<Pair> pairs;
            pairs.reserve(layer * (layer - 1) / 2);
            std::vector<float> cen(layer * 3);
            for (int i = begin; i < end; ++i) centroid(H[i].box, &cen[(size_t)(i - begin) * 3]);
            for (int i = end - 1; i > begin; --i) {
                const float* ci = &cen[(size_t)(i - begin) * 3];
                for (int k = i - 1; k >= begin; --k) {
                    const float* ck = &cen[(size_t)(k - begin) * 3];
                    float dx = ck[0] - ci[0], dy = ck[1] - ci[1], dz = ck[2] - ci[2];
                    float dist = sqrtf(dx * dx + dy * dy + dz * dz);   // glm::distance = length(p1 - p0)
                    if (dist < std::numeric_limits<float>::max()) pairs.push_back(Pair{dist, i, k});
                }
            }
            auto before = [](const Pair& a, const Pair& b) {
                if (a.dist != b.dist) return a.dist < b.dist;
                if (a.i != b.i) return a.i > b.i;
                return a.k > b.k;
            };
            // most nodes find their partner among the few closest pairs: sort the list in growing
            // chunks (selection + sort of the chunk) instead of all of it
            size_t unpaired = layer;
            for (size_t done = 0, chunk = 8 * layer; done < pairs.size() && made < next_layer && unpaired > 1; chunk *= 4) {
                const size_t end_c = std::min(pairs.size(), done + chunk);
                if (end_c < pairs.size()) std::nth_element(pairs.begin() + done, pairs.begin() + end_c, pairs.end(), before);
                std::sort(pairs.begin() + done, pairs.begin() + end_c, before);
                for (size_t x = done; x < end_c && made < next_layer; ++x) {
                    const Pair& p = pairs[x];
                    if (H[p.i].accessed || H[p.k].accessed) continue;
                    join(p.i, p.k); ++made; unpaired -= 2;
                }
                done = end_c;
            }
            for (; made < next_layer; ++made) copy_up();
        }
        for (; made < next_layer; ++made) {   // the reference's own O(n^3) scan, kept for very wide layers
            int bi = -1, bk = -1;
            float best = std::numeric_limits<float>::max();
            for (int i = end - 1; i > begin; --i) {
                if (H[i].accessed) continue;
                float ci[3]; centroid(H[i].box, ci);
                for (int k = i - 1; k >= begin; --k) {
                    if (H[k].accessed) continue;
                    float ck[3]; centroid(H[k].box, ck);
                    float dx = ck[0] - ci[0], dy = ck[1] - ci[1], dz = ck[2] - ci[2];
                    float dist = sqrtf(dx * dx + dy * dy + dz * dz);   // glm::distance = length(p1 - p0)
                    if (dist < best) { best = dist; bi = i; bk = k; }
                }
            }
            if (bi < 0) copy_up();
            else join(bi, bk);
        }
        layer = next_layer;
    }
    return RT_OK;
}

void HostScene::write_triangle_slot(uint32_t slot, uint32_t gid, uint32_t rank)
{
    const rt_triangle& t = triangles[gid];
    tris4[(size_t)slot * 3 + 0] = {t.v0[0], t.v0[1], t.v0[2], u2f(gid)};
    tris4[(size_t)slot * 3 + 1] = {t.v1[0] - t.v0[0], t.v1[1] - t.v0[1], t.v1[2] - t.v0[2], u2f((uint32_t)t.material)};
    tris4[(size_t)slot * 3 + 2] = {t.v2[0] - t.v0[0], t.v2[1] - t.v0[1], t.v2[2] - t.v0[2], u2f(rank)};
}

uint32_t HostScene::add_triangle_slot(uint32_t gid, uint32_t rank)
{
    if (gpu_bvh) {   // a free triangle: one slot, no node
        Occurrence o{gid, 1u, rank, gpu_slots, 0xFFFFFFFFu, {0, 0, 0}, {0, 0, 0}};
        occurrences.push_back(o);
        return gpu_slots++;
    }
    uint32_t slot = (uint32_t)(tris4.size() / 3);
    tris4.resize(tris4.size() + 3);
    write_triangle_slot(slot, gid, rank);
    return slot;
}

namespace {
// fn(begin, end) over [0, n) on up to 16 threads (the calling thread included)
template <class F>
void parallel_ranges(size_t n, size_t min_chunk, F fn)
{
    const size_t hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const size_t nt = std::max<size_t>(1, std::min(hw, n / std::max<size_t>(1, min_chunk)));
    if (nt <= 1) { fn((size_t)0, n); return; }
    std::vector<std::thread> pool;
    for (size_t t = 1; t < nt; ++t) pool.emplace_back([=] { fn(n * t / nt, n * (t + 1) / nt); });
    fn((size_t)0, n / nt);
    for (auto& t : pool) t.join();
}
} // namespace

// ---- triangle BVH of one Mesh: binned SAH, <= 2 triangles per leaf, two-child nodes (built as 64-byte
//      float nodes, shipped as 32-byte quantised nodes: quantize_nodes) ----
namespace {
struct BuildTri { float mn[3], mx[3], c[3]; uint32_t gid; };
// triangles per leaf: 2 (best of 1..8 measured on config 3: fewer exact tests outweigh the extra nodes); RT_LEAF_MAX overrides it for such measurements
static const uint32_t kLeafMax = [] { const char* e = getenv("RT_LEAF_MAX"); int v = e ? atoi(e) : 2; return (uint32_t)(v < 1 ? 1 : v > 8 ? 8 : v); }();
constexpr int kMaxDepth = 24;   // inner-node depth bound of the two-wide tree: 25 levels = 13 four-wide levels (RT_STACK_DEPTH, rt_core.hpp)
constexpr int kMaxBins = 64;
// SAH bins per axis: RT_SAH_BINS overrides the default for measurements (8..64)
static const int kBins = [] { const char* e = getenv("RT_SAH_BINS"); int v = e ? atoi(e) : 16; return v < 2 ? 2 : v > kMaxBins ? kMaxBins : v; }();

inline float half_area(const float* mn, const float* mx)
{
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return dx * dy + dy * dz + dz * dx;
}
} // namespace

// The layout after appending parts (rt_scene.hpp: BvhPart) is that of a sequential depth-first build:
// [node][left subtree][right subtree], triangle slots in leaf order.
namespace {
uint32_t relocate(uint32_t ref, uint32_t node_off, uint32_t slot_off)
{
    if (ref == REF_NONE) return ref;
    return (ref & REF_LEAF) ? ref + slot_off : ref + node_off;   // slots sit in the low 28 bits and do not overflow them (checked by validate)
}
void append_part(HostScene::BvhPart& dst, const HostScene::BvhPart& src, uint32_t& ref)
{
    const uint32_t node_off = (uint32_t)(dst.nodes.size() / 4), slot_off = (uint32_t)dst.order.size();
    for (size_t i = 0; i + 3 < src.nodes.size(); i += 4) {
        dst.nodes.push_back(src.nodes[i]); dst.nodes.push_back(src.nodes[i + 1]); dst.nodes.push_back(src.nodes[i + 2]);
        q4 r = src.nodes[i + 3];
        r.x = u2f(relocate(f2u(r.x), node_off, slot_off)); r.y = u2f(relocate(f2u(r.y), node_off, slot_off));
        dst.nodes.push_back(r);
    }
    dst.order.insert(dst.order.end(), src.order.begin(), src.order.end());
    dst.max_depth = std::max(dst.max_depth, src.max_depth);
    ref = relocate(ref, node_off, slot_off);
}

// returns the (local) child reference for T[lo,hi); `par` = levels of the tree that may still fork threads
uint32_t build_subtree(std::vector<BuildTri>& T, uint32_t lo, uint32_t hi, int depth, int par, HostScene::BvhPart& out)
{
    auto bounds = [&](uint32_t a0, uint32_t a1, float* mn, float* mx) {
        for (int a = 0; a < 3; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; }
        for (uint32_t i = a0; i < a1; ++i)
            for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], T[i].mn[a]); mx[a] = std::max(mx[a], T[i].mx[a]); }
    };
    const uint32_t n = hi - lo;
    if (n <= kLeafMax) {
        const uint32_t first = (uint32_t)out.order.size();
        for (uint32_t i = lo; i < hi; ++i) out.order.push_back(T[i].gid);
        return REF_LEAF | ((n - 1u) << 28) | first;
    }
    out.max_depth = std::max<uint32_t>(out.max_depth, (uint32_t)depth + 1);
    // centroid bounds
    float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = lo; i < hi; ++i)
        for (int a = 0; a < 3; ++a) { cmn[a] = std::min(cmn[a], T[i].c[a]); cmx[a] = std::max(cmx[a], T[i].c[a]); }
    int best_axis = -1, best_bin = -1;
    float best_cost = INFINITY;
    for (int a = 0; a < 3; ++a) {
        float ext = cmx[a] - cmn[a];
        if (!(ext > 0.0f)) continue;
        uint32_t cnt[kMaxBins] = {0};
        float bmn[kMaxBins][3], bmx[kMaxBins][3];
        for (int b = 0; b < kBins; ++b) for (int k = 0; k < 3; ++k) { bmn[b][k] = INFINITY; bmx[b][k] = -INFINITY; }
        float scale = (float)kBins / ext;
        for (uint32_t i = lo; i < hi; ++i) {
            int b = std::min(kBins - 1, std::max(0, (int)((T[i].c[a] - cmn[a]) * scale)));
            cnt[b]++;
            for (int k = 0; k < 3; ++k) { bmn[b][k] = std::min(bmn[b][k], T[i].mn[k]); bmx[b][k] = std::max(bmx[b][k], T[i].mx[k]); }
        }
        float rarea[kMaxBins]; uint32_t rcnt[kMaxBins];
        float amn[3] = {INFINITY, INFINITY, INFINITY}, amx[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t c = 0;
        for (int b = kBins - 1; b > 0; --b) {
            c += cnt[b];
            for (int k = 0; k < 3; ++k) { amn[k] = std::min(amn[k], bmn[b][k]); amx[k] = std::max(amx[k], bmx[b][k]); }
            rcnt[b] = c; rarea[b] = c ? half_area(amn, amx) : 0.0f;
        }
        for (int k = 0; k < 3; ++k) { amn[k] = INFINITY; amx[k] = -INFINITY; }
        c = 0;
        for (int b = 0; b < kBins - 1; ++b) {
            c += cnt[b];
            for (int k = 0; k < 3; ++k) { amn[k] = std::min(amn[k], bmn[b][k]); amx[k] = std::max(amx[k], bmx[b][k]); }
            if (c == 0 || rcnt[b + 1] == 0) continue;
            float cost = half_area(amn, amx) * (float)c + rarea[b + 1] * (float)rcnt[b + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
        }
    }
    uint32_t mid = lo;
    const uint64_t cap_child = (uint64_t)kLeafMax << std::max(0, kMaxDepth - depth - 1);
    if (best_axis >= 0) {
        float ext = cmx[best_axis] - cmn[best_axis];
        float scale = (float)kBins / ext;
        auto it = std::partition(T.begin() + lo, T.begin() + hi, [&](const BuildTri& t) {
            int b = std::min(kBins - 1, std::max(0, (int)((t.c[best_axis] - cmn[best_axis]) * scale)));
            return b <= best_bin;
        });
        mid = (uint32_t)(it - T.begin());
    }
    if (mid == lo || mid == hi || (uint64_t)(mid - lo) > cap_child || (uint64_t)(hi - mid) > cap_child) {
        // median split along the widest centroid axis (also the depth-bounding fallback)
        int a = 0;
        if (cmx[1] - cmn[1] > cmx[a] - cmn[a]) a = 1;
        if (cmx[2] - cmn[2] > cmx[a] - cmn[a]) a = 2;
        mid = lo + n / 2;
        std::nth_element(T.begin() + lo, T.begin() + mid, T.begin() + hi,
                         [&](const BuildTri& x, const BuildTri& y) { return x.c[a] < y.c[a]; });
    }
    const size_t me = out.nodes.size() / 4;
    out.nodes.resize(out.nodes.size() + 4);
    float lmn[3], lmx[3], rmn[3], rmx[3];
    bounds(lo, mid, lmn, lmx); bounds(mid, hi, rmn, rmx);
    uint32_t l, r;
    if (par > 0 && n >= 32768) {   // the two halves touch disjoint ranges of T: build them side by side
        HostScene::BvhPart lp, rp;
        std::thread th([&] { l = build_subtree(T, lo, mid, depth + 1, par - 1, lp); });
        r = build_subtree(T, mid, hi, depth + 1, par - 1, rp);
        th.join();
        append_part(out, lp, l);
        append_part(out, rp, r);
    } else {
        l = build_subtree(T, lo, mid, depth + 1, 0, out);
        r = build_subtree(T, mid, hi, depth + 1, 0, out);
    }
    out.nodes[me * 4 + 0] = {lmn[0], lmn[1], lmn[2], lmx[0]};
    out.nodes[me * 4 + 1] = {lmx[1], lmx[2], rmn[0], rmn[1]};
    out.nodes[me * 4 + 2] = {rmn[2], rmx[0], rmx[1], rmx[2]};
    out.nodes[me * 4 + 3] = {u2f(l), u2f(r), 0.0f, 0.0f};
    return (uint32_t)me;
}
} // namespace

// builds the part of one mesh (no rank, no global indices: reusable for every place the mesh is reached from)
void HostScene::build_mesh_part(uint32_t mesh, int par, BvhPart& part) const
{
    const rt_mesh& m = meshes[mesh];
    std::vector<BuildTri> T(m.num_triangles);
    for (uint32_t k = 0; k < m.num_triangles; ++k) {
        const rt_triangle& t = triangles[m.first_triangle + k];
        BuildTri& b = T[k];
        b.gid = m.first_triangle + k;
        for (int a = 0; a < 3; ++a) {
            b.mn[a] = std::min(t.v0[a], std::min(t.v1[a], t.v2[a]));
            b.mx[a] = std::max(t.v0[a], std::max(t.v1[a], t.v2[a]));
            b.c[a] = 0.5f * (b.mn[a] + b.mx[a]);
        }
    }
    if (m.num_triangles <= kLeafMax) {   // a single leaf hangs below a one-child root node
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (auto& b : T) for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); }
        for (auto& b : T) part.order.push_back(b.gid);
        const uint32_t leaf = REF_LEAF | ((m.num_triangles - 1u) << 28) | 0u;
        part.nodes.push_back({mn[0], mn[1], mn[2], mx[0]});
        // no second child: an inverted box and REF_NONE (collapse_node leaves the slots empty)
        part.nodes.push_back({mx[1], mx[2], INFINITY, INFINITY});
        part.nodes.push_back({INFINITY, -INFINITY, -INFINITY, -INFINITY});
        part.nodes.push_back({u2f(leaf), u2f((uint32_t)REF_NONE), 0.0f, 0.0f});
        part.root = 0;
    } else {
        part.root = build_subtree(T, 0, m.num_triangles, 0, par, part);
    }
}

// All mesh BVHs, built ahead of the emission pass on up to 16 threads: meshes side by side when there are
// many (config 5: 2500 instances), the top levels of each tree when there are few (config 3: one mesh).
void HostScene::prebuild_mesh_parts()
{
    mesh_part_.clear();
    mesh_part_.resize(meshes.size());
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const int par = meshes.size() >= hw ? 0 : (hw >= 16 ? 4 : hw >= 8 ? 3 : hw >= 4 ? 2 : hw >= 2 ? 1 : 0);
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t i; (i = next.fetch_add(1)) < meshes.size();) build_mesh_part((uint32_t)i, par, mesh_part_[i]);
    };
    const unsigned nthreads = (unsigned)std::min<size_t>(meshes.size(), par > 0 ? 1 : hw);
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
}

uint32_t HostScene::mesh_bvh(uint32_t mesh, uint32_t rank_base)
{
    // triangle records carry their absolute rank, so a Mesh reachable through two gate nodes gets two BVH copies
    if (mesh_root_[mesh] != 0xFFFFFFFFu && mesh_rank_[mesh] == rank_base) return mesh_root_[mesh];
    const rt_mesh& m = meshes[mesh];
    if (gpu_bvh) {
        Occurrence o{m.first_triangle, m.num_triangles, rank_base, gpu_slots, gpu_nodes, {0, 0, 0}, {0, 0, 0}};
        std::memcpy(o.mn, mesh_box_[mesh].mn, 12); std::memcpy(o.mx, mesh_box_[mesh].mx, 12);
        occurrences.push_back(o);
        gpu_slots += m.num_triangles;
        gpu_nodes += std::max(1u, m.num_triangles - 1u);   // a radix tree over n keys has n - 1 inner nodes; <= 2 triangles: one root node
        mesh_root_[mesh] = o.node_off; mesh_rank_[mesh] = rank_base;
        return o.node_off;
    }
    const BvhPart& part = mesh_part_[mesh];
    // reserve this occurrence's node and slot ranges now (the layout is that of a sequential build); the
    // copies themselves run on all threads after the emission pass (run_append_jobs)
    const uint32_t node_off = (uint32_t)(fnodes4.size() / 4), slot_off = (uint32_t)(tris4.size() / 3);
    fnodes4.resize(fnodes4.size() + part.nodes.size());
    tris4.resize(tris4.size() + 3 * part.order.size());
    append_jobs_.push_back(AppendJob{mesh, rank_base, node_off, slot_off});
    const uint32_t root = relocate(part.root, node_off, slot_off);
    bvh_max_depth = std::max(bvh_max_depth, part.max_depth);
    mesh_root_[mesh] = root; mesh_rank_[mesh] = rank_base;
    return root;
}

void HostScene::run_append_jobs()
{
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t j; (j = next.fetch_add(1)) < append_jobs_.size();) {
            const AppendJob& job = append_jobs_[j];
            const BvhPart& part = mesh_part_[job.mesh];
            const uint32_t first = meshes[job.mesh].first_triangle;
            q4* dst = &fnodes4[(size_t)job.node_off * 4];
            for (size_t i = 0; i + 3 < part.nodes.size(); i += 4) {
                dst[i] = part.nodes[i]; dst[i + 1] = part.nodes[i + 1]; dst[i + 2] = part.nodes[i + 2];
                q4 r = part.nodes[i + 3];
                r.x = u2f(relocate(f2u(r.x), job.node_off, job.slot_off)); r.y = u2f(relocate(f2u(r.y), job.node_off, job.slot_off));
                dst[i + 3] = r;
            }
            for (size_t k = 0; k < part.order.size(); ++k)
                write_triangle_slot(job.slot_off + (uint32_t)k, part.order[k], job.rank_base + (part.order[k] - first));
        }
    };
    // one big mesh: split by hand would be needed to go faster; many meshes: one job each
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < std::min<size_t>(hw, append_jobs_.size()); ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    append_jobs_.clear();
}

// 64-byte float nodes -> 32-byte two-wide nodes -> 64-byte four-wide nodes (collapse_node, rt_core.hpp).
// The 12 box coordinates become 16-bit grid indices in ONE frame
// for the whole scene (qbase + q * qscale), rounded outwards by a full grid step beyond floor/ceil, so
// the decoded box always contains the float box (checked below with the decode arithmetic of
// cull_box's callers); the traversal's padded slab test then works on a superset, which culling allows.
void HostScene::quantize_nodes()
{
    const size_t n = fnodes4.size() / 4;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto box_of = [&](size_t i, int child, float* lo, float* hi) {
        const float* f = &fnodes4[i * 4].x;
        for (int a = 0; a < 3; ++a) { lo[a] = f[child * 6 + a]; hi[a] = f[child * 6 + 3 + a]; }
    };
    std::mutex mu;
    parallel_ranges(n, 1 << 16, [&](size_t b, size_t e) {
        float tmn[3] = {INFINITY, INFINITY, INFINITY}, tmx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = b; i < e; ++i)
            for (int c = 0; c < 2; ++c) {
                float lo[3], hi[3]; box_of(i, c, lo, hi);
                if (!(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2])) continue;   // the inverted box of a one-leaf root
                for (int a = 0; a < 3; ++a) { tmn[a] = std::min(tmn[a], lo[a]); tmx[a] = std::max(tmx[a], hi[a]); }
            }
        std::lock_guard<std::mutex> lk(mu);
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], tmn[a]); mx[a] = std::max(mx[a], tmx[a]); }
    });
    for (int a = 0; a < 3; ++a) {
        if (!(mn[a] <= mx[a])) { mn[a] = 0.0f; mx[a] = 1.0f; }
        qbase[a] = mn[a];
        qscale[a] = std::max((mx[a] - mn[a]) / 65533.0f, 1e-30f);
    }
    std::vector<q4> nodes2(n * 2, q4{0, 0, 0, 0});   // quantised two-wide nodes: the input of collapse_node()
    parallel_ranges(n, 1 << 16, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
        uint32_t q[12] = {0};
        for (int c = 0; c < 2; ++c) {
            float lo[3], hi[3]; box_of(i, c, lo, hi);
            for (int a = 0; a < 3; ++a) {
                if (!(lo[a] <= hi[a])) { q[c * 6 + a] = 0; q[c * 6 + 3 + a] = 0; continue; }   // the dummy child: a point box at the grid origin
                double l = std::floor(((double)lo[a] - (double)qbase[a]) / (double)qscale[a]) - 1.0;
                double h = std::ceil(((double)hi[a] - (double)qbase[a]) / (double)qscale[a]) + 1.0;
                uint32_t ql = (uint32_t)std::min(65535.0, std::max(0.0, l)), qh = (uint32_t)std::min(65535.0, std::max(0.0, h));
                // the decode the kernels' arithmetic implies: base + q*scale in float
                while (ql > 0 && qbase[a] + (float)ql * qscale[a] > lo[a]) --ql;
                while (qh < 65535 && qbase[a] + (float)qh * qscale[a] < hi[a]) ++qh;
                q[c * 6 + a] = ql; q[c * 6 + 3 + a] = qh;
            }
        }
        uint32_t w[6];
        for (int k = 0; k < 6; ++k) w[k] = q[2 * k] | (q[2 * k + 1] << 16);
        nodes2[i * 2 + 0] = {u2f(w[0]), u2f(w[1]), u2f(w[2]), u2f(w[3])};
        nodes2[i * 2 + 1] = {u2f(w[4]), u2f(w[5]), fnodes4[i * 4 + 3].x, fnodes4[i * 4 + 3].y};
    }
    });
    // four-wide nodes, same indices (a Mesh's root keeps its index; nodes on odd levels go unused); their unused
    // slots refer to one degenerate triangle appended behind the last slot (collapse_node, rt_core.hpp)
    const uint32_t null_slot = (uint32_t)(tris4.size() / 3);
    tris4.resize(tris4.size() + 3, q4{0, 0, 0, 0});
    const uint32_t empty_ref = REF_LEAF | null_slot;
    nodes4.assign(n * 4, q4{0, 0, 0, 0});
    parallel_ranges(n, 1 << 16, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) collapse_node(nodes2.data(), (uint32_t)i, empty_ref, &nodes4[i * 4]);
    });
}

// The 16-byte copies of the gate nodes (QGate, rt_core.hpp): boxes on a 16-bit grid over the union of the gate boxes,
// rounded outwards and checked with the kernels' own decode (gate_dq).
void HostScene::build_qgates()
{
    qgates.clear();
    if (gates.size() > RT_QGATE_MAX || gates.size() >= (1u << 24)) return;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto boxed = [](const GateNode& g) { const uint32_t k = g.kind & G_KIND_MASK; return (k == G_GATED || k == G_MESH) && !(g.kind & G_SAME_BOX); };
    for (const GateNode& g : gates) {
        if (!boxed(g)) continue;
        for (int a = 0; a < 3; ++a) {
            if (!(std::isfinite(g.mn[a]) && std::isfinite(g.mx[a]) && g.mn[a] <= g.mx[a])) return;   // not a box a grid can hold: global walk
            mn[a] = std::min(mn[a], g.mn[a]); mx[a] = std::max(mx[a], g.mx[a]);
        }
    }
    for (int a = 0; a < 3; ++a) {
        if (!(mn[a] <= mx[a])) { mn[a] = 0.0f; mx[a] = 1.0f; }
        gq_base[a] = mn[a];
        gq_scale[a] = std::max((mx[a] - mn[a]) / 65533.0f, 1e-30f);
    }
    qgates.resize(gates.size());
    for (size_t i = 0; i < gates.size(); ++i) {
        const GateNode& g = gates[i];
        const uint32_t kind = g.kind & G_KIND_MASK;
        QGate q{0, 0, 0, 0};
        q.w = (g.skip & 0xFFFFFFu) | (kind << 24) | ((g.kind & G_SAME_BOX) ? (1u << 27) : 0u);
        if (boxed(g)) {
            uint32_t w[3];
            for (int a = 0; a < 3; ++a) {
                const double l = std::floor(((double)g.mn[a] - (double)gq_base[a]) / (double)gq_scale[a]) - 1.0;
                const double h = std::ceil(((double)g.mx[a] - (double)gq_base[a]) / (double)gq_scale[a]) + 1.0;
                uint32_t ql = (uint32_t)std::min(65535.0, std::max(0.0, l)), qh = (uint32_t)std::min(65535.0, std::max(0.0, h));
                while (ql > 0 && gate_dq(ql, gq_scale[a], gq_base[a]) > g.mn[a]) --ql;
                while (qh < 65535 && gate_dq(qh, gq_scale[a], gq_base[a]) < g.mx[a]) ++qh;
                if (gate_dq(ql, gq_scale[a], gq_base[a]) > g.mn[a] || gate_dq(qh, gq_scale[a], gq_base[a]) < g.mx[a]) { qgates.clear(); return; }   // cannot enclose it
                w[a] = ql | (qh << 16);
            }
            q.x = w[0]; q.y = w[1]; q.z = w[2];
        } else if (kind == G_MESH) {
            q.x = g.a;   // G_SAME_BOX: entered without a test, only the root is needed
        }
        qgates[i] = q;
    }
}

void HostScene::emit_mesh(uint32_t mesh, uint32_t item, uint32_t rank_base)
{
    GateNode g{};
    std::memcpy(g.mn, mesh_box_[mesh].mn, 12); std::memcpy(g.mx, mesh_box_[mesh].mx, 12);
    g.kind = G_MESH; g.a = mesh_bvh(mesh, rank_base); g.rank = rank_base; g.prim = meshes[mesh].first_triangle; g.item = item;
    g.skip = g.a;   // a Mesh node is a leaf (its successor is the next node): the field carries the root a second time, inside the 32 bytes the per-lane walk fetches first (gate_step)
    gates.push_back(g);
}

int HostScene::emit_bvh(uint32_t bvh, uint32_t item, uint32_t rank_base)
{
    std::vector<RefNode> H;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = reference_hierarchy(bvhs[bvh], H);
    hierarchy_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc) return rc;
    const uint32_t root = (uint32_t)H.size() - 1;
    // ranks follow the reference's breadth-first visiting order (BVHAccel::hit, main.cpp:425-470)
    std::vector<uint32_t> rank(H.size(), 0);
    {
        std::vector<uint32_t> q{root};
        uint32_t r = rank_base;
        for (size_t h = 0; h < q.size(); ++h) {
            const RefNode& n = H[q[h]];
            if (n.leaf) { rank[q[h]] = r; r += prim_count(n.shape); }
            else { q.push_back(n.left); q.push_back(n.right); }
        }
    }
    std::function<void(uint32_t)> emit = [&](uint32_t i) {
        const RefNode& n = H[i];
        if (n.leaf) { emit_shape(n.shape, item, rank[i]); return; }
        size_t me = gates.size();
        GateNode g{};
        std::memcpy(g.mn, n.box.mn, 12); std::memcpy(g.mx, n.box.mx, 12);
        g.kind = G_GATED; g.item = item;
        gates.push_back(g);
        emit(n.left); emit(n.right);
        gates[me].skip = (uint32_t)gates.size();
    };
    emit(root);
    return RT_OK;
}

void HostScene::emit_shape(const rt_shape_ref& r, uint32_t item, uint32_t rank_base)
{
    GateNode g{};
    g.item = item; g.rank = rank_base;
    switch (r.kind) {
    case RT_SHAPE_PLANE: g.kind = G_PLANE; g.a = r.index; std::memcpy(g.mn, planes[r.index].point, 12); std::memcpy(g.mx, planes[r.index].normal, 12); g.skip = (uint32_t)gates.size() + 1; gates.push_back(g); break;
    case RT_SHAPE_SPHERE: g.kind = G_SPHERE; g.a = r.index; std::memcpy(g.mn, spheres[r.index].centre, 12);
        g.mx[0] = spheres[r.index].radius * spheres[r.index].radius; g.mx[1] = spheres[r.index].radius; g.mx[2] = 0.0f;   // mRadiusSqr{radius*radius}
        g.skip = (uint32_t)gates.size() + 1; gates.push_back(g); break;
    case RT_SHAPE_TRIANGLE: g.kind = G_TRI; g.a = add_triangle_slot(r.index, rank_base); g.skip = (uint32_t)gates.size() + 1; gates.push_back(g); break;
    case RT_SHAPE_MESH: emit_mesh(r.index, item, rank_base); break;
    case RT_SHAPE_MULTIMESH: {
        // A MultiMesh of ONE Mesh whose box is bit-identical to its own: the reference tests the same box twice (MultiMesh::hit,
        // then Mesh::hit), with the same answer — one Mesh node with that box says the same with one gate node fewer on every
        // walk (config 5: 2500 such shapes, 1.3 of the 101.5 gate tests per AO ray).
        if (multimeshes[r.index].num_meshes == 1 && !getenv("RT_KEEP_MM_GATE")) {
            const uint32_t mi = multimeshes[r.index].first_mesh;
            if (std::memcmp(mm_box_[r.index].mn, mesh_box_[mi].mn, 12) == 0 && std::memcmp(mm_box_[r.index].mx, mesh_box_[mi].mx, 12) == 0) { emit_mesh(mi, item, rank_base); break; }
        }
        size_t me = gates.size();
        std::memcpy(g.mn, mm_box_[r.index].mn, 12); std::memcpy(g.mx, mm_box_[r.index].mx, 12);
        g.kind = G_GATED;
        gates.push_back(g);
        uint32_t rank = rank_base;
        for (uint32_t k = 0; k < multimeshes[r.index].num_meshes; ++k) {
            uint32_t mi = multimeshes[r.index].first_mesh + k;
            emit_mesh(mi, item, rank);
            rank += meshes[mi].num_triangles;
        }
        gates[me].skip = (uint32_t)gates.size();
        break;
    }
    case RT_SHAPE_BVH: emit_bvh(r.index, item, rank_base); break;
    }
}

bool HostScene::subtree_opaque(const rt_shape_ref& r)
{
    auto opaque = [&](int32_t m) { return materials[m].kind != RT_MAT_REFRACTION; };
    switch (r.kind) {
    case RT_SHAPE_PLANE: return opaque(planes[r.index].material);
    case RT_SHAPE_SPHERE: return opaque(spheres[r.index].material);
    case RT_SHAPE_TRIANGLE: return opaque(triangles[r.index].material);
    case RT_SHAPE_MESH:
        for (uint32_t k = 0; k < meshes[r.index].num_triangles; ++k) if (!opaque(triangles[meshes[r.index].first_triangle + k].material)) return false;
        return true;
    case RT_SHAPE_MULTIMESH:
        for (uint32_t k = 0; k < multimeshes[r.index].num_meshes; ++k) if (!subtree_opaque({RT_SHAPE_MESH, multimeshes[r.index].first_mesh + k})) return false;
        return true;
    case RT_SHAPE_BVH:
        for (uint32_t k = 0; k < bvhs[r.index].num_leaves; ++k) if (!subtree_opaque(bvh_leaves[bvhs[r.index].first_leaf + k])) return false;
        return true;
    }
    return false;
}

void HostScene::fill_lead_planes(DScene& D) const
{
    D.num_lead_planes = 0;
    D.lead_axis = 0;
    for (size_t i = 0; i < gates.size() && i < RT_LEAD_PLANES; ++i) {
        const GateNode& g = gates[i];
        if ((g.kind & G_KIND_MASK) != G_PLANE) break;   // top-level nodes come first; a plane inside a BVHAccel follows a gated node
        float* pl = D.lead_planes[D.num_lead_planes++];
        std::memcpy(pl, g.mn, 12); std::memcpy(pl + 3, g.mx, 12);
        pl[6] = u2f(g.rank); pl[7] = u2f(g.a);
        // normal exactly +-(unit axis)?  (-0 components count as 0)
        uint32_t axis = 3;
        for (uint32_t k = 0; k < 3; ++k)
            if (std::fabs(g.mx[k]) == 1.0f && g.mx[(k + 1) % 3] == 0.0f && g.mx[(k + 2) % 3] == 0.0f) axis = k;
        D.lead_axis |= axis << (4u * (D.num_lead_planes - 1));
    }
}

int HostScene::build()
{
    error.clear();
    int rc = validate();
    if (rc) return rc;
    gates.clear(); planes4.clear(); spheres4.clear(); tris4.clear(); attr4.clear(); nodes4.clear(); fnodes4.clear();
    dmaterials.clear(); dlights.clear(); dtextures.clear(); texels.clear(); item_opaque.clear();
    bvh_max_depth = 0;
    append_jobs_.clear();
    hierarchy_seconds_ = 0.0;

    // Mesh / MultiMesh gate boxes (main.cpp:875-881, 757-765)
    mesh_box_.assign(meshes.size(), Aabb{});
    for (size_t i = 0; i < meshes.size(); ++i) {
        Aabb& b = mesh_box_[i];
        box_start(b, triangles[meshes[i].first_triangle].v0);
        for (uint32_t k = 0; k < meshes[i].num_triangles; ++k) {
            const rt_triangle& t = triangles[meshes[i].first_triangle + k];
            box_add(b, t.v0); box_add(b, t.v1); box_add(b, t.v2);
        }
    }
    mm_box_.assign(multimeshes.size(), Aabb{});
    for (size_t i = 0; i < multimeshes.size(); ++i) {
        mm_box_[i] = mesh_box_[multimeshes[i].first_mesh];
        for (uint32_t k = 0; k < multimeshes[i].num_meshes; ++k) box_merge(mm_box_[i], mesh_box_[multimeshes[i].first_mesh + k]);
    }
    mesh_root_.assign(meshes.size(), 0xFFFFFFFFu);
    const bool timing = getenv("RT_BUILD_TIMING") != nullptr;   // prints where rt_scene_commit's host time goes
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t_begin = now();
    occurrences.clear(); gpu_slots = gpu_nodes = 0;
    if (!gpu_bvh) prebuild_mesh_parts(); else { mesh_part_.clear(); }
    const auto t_parts = now();
    mesh_rank_.assign(meshes.size(), 0);

    for (auto& p : planes) {
        planes4.push_back({p.point[0], p.point[1], p.point[2], u2f((uint32_t)p.material)});
        planes4.push_back({p.normal[0], p.normal[1], p.normal[2], 0.0f});
    }
    for (auto& s : spheres) {
        spheres4.push_back({s.centre[0], s.centre[1], s.centre[2], s.radius});
        spheres4.push_back({s.radius * s.radius, u2f((uint32_t)s.material), 0.0f, 0.0f});   // mRadiusSqr{radius*radius}
    }
    extent = 1.0f;
    attr4.resize(triangles.size() * 2);
    {
        std::mutex mu;
        parallel_ranges(triangles.size(), 1 << 16, [&](size_t b, size_t e) {
            float ext = 1.0f;
            for (size_t i = b; i < e; ++i) {
                const rt_triangle& t = triangles[i];
                attr4[i * 2] = {t.uv0[0], t.uv0[1], t.uv1[0], t.uv1[1]};
                attr4[i * 2 + 1] = {t.uv2[0], t.uv2[1], 0.0f, 0.0f};
                for (int k = 0; k < 3; ++k) ext = std::max(ext, std::max(std::fabs(t.v0[k]), std::max(std::fabs(t.v1[k]), std::fabs(t.v2[k]))));
            }
            std::lock_guard<std::mutex> lk(mu);
            extent = std::max(extent, ext);
        });
    }
    {   // one growth step instead of many: every mesh is normally reached once
        size_t nn = 0, ns = 0;
        for (auto& part : mesh_part_) { nn += part.nodes.size(); ns += part.order.size(); }
        fnodes4.reserve(nn); tris4.reserve(3 * (ns + 64));
    }

    // gate nodes in scene order; ranks = the reference's visiting order
    uint32_t rank = 0;
    for (uint32_t i = 0; i < scene.size(); ++i) {
        if (scene[i].kind == RT_SHAPE_BVH) {
            const rt_bvh& b = bvhs[scene[i].index];
            if (b.num_leaves == 1) emit_shape(bvh_leaves[b.first_leaf], i, rank);   // root.leaf: shape->hit, no gate
            else { rc = emit_bvh(scene[i].index, i, rank); if (rc) return rc; }
        } else {
            emit_shape(scene[i], i, rank);
        }
        rank += prim_count(scene[i]);
        item_opaque.push_back(subtree_opaque(scene[i]) ? 1 : 0);
    }
    if (item_opaque.empty()) item_opaque.push_back(1);
    // Shadow rays may hand their meshes to the any-hit BVH kernel when no Mesh shares a world->scene entry with a
    // transparent material (Material::shadowed needs the entry's CLOSEST hit then, main.cpp:52-70).
    shadow_defer = false;
    {
        bool any_mesh = false, all_opaque = true;
        for (const GateNode& g : gates)
            if ((g.kind & G_KIND_MASK) == G_MESH) { any_mesh = true; all_opaque = all_opaque && item_opaque[g.item] != 0; }
        shadow_defer = any_mesh && all_opaque;
    }
    // A gate whose box is bit-identical to the enclosing gate's (a MultiMesh of one Mesh) repeats a test
    // that has just passed: flag it.  `extent` (the scale of the conservative paddings) covers gate boxes too.
    {
        std::vector<uint32_t> open_gates;   // enclosing gated nodes of the node being visited
        for (uint32_t i = 0; i < gates.size(); ++i) {
            while (!open_gates.empty() && gates[open_gates.back()].skip <= i) open_gates.pop_back();
            GateNode& g = gates[i];
            const uint32_t kind = g.kind & G_KIND_MASK;
            if (kind != G_GATED && kind != G_MESH) continue;
            for (int k = 0; k < 3; ++k)
                if (std::isfinite(g.mn[k]) && std::isfinite(g.mx[k])) extent = std::max(extent, std::max(std::fabs(g.mn[k]), std::fabs(g.mx[k])));
            if (!open_gates.empty()) {
                const GateNode& e = gates[open_gates.back()];
                if (std::memcmp(e.mn, g.mn, 12) == 0 && std::memcmp(e.mx, g.mx, 12) == 0) g.kind |= G_SAME_BOX;
            }
            if (kind == G_GATED) open_gates.push_back(i);
        }
    }
    if (bvh_max_depth > (uint32_t)kMaxDepth + 1) { error = "triangle BVH deeper than the traversal stack"; return RT_ERR_UNSUPPORTED; }
    const auto t_walk = now();
    run_append_jobs();
    const auto t_emit = now();
    if (gpu_bvh) {   // the node grid of quantize_nodes(), from the boxes the nodes will lie in
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (auto& o : occurrences) {
            if (o.node_off == 0xFFFFFFFFu) continue;
            for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], o.mn[a]); mx[a] = std::max(mx[a], o.mx[a]); }
        }
        for (int a = 0; a < 3; ++a) {
            if (!(mn[a] <= mx[a])) { mn[a] = 0.0f; mx[a] = 1.0f; }
            qbase[a] = mn[a];
            qscale[a] = std::max((mx[a] - mn[a]) / 65533.0f, 1e-30f);
        }
        if ((uint64_t)gpu_slots >= (1ull << 28) - 1u) { error = "too many triangle slots for a leaf reference"; return RT_ERR_UNSUPPORTED; }
    } else
    quantize_nodes();
    mesh_part_.clear(); mesh_part_.shrink_to_fit();
    build_qgates();
    if (timing) std::fprintf(stderr, "[rt_scene] reference hierarchies alone %.3f s\n", hierarchy_seconds_);
    if (timing) std::fprintf(stderr, "[rt_scene] triangle BVHs %.3f s, gate emission (reference hierarchies) %.3f s, node / triangle record copies %.3f s, node quantisation %.3f s\n",
                             secs(t_begin, t_parts), secs(t_parts, t_walk), secs(t_walk, t_emit), secs(t_emit, now()));

    has_bounce_materials = false;
    for (auto& m : materials) {
        DMaterial d{};
        d.kind = m.kind; std::memcpy(d.colour, m.colour, 12);
        d.kd = m.kd; d.ka = m.ka; d.ks = m.ks; d.spec_exp = m.spec_exp; d.ior = m.ior; d.texture = m.texture;
        dmaterials.push_back(d);
        if (m.kind == RT_MAT_MIRROR || m.kind == RT_MAT_REFRACTION) has_bounce_materials = true;
    }
    for (auto& l : lights) {
        DLight d{};
        d.kind = l.kind; std::memcpy(d.colour, l.colour, 12); d.radiance = l.radiance;
        if (l.kind == RT_LIGHT_DIRECTIONAL) {   // Directional::setDirection normalises (main.cpp:1315-1318)
            f3 n = normalize(F3(l.v[0], l.v[1], l.v[2]));
            d.v[0] = n.x; d.v[1] = n.y; d.v[2] = n.z;
        } else std::memcpy(d.v, l.v, 12);
        dlights.push_back(d);
    }
    for (auto& t : textures) {
        DTexture d{};
        d.offset = (uint32_t)texels.size(); d.width = t.width; d.height = t.height;
        d.set = (t.rgba && t.width > 0 && t.height > 0) ? 1 : 0;
        if (d.set) texels.insert(texels.end(), t.rgba, t.rgba + (size_t)t.width * t.height * 4);
        dtextures.push_back(d);
    }
    return RT_OK;
}

} // namespace rt
