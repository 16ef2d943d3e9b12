// rt_scene.hpp — host-side scene store and flattener (no HIP here: pure C++17).
//
// Holds the copies made by rt_scene_set_*, validates them, rebuilds the reference's
// shape-level structure (BoundingVolumeBox / Mesh / MultiMesh gates and BVHAccel's bottom-up
// nearest-centroid hierarchy, reference main.cpp:191-243, 323-423, 742-766, 851-882) and
// builds one binned-SAH triangle BVH per Mesh in the 64-byte node layout the kernels fetch.
#pragma once
#include <string>
#include <vector>

#include "rt_core.hpp"

namespace rt {

struct Aabb { float mn[3], mx[3]; };

struct HostScene {
    // ---- description (copied from the caller) ----
    std::vector<rt_plane> planes;
    std::vector<rt_sphere> spheres;
    std::vector<rt_triangle> triangles;
    std::vector<rt_mesh> meshes;
    std::vector<rt_multimesh> multimeshes;
    std::vector<rt_bvh> bvhs;
    std::vector<rt_shape_ref> bvh_leaves;
    std::vector<rt_shape_ref> scene;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;              // rgba pointers re-targeted into texel_store
    std::vector<std::vector<uint8_t>> texel_store;
    std::vector<rt_light> lights;
    rt_ambient ambient{};

    // ---- flattened form (what gets uploaded) ----
    std::vector<GateNode> gates;
    std::vector<QGate> qgates;          // compact copy for the LDS-resident gate walk (empty: more than RT_QGATE_MAX nodes)
    float gq_base[3] = {0, 0, 0}, gq_scale[3] = {1, 1, 1};
    void build_qgates();
    std::vector<q4> planes4, spheres4, tris4, attr4, nodes4;
    std::vector<q4> fnodes4;            // build-time float nodes (4 q4 each, two-wide); nodes4 holds the quantised four-wide nodes (4 q4 each)
    float qbase[3] = {0, 0, 0}, qscale[3] = {1, 1, 1};   // node grid: coordinate = qbase + q * qscale
    std::vector<DMaterial> dmaterials;
    std::vector<DLight> dlights;
    std::vector<DTexture> dtextures;
    std::vector<uint8_t> texels;
    std::vector<uint8_t> item_opaque;
    bool shadow_defer = false;   // see DScene::shadow_defer
    float extent = 0.0f;
    bool has_bounce_materials = false;
    uint32_t bvh_max_depth = 0;
    void fill_lead_planes(DScene& D) const;   // the leading top-level Plane nodes of `gates`, for DScene::lead_planes

    std::string error;

    int build();   // rt_status

    // with gpu_bvh set, build() does everything but the triangle BVHs: it lays out the slot and
    // node ranges of every Mesh occurrence (and of the free triangles) and leaves tris4 / nodes4 empty; the LBVH
    // builder on the device (rt_lbvh.hip) fills them.  Node 0 of an occurrence's range is its root.
    struct Occurrence { uint32_t first_triangle, count, rank_base, slot_off, node_off; float mn[3], mx[3]; };
    bool gpu_bvh = false;
    std::vector<Occurrence> occurrences;   // gpu_bvh only
    uint32_t gpu_slots = 0, gpu_nodes = 0; // totals of the layout (gpu_bvh only)
    // One (sub)tree built into its own buffers, with node indices and triangle slots local to them, so that
    // independent subtrees and meshes can be built on different threads and appended afterwards.
    struct BvhPart {
        std::vector<q4> nodes;          // 4 per node: child boxes + {ref0, ref1}
        std::vector<uint32_t> order;    // global triangle ids in slot order
        uint32_t max_depth = 0;
        uint32_t root = REF_NONE;       // reference of the part's root (local)
    };

    // reference hierarchy of one BVHAccel, exposed for tests (node = {leaf,left,right,box})
    struct RefNode { bool leaf; rt_shape_ref shape; Aabb box; uint32_t left, right; bool accessed; };
    int reference_hierarchy(const rt_bvh& b, std::vector<RefNode>& out);
    bool force_reference_scan = false;   // tests: use the reference's own O(n^3) pairing scan for every layer

private:
    std::vector<Aabb> mesh_box_, mm_box_;
    std::vector<uint32_t> mesh_root_, mesh_rank_; // triangle-BVH root per mesh (0xFFFFFFFF = not built yet) and the rank base it was built for
    int validate();
    int shape_box(const rt_shape_ref& r, Aabb& out);
    uint32_t prim_count(const rt_shape_ref& r);
    uint32_t mesh_bvh(uint32_t mesh, uint32_t rank_base);
    void build_mesh_part(uint32_t mesh, int par, BvhPart& part) const;
    void prebuild_mesh_parts();
    std::vector<BvhPart> mesh_part_;
    struct AppendJob { uint32_t mesh, rank_base, node_off, slot_off; };
    std::vector<AppendJob> append_jobs_;
    void run_append_jobs();
    double hierarchy_seconds_ = 0.0;
    void write_triangle_slot(uint32_t slot, uint32_t gid, uint32_t rank);
    void quantize_nodes();
    void emit_shape(const rt_shape_ref& r, uint32_t item, uint32_t rank_base);
    void emit_mesh(uint32_t mesh, uint32_t item, uint32_t rank_base);
    int emit_bvh(uint32_t bvh, uint32_t item, uint32_t rank_base);
    uint32_t add_triangle_slot(uint32_t gid, uint32_t rank);
    bool subtree_opaque(const rt_shape_ref& r);
};

} // namespace rt
