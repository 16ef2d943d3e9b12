// tests/emul/emul.cpp — TEST-ONLY host emulation of the device pipeline.
//
// Compiles the very same per-item kernel bodies (raytracer_amd/csrc/rt_core.hpp,
// rt_pipeline.hpp) and the host flattener (rt_scene.cpp) with g++ and runs them sequentially,
// so the CPU test-suite (-m "not gpu") can check the device arithmetic, the flattened scene and
// the triangle BVH against the oracle without a GPU.  It is not part of librt_amd.so and the
// product never loads it.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../raytracer_amd/csrc/rt_pipeline.hpp"
#include "../../raytracer_amd/csrc/rt_scene.hpp"

using namespace rt;

namespace {
struct LoopBackend {
    DScene S; Batch B; bool stats;
    std::vector<uint32_t> stack = std::vector<uint32_t>(RT_STACK_DEPTH);
    Counters cnt[3] = {};
    void begin_batch(const RenderArgs&) { std::memset(B.counters, 0, sizeof(uint32_t) * C_TOTAL); }
    void raygen(const RenderArgs& A, uint32_t n) { for (uint32_t i = 0; i < n; ++i) raygen_item(S, A, B, i); }
    uint32_t count(int d, uint32_t fixed) { return d == 0 ? fixed : B.counters[C_QUEUE + d]; }
    void trace(const RenderArgs& A, int d, uint32_t fixed) {
        Stack st{stack.data(), 1};
        QCursor qc{0, 0};
        for (uint32_t i = 0, n = count(d, fixed); i < n; ++i) {
            bool overflow = false;
            trace_item<true>(S, A, B, d, i, true, st, cnt[0], qc, overflow);
            if (overflow) B.ao_ovf[B.counters[C_TR_OVF + d]++] = i;   // a "wave" of one lane
        }
    }
    void trace_overflow(const RenderArgs& A, int d, uint32_t fixed) {
        Stack st{stack.data(), 1};
        for (uint32_t e = 0; e < B.counters[C_TR_OVF + d]; ++e) trace_overflow_item<true>(S, A, B, d, B.ao_ovf[e], 0, count(d, fixed), st, cnt[0]);
    }
    void bvh_trace(const RenderArgs& A, int d) {
        Stack st{stack.data(), 1};
        for (uint32_t i = 0, n = std::min(B.counters[C_BQ + d], B.cq_cap); i < n; ++i) bvh_item<BVH_CLOSEST, true>(S, B, i, st, cnt[0]);
    }
    void bvh_ao(const RenderArgs& A, int d) {
        Stack st{stack.data(), 1};
        for (uint32_t i = 0, n = std::min(B.counters[C_BQA + d], B.bq_cap); i < n; ++i) bvh_item<BVH_AO, true>(S, B, i, st, cnt[2]);
    }
    void prep(const RenderArgs& A, int d, uint32_t fixed) {
        QCursor q{0, 0};
        uint32_t skipped = 0;
        for (uint32_t i = 0, n = count(d, fixed); i < n; ++i) prep_item(S, A, B, d, i, true, q, skipped);
        B.stats[ST_SKIPPED] += skipped;
    }
    void shadow(const RenderArgs& A, int d) {
        Stack st{stack.data(), 1};
        const uint32_t n = B.counters[C_SH + d];
        if (!S.shadow_defer) {
            for (uint32_t i = 0; i < n; ++i) shadow_item<true>(S, A, B, d, i, st, cnt[1]);
            return;
        }
        QCursor qc{0, 0};
        for (uint32_t i = 0; i < n; ++i) {
            bool overflow = false;
            shadow_defer_item<true>(S, A, B, d, i, true, st, cnt[1], qc, overflow);
            if (overflow) B.ao_ovf[B.counters[C_SH_OVF + d]++] = i;   // a "wave" of one lane
        }
        for (uint32_t i = 0, m = std::min(B.counters[C_BQS + d], B.bq_cap); i < m; ++i) bvh_item<BVH_SHADOW, true>(S, B, i, st, cnt[1]);
        const uint32_t rays = cnt[1].rays;
        for (uint32_t e = 0; e < B.counters[C_SH_OVF + d]; ++e) shadow_overflow_item<true>(S, A, B, d, B.ao_ovf[e], 0, n, st, cnt[1]);
        cnt[1].rays = rays;   // counted by shadow_defer_item
    }
    void ao(const RenderArgs& A, int d, uint32_t fixed) {
        Stack st{stack.data(), 1};
        QCursor qc{0, 0};
        for (uint32_t slot = 0, nslots = count(d, fixed); slot < nslots; ++slot) {
            bool overflow = false;
            ao_item<true>(S, A, B, d, slot, true, st, cnt[2], qc, overflow);
            if (overflow) B.ao_ovf[B.counters[C_AO + d]++] = slot;   // a "wave" of one lane
        }
    }
    void ao_overflow(const RenderArgs& A, int d, uint32_t fixed) {
        Stack st{stack.data(), 1};
        const uint32_t nslots = count(d, fixed);
        for (uint32_t e = 0; e < B.counters[C_AO + d]; ++e) ao_overflow_item<true>(S, A, B, B.ao_ovf[e], 0, nslots, st, cnt[2]);
    }
    void shade(const RenderArgs& A, int d, uint32_t fixed) { for (uint32_t i = 0, n = count(d, fixed); i < n; ++i) shade_item(S, A, B, d, i); }
    void resolve(const RenderArgs& A, uint32_t n) {
        for (uint32_t i = 0; i < n; ++i) resolve_item(S, A, B, i);
        for (int d = 1; d < A.levels; ++d) B.stats[ST_BOUNCE] += B.counters[C_QUEUE + d];
        B.stats[ST_SHADOW] += cnt[1].rays; B.stats[ST_AO] += cnt[2].rays;
        cnt[1].rays = cnt[2].rays = 0;
    }
};
} // namespace

static void fill_scene(const HostScene& H, DScene& S)
{
    S = DScene{};
    S.gates = H.gates.data(); S.num_gates = (uint32_t)H.gates.size(); S.gates_fast = nullptr; S.num_gates_fast = 0;
    S.qgates = H.qgates.data(); S.num_qgates = (uint32_t)H.qgates.size();
    S.gq_base = F3(H.gq_base[0], H.gq_base[1], H.gq_base[2]); S.gq_scale = F3(H.gq_scale[0], H.gq_scale[1], H.gq_scale[2]);
    S.planes = H.planes4.data(); S.spheres = H.spheres4.data(); S.tris = H.tris4.data();
    S.tri_attr = H.attr4.data(); S.nodes = H.nodes4.data();
    S.materials = H.dmaterials.data(); S.lights = H.dlights.data(); S.num_lights = (uint32_t)H.dlights.size();
    S.textures = H.dtextures.data(); S.texels = H.texels.data(); S.num_textures = (uint32_t)H.dtextures.size();
    S.item_opaque = H.item_opaque.data(); S.num_items = (uint32_t)H.item_opaque.size(); S.shadow_defer = H.shadow_defer ? 1u : 0u;
    S.ambient = H.ambient;
    S.prim_base_sphere = (uint32_t)H.planes.size(); S.prim_base_tri = (uint32_t)(H.planes.size() + H.spheres.size());
    S.cull_pad = 1.0f / 262144.0f; S.extent = H.extent; S.debug = 0;
    H.fill_lead_planes(S);
    S.qbase = F3(H.qbase[0], H.qbase[1], H.qbase[2]); S.qscale = F3(H.qscale[0], H.qscale[1], H.qscale[2]);
}

extern "C" {

static std::string g_err;
const char* emul_last_error() { return g_err.c_str(); }

// out_counts: {primary, shadow, ao, bounce, nodes[3], tris[3], gates[3], skipped}
int emul_render(const rt_scene_desc* d, const rt_camera* cam, const rt_params* p,
                float* rgb, uint32_t* primid, float* t, unsigned long long* out_counts)
{
    HostScene H;
    H.planes.assign(d->planes, d->planes + d->num_planes);
    H.spheres.assign(d->spheres, d->spheres + d->num_spheres);
    H.triangles.assign(d->triangles, d->triangles + d->num_triangles);
    H.meshes.assign(d->meshes, d->meshes + d->num_meshes);
    H.multimeshes.assign(d->multimeshes, d->multimeshes + d->num_multimeshes);
    H.bvhs.assign(d->bvhs, d->bvhs + d->num_bvhs);
    H.bvh_leaves.assign(d->bvh_leaves, d->bvh_leaves + d->num_bvh_leaves);
    H.scene.assign(d->scene, d->scene + d->num_scene);
    H.materials.assign(d->materials, d->materials + d->num_materials);
    H.textures.assign(d->textures, d->textures + d->num_textures);
    H.lights.assign(d->lights, d->lights + d->num_lights);
    H.ambient = d->ambient;
    int rc = H.build();
    if (rc) { g_err = H.error; return rc; }

    LoopBackend be;
    DScene& S = be.S;
    fill_scene(H, S);

    RenderArgs A{};
    A.cam = make_camera(*cam);
    A.width = p->width; A.height = p->height; A.x0 = p->x0; A.y0 = p->y0; A.x1 = p->x1; A.y1 = p->y1;
    A.spp = p->num_samples; A.sampler = p->sampler; A.max_depth = p->max_depth; A.levels = p->max_depth + 2;
    A.light_atten = p->light_attenuation; A.seed = p->seed;
    A.tiles_x = (uint32_t)((p->x1 - p->x0 + RT_TILE - 1) / RT_TILE); A.tiles_y = (uint32_t)((p->y1 - p->y0 + RT_TILE - 1) / RT_TILE);
    A.tile_rank = p->tile_rank; A.tile_world = p->tile_world > 0 ? p->tile_world : 1;
    A.out_rgb = rgb; A.out_prim = primid; A.out_t = t; A.slab = 0;
    uint32_t total = A.tiles_x * A.tiles_y;
    uint32_t owned = total > (uint32_t)A.tile_rank ? (total - A.tile_rank + A.tile_world - 1) / A.tile_world : 0;
    uint32_t tiles_per_batch = 4;
    uint32_t P = tiles_per_batch * RT_TILE * RT_TILE * (uint32_t)A.spp;
    uint32_t nl = S.num_lights ? S.num_lights : 1;
    uint32_t n2 = (uint32_t)(S.ambient.occlusion_samples * S.ambient.occlusion_samples);
    Batch& B = be.B;
    B.P = P;
    std::vector<q4> ro0(P), ro1(P), rd0(P), rd1(P), hit(P), nrm(P), lvl((size_t)P * A.levels);
    std::vector<uint32_t> pstate(P), q0(P), q1(P), aos(P), shq((size_t)P * nl), counters(C_TOTAL);
    std::vector<uint32_t> aot((size_t)P * (n2 ? n2 : 1)), aovf((size_t)P * (n2 ? n2 : 1));
    std::vector<float> shr((size_t)P * nl), lph((size_t)P * A.levels * (nl > 1 ? nl - 1 : 1));
    std::vector<uint8_t> kind((size_t)P * A.levels);
    std::vector<unsigned long long> stats(ST_TOTAL, 0);
    // a deliberately small queue so the inline fallback of defer_bvh is exercised too
    std::vector<q4> aoo(P), aou(P), aow(P);
    B.bq_cap = P * (n2 ? n2 : 1) / 3 + 1;
    B.cq_cap = B.bq_cap;
    B.sh_cap = (uint32_t)shq.size();
    std::vector<q4> bqa(B.bq_cap), bqb(B.bq_cap), bqc(B.bq_cap); std::vector<uint32_t> bqr((size_t)B.bq_cap * RT_DEFER_ROOTS);
    B.bq_a = bqa.data(); B.bq_b = bqb.data(); B.bq_c = bqc.data(); B.bq_roots = bqr.data();
    B.ray_o[0] = ro0.data(); B.ray_o[1] = ro1.data(); B.ray_d[0] = rd0.data(); B.ray_d[1] = rd1.data();
    B.hit = hit.data(); B.nrm = nrm.data(); B.pstate = pstate.data(); B.queue[0] = q0.data(); B.queue[1] = q1.data();
    B.ao_o = aoo.data(); B.ao_u = aou.data(); B.ao_w = aow.data(); B.ao_slot = aos.data(); B.sh_queue = shq.data(); B.ao_key = aot.data(); B.ao_ovf = aovf.data(); B.ovf_cap = (uint32_t)aovf.size(); B.sh_res = shr.data();
    B.lvl = lvl.data(); B.lvl_ph = lph.data(); B.lvl_kind = kind.data(); B.counters = counters.data(); B.stats = stats.data();
    render_frame(be, A, owned, tiles_per_batch, H.has_bounce_materials);
    if (out_counts) {
        out_counts[0] = (unsigned long long)(p->x1 - p->x0) * (p->y1 - p->y0) * A.spp;
        out_counts[1] = stats[ST_SHADOW]; out_counts[2] = stats[ST_AO]; out_counts[3] = stats[ST_BOUNCE];
        for (int k = 0; k < 3; ++k) { out_counts[4 + k] = be.cnt[k].nodes; out_counts[7 + k] = be.cnt[k].tris; out_counts[10 + k] = be.cnt[k].gates; }
        out_counts[13] = stats[ST_SKIPPED];
    }
    return 0;
}

// reference hierarchy of bvh #which as rebuilt by the product's flattener: {leaf,left,right,min,max} per node
int emul_bvh_dump(const rt_scene_desc* d, uint32_t which, float* out, int cap_nodes)
{
    HostScene H;
    H.spheres.assign(d->spheres, d->spheres + d->num_spheres);
    H.planes.assign(d->planes, d->planes + d->num_planes);
    H.triangles.assign(d->triangles, d->triangles + d->num_triangles);
    H.meshes.assign(d->meshes, d->meshes + d->num_meshes);
    H.multimeshes.assign(d->multimeshes, d->multimeshes + d->num_multimeshes);
    H.bvhs.assign(d->bvhs, d->bvhs + d->num_bvhs);
    H.bvh_leaves.assign(d->bvh_leaves, d->bvh_leaves + d->num_bvh_leaves);
    H.scene.assign(d->scene, d->scene + d->num_scene);
    H.materials.assign(d->materials, d->materials + d->num_materials);
    H.textures.assign(d->textures, d->textures + d->num_textures);
    H.lights.assign(d->lights, d->lights + d->num_lights);
    H.ambient = d->ambient;
    H.force_reference_scan = getenv("EMUL_REFERENCE_SCAN") != nullptr;
    if (H.build() != 0 || which >= H.bvhs.size()) { g_err = H.error; return -1; }
    std::vector<HostScene::RefNode> N;
    if (H.reference_hierarchy(H.bvhs[which], N) != 0) return -1;
    int n = (int)N.size();
    for (int i = 0; i < n && i < cap_nodes; ++i) {
        float* o = out + (size_t)i * 9;
        o[0] = N[i].leaf ? 1.0f : 0.0f;
        o[1] = N[i].leaf ? -1.0f : (float)N[i].left;
        o[2] = N[i].leaf ? -1.0f : (float)N[i].right;
        for (int k = 0; k < 3; ++k) { o[3 + k] = N[i].box.mn[k]; o[6 + k] = N[i].box.mx[k]; }
    }
    return n;
}

// Soundness of the conservative gate classification (rt_core.hpp gate_maybe / gate_surely) against the
// exact reference test, on random and adversarial (edge / corner grazing, tiny direction components,
// origins on box planes) rays.  out = {tested, maybe_false, surely_true, ambiguous, violations, accepted by gate_class alone}
int emul_gate_check(uint32_t n, uint32_t seed, double* out)
{
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)((s >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 53); };
    double cnt[6] = {0, 0, 0, 0, 0, 0};
    const float cull_pad = 1.0f / 262144.0f;
    for (uint32_t it = 0; it < n; ++it) {
        float mn[3], mx[3];
        float ext = 1.0f;
        for (int a = 0; a < 3; ++a) {
            float c = (float)((rnd() - 0.5) * 2000.0), h = (float)(rnd() * rnd() * 600.0);
            if ((it & 63) == 7 && a == (int)(it >> 6) % 3) h = 0.0f;   // flat boxes
            mn[a] = c - h; mx[a] = c + h;
            ext = fmaxf(ext, fmaxf(fabsf(mn[a]), fabsf(mx[a])));
        }
        f3 o, d;
        const int mode = it % 8;
        o = F3((float)((rnd() - 0.5) * 3000.0), (float)((rnd() - 0.5) * 3000.0), (float)((rnd() - 0.5) * 3000.0));
        if (mode == 1) { o.x = mn[0] + (float)rnd() * (mx[0] - mn[0]); o.y = mn[1] + (float)rnd() * (mx[1] - mn[1]); o.z = mn[2] + (float)rnd() * (mx[2] - mn[2]); }   // inside
        if (mode == 2) o.y = (it & 8) ? mn[1] : mx[1];   // on a plane
        // aim at a point of the box: interior, face, edge or corner
        f3 tgt;
        float* tp = &tgt.x;
        for (int a = 0; a < 3; ++a) {
            double u = rnd();
            if (mode >= 3 && mode <= 5 && a != (int)(it >> 3) % 3) u = (rnd() < 0.5) ? 0.0 : 1.0;   // edges
            if (mode == 6) u = (rnd() < 0.5) ? 0.0 : 1.0;                                          // corners
            tp[a] = mn[a] + (float)u * (mx[a] - mn[a]);
            if (mode == 5) tp[a] += (float)((rnd() - 0.5) * 1e-3);
        }
        d = tgt - o;
        if (mode == 7) { d = F3((float)(rnd() - 0.5), (float)(rnd() - 0.5), (float)(rnd() - 0.5)); float* dp = &d.x; dp[it % 3] = (float)((rnd() - 0.5) * 4e-5); }
        if (dot(d, d) == 0.0f) continue;
        d = normalize(d);
        if ((it & 3) == 3) d = d * -1.0f;
        const RayGate rg = make_gate(cull_pad, ext, o, d);
        const bool exact = box_intersects(mn, mx, o, d);
        const bool maybe = gate_maybe(mn, mx, rg);
        const int cls = gate_class(mn, mx, rg);   // the fused form: 0 <=> !maybe, 1 => the exact test passes
        const bool surely = maybe && (cls == 1 || gate_surely(mn, mx, o, d, rg));
        cnt[0]++;
        if ((cls == 0) != !maybe || (cls == 1 && !exact) || (bool)RT_GATE_ENTER(mn, mx, o, d, rg) != exact) cnt[4]++;
        if (cls == 1) cnt[5]++;
        if (!maybe) { cnt[1]++; if (exact) cnt[4]++; }
        else if (surely) { cnt[2]++; if (!exact) cnt[4]++; }
        else cnt[3]++;
    }
    for (int k = 0; k < 6; ++k) out[k] = cnt[k];
    return 0;
}

// chunk size a launch takes for its queue appends (rt_pipeline.hpp)
uint32_t emul_queue_chunk(unsigned long long appends, uint32_t waves, uint32_t largest) { return queue_chunk(appends, waves, largest); }

// The gate walk from the compact LDS copy of the gate nodes (gate_step_q: k_ao_walk_lds) against the walk over the full
// nodes (gate_step), ray by ray: same visiting sequence, same closest primitive, same mesh roots in the same order.
// Rays: random origins inside / around the scene, random directions, plus rays aimed at box corners, edges and faces of
// the gate boxes (where a conservative classification is most likely to go wrong) and rays starting on box planes.
// out = {rays, gate steps, steps the 16-byte copy decided without the exact box, mismatches, nodes in the copy}
int emul_qgate_compare(const rt_scene_desc* d, uint32_t nrays, uint32_t seed, unsigned long long* out)
{
    HostScene H;
    H.planes.assign(d->planes, d->planes + d->num_planes);
    H.spheres.assign(d->spheres, d->spheres + d->num_spheres);
    H.triangles.assign(d->triangles, d->triangles + d->num_triangles);
    H.meshes.assign(d->meshes, d->meshes + d->num_meshes);
    H.multimeshes.assign(d->multimeshes, d->multimeshes + d->num_multimeshes);
    H.bvhs.assign(d->bvhs, d->bvhs + d->num_bvhs);
    H.bvh_leaves.assign(d->bvh_leaves, d->bvh_leaves + d->num_bvh_leaves);
    H.scene.assign(d->scene, d->scene + d->num_scene);
    H.materials.assign(d->materials, d->materials + d->num_materials);
    H.textures.assign(d->textures, d->textures + d->num_textures);
    H.lights.assign(d->lights, d->lights + d->num_lights);
    H.ambient = d->ambient;
    int rc = H.build();
    if (rc) { g_err = H.error; return rc; }
    DScene S;
    fill_scene(H, S);
    out[0] = out[1] = out[2] = out[3] = 0; out[4] = S.num_qgates;
    if (!S.num_qgates) return 0;
    std::vector<uint32_t> boxed;
    for (uint32_t i = 0; i < S.num_gates; ++i) { const uint32_t k = S.gates[i].kind & G_KIND_MASK; if (k == G_GATED || k == G_MESH) boxed.push_back(i); }
    uint64_t rng = 0x9E3779B97F4A7C15ull ^ ((uint64_t)seed << 17);
    auto u01 = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 40) * (1.0 / 16777216.0)); };
    const float E = S.extent > 0 ? S.extent : 1.0f;
    std::vector<uint32_t> stack_a(RT_DEFER_ROOTS + 1), stack_b(RT_DEFER_ROOTS + 1), roots_a, roots_b;
    for (uint32_t r = 0; r < nrays; ++r) {
        f3 o = F3((u01() * 2 - 1) * E * 1.1f, (u01() * 2 - 1) * E * 1.1f, (u01() * 2 - 1) * E * 1.1f);
        f3 t = F3((u01() * 2 - 1) * E, (u01() * 2 - 1) * E, (u01() * 2 - 1) * E);
        const uint32_t mode = r % 4u;
        if (mode != 0 && !boxed.empty()) {   // aim at a feature of a gate box
            const GateNode& g = S.gates[boxed[(size_t)(u01() * boxed.size()) % boxed.size()]];
            auto pick = [&](int a) { const float w = u01(); return w < 0.4f ? g.mn[a] : (w < 0.8f ? g.mx[a] : g.mn[a] + (g.mx[a] - g.mn[a]) * u01()); };
            t = F3(pick(0), pick(1), pick(2));
            if (mode == 3) { o = F3(pick(0), pick(1), pick(2)); t = F3((u01() * 2 - 1) * E, (u01() * 2 - 1) * E, (u01() * 2 - 1) * E); }   // origin on the box
        }
        f3 dir = t - o;
        if (!(dot(dir, dir) > 0.0f)) continue;
        dir = normalize(dir);
        if (r % 16u == 5u) dir.y = 0.0f;               // axis-parallel components
        if (r % 16u == 9u) { dir.x = 0.0f; dir.z = 1e-6f; }
        const RayGate rg = make_gate(S.cull_pad, S.extent, o, dir);
        const RayCull4 gc = make_cull4_from(make_cull_grid(S.cull_pad, S.extent, S.gq_base, S.gq_scale, o, dir));
        Hit ha, hb;
        ha.t = hb.t = 1000.0f; ha.prim = hb.prim = H_MISS; ha.u = ha.v = hb.u = hb.v = 0; ha.rank = hb.rank = 0;
        Counters ca{0, 0, 0, 0}, cb{0, 0, 0, 0};
        uint32_t ia = lead_planes_hit(S, o, dir, rg, ha), ib = lead_planes_hit(S, o, dir, rg, hb), spa = 0, spb = 0;
        roots_a.clear(); roots_b.clear();
        bool bad = false;
        while (ia < S.num_gates || ib < S.num_gates) {
            if (ia != ib) { bad = true; break; }
            const uint32_t k = (S.qgates[ib].w >> 24) & 7u;
            const uint32_t i0 = ib;
            ia = gate_step<true>(S, o, dir, rg, ia, ha, Stack{stack_a.data(), 1}, spa, ca);
            ib = gate_step_q<true>(S, S.qgates, o, dir, rg, gc, ib, hb, Stack{stack_b.data(), 1}, spb, cb);
            out[1]++;
            if (k == G_GATED || k == G_MESH) {   // decided from the 16 bytes alone: flagged as same box, or the decoded box was missed
                const QGate q = S.qgates[i0];
                const bool from_q = (q.w & (1u << 27)) != 0u || !gate_maybe4(gc, q.x, q.y, q.z);
                if (from_q) out[2]++;
            }
            for (uint32_t k2 = 0; k2 < spa; ++k2) roots_a.push_back(stack_a[k2]);
            for (uint32_t k2 = 0; k2 < spb; ++k2) roots_b.push_back(stack_b[k2]);
            spa = spb = 0;
        }
        if (!bad) bad = ia != ib || roots_a != roots_b || f2u(ha.t) != f2u(hb.t) || ha.prim != hb.prim || ha.rank != hb.rank || ca.gates != cb.gates;
        out[0]++;
        if (bad) out[3]++;
    }
    return 0;
}

} // extern "C"
