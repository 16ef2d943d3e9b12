// oracle/ref_harness.cpp — TEST INFRASTRUCTURE, never shipped, never on the product path.
//
// Builds the *real* reference (andrewyung/Raytracer) into oracle/_ref/libref_oracle.so by
// #include-ing its translation unit where it lies under /root/reference (nothing is copied).
// The reference's main() is compiled as an unused static function, so its one unresolved
// external (atlas::utils::loadObjMesh, whose library is not in the reference repo) is never
// referenced: meshes are handed to the reference's own MultiMesh/Mesh constructors as
// atlas::utils::ObjMesh values built from an rt_scene_desc.
//
// What is ours in this file: the scene construction from rt_scene_desc, the per-pixel loop
// (the reference's own loop mis-indexes non-square images, SURVEY Q14, and its shared Sampler
// is racy, Q18), the rand() hook and the probe entry points.  Everything that computes —
// Camera::calculateRay, Shape::hit, Material::shade and all they call — is the reference's code.
//
// Two textual hooks are applied to the reference's function bodies (not to its header):
//   rand            -> harness_rand      (libc rand | counter hash of include/rt_rng.h | constant)
//   maxBounceDepth  -> (g_max_depth)     (a run-time int instead of the compile-time 3)
#include <condition_variable>
#include <functional>
#include <optional>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <atomic>
#include <string>
#include <vector>
#include <memory>
#include <chrono>

#include "Raytracer.hpp" // the reference's header (found through -I/root/reference/Raytracer)

#include "../include/rt_amd.h"
#include "../include/rt_rng.h"

// ---- hooks ------------------------------------------------------------------------------
static int g_max_depth = 3;
static int g_rng_mode = 0; // 0 libc rand, 1 counter hash, 2 constant RAND_MAX/2
static uint32_t g_seed = 0;
static thread_local uint32_t tl_pixel = 0, tl_sample = 0, tl_counter = 0;
static thread_local uint64_t tl_rays = 0;

static int harness_rand()
{
    if (g_rng_mode == 1) return (int)rt_rand31(g_seed, tl_pixel, tl_sample, tl_counter++);
    if (g_rng_mode == 2) return RAND_MAX / 2;
    return std::rand();
}

#define rand harness_rand
#define maxBounceDepth (g_max_depth)
#define main static __attribute__((unused)) reference_main_unused
#include "main.cpp" // the reference's translation unit (through -I/root/reference/Raytracer)
#undef main
#undef rand
#undef maxBounceDepth

// ---- helpers ----------------------------------------------------------------------------
namespace {

struct FlatAmbient : Light // world->ambient with the base-class Light::L (= radiance*colour): "AO off"
{
    atlas::math::Vector getDirection(ShadeRec const&) override { return atlas::math::Vector{0.0f}; }
    atlas::math::Vector getSourcePoint(ShadeRec const&) override { return atlas::math::Vector{0.0f}; }
};

struct RayCounter : Shape // appended to world->scene: one call per closest-hit query over the scene
{
    bool hit(atlas::math::Ray<atlas::math::Vector> const&, ShadeRec&) const override { ++tl_rays; return false; }
    bool intersectRay(atlas::math::Ray<atlas::math::Vector> const&, float&) const override { return false; }
};

inline Point P3(const float* v) { return Point{v[0], v[1], v[2]}; }
inline Vector2 P2(const float* v) { return Vector2{v[0], v[1]}; }

bool write_tga(const std::string& path, const rt_texture& t)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    unsigned char hdr[18] = {0};
    hdr[2] = 2; // uncompressed true colour
    hdr[12] = (unsigned char)(t.width & 255); hdr[13] = (unsigned char)(t.width >> 8);
    hdr[14] = (unsigned char)(t.height & 255); hdr[15] = (unsigned char)(t.height >> 8);
    hdr[16] = 32; hdr[17] = 0x28; // 8 alpha bits, top-left origin
    std::fwrite(hdr, 1, 18, f);
    std::vector<unsigned char> row((size_t)t.width * 4);
    for (int y = 0; y < t.height; ++y) {
        for (int x = 0; x < t.width; ++x) {
            const uint8_t* s = t.rgba + ((size_t)y * t.width + x) * 4;
            row[x * 4 + 0] = s[2]; row[x * 4 + 1] = s[1]; row[x * 4 + 2] = s[0]; row[x * 4 + 3] = s[3];
        }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
    return true;
}

struct Built
{
    std::shared_ptr<World> world;
    std::vector<std::shared_ptr<Material>> materials;
    std::vector<std::shared_ptr<Plane>> planes;
    std::vector<std::shared_ptr<Sphere>> spheres;
    std::vector<std::shared_ptr<Triangle>> tris;       // only free Triangle shapes (else null)
    std::vector<std::shared_ptr<Mesh>> meshes;          // only stand-alone Mesh shapes
    std::vector<std::shared_ptr<MultiMesh>> multimeshes;
    std::vector<std::shared_ptr<BVHAccel>> bvhs;
    std::string error;
};

void apply_textured_params(Textured* t, const rt_material& m)
{
    t->setAmbientReflection(m.ka);
    t->setSpecularReflection(m.ks);
    t->setSpecularExp(m.spec_exp);
}

std::shared_ptr<Material> make_material(const rt_scene_desc& d, const rt_material& m, const std::string& subdir)
{
    switch (m.kind) {
    case RT_MAT_MATTE: {
        auto p = std::make_shared<Matte>();
        p->setDiffuseColour(Colour{m.colour[0], m.colour[1], m.colour[2]});
        p->setDiffuseReflection(m.kd);
        p->setAmbientReflection(m.ka);
        return p;
    }
    case RT_MAT_SPECULAR: {
        auto p = std::make_shared<Specular>();
        p->setDiffuseColour(Colour{m.colour[0], m.colour[1], m.colour[2]});
        p->setDiffuseReflection(m.kd);
        p->setAmbientReflection(m.ka);
        p->setSpecularCoefficient(m.ks);
        p->setSpecularExp(m.spec_exp);
        return p;
    }
    case RT_MAT_TEXTURED: {
        tinyobj::material_t mt;
        mt.diffuse_texname = (m.texture >= 0 && (uint32_t)m.texture < d.num_textures)
                                 ? "tex" + std::to_string(m.texture) + ".tga" : std::string("missing.tga");
        auto p = std::make_shared<Textured>(mt, subdir);
        apply_textured_params(p.get(), m);
        return p;
    }
    case RT_MAT_MIRROR: {
        auto p = std::make_shared<Mirror>();
        p->setAmbientColour(Colour{m.colour[0], m.colour[1], m.colour[2]});
        p->setAmbientReflection(m.ka);
        p->mSpecularBRDF->setSpecularCoefficient(m.ks); // no public setter (built with -fno-access-control)
        p->mSpecularBRDF->setSpecularExp(m.spec_exp);
        return p;
    }
    case RT_MAT_REFRACTION: {
        auto p = std::make_shared<Refraction>();
        p->setIndexOfRefraction(m.ior);
        return p;
    }
    }
    return nullptr;
}

// ObjMesh for a run of rt_mesh records: one atlas::utils::Shape per Mesh, one vertex per index
// (the contract the reference reads: main.cpp:854-872).
bool make_objmesh(const rt_scene_desc& d, uint32_t first_mesh, uint32_t num_meshes,
                  atlas::utils::ObjMesh& out, std::vector<int>& mat_of_mesh, std::string& err)
{
    std::vector<int> slot_tex; // ObjMesh.materials slot -> desc material index
    for (uint32_t i = 0; i < num_meshes; ++i) {
        const rt_mesh& m = d.meshes[first_mesh + i];
        if (m.num_triangles == 0) { err = "empty mesh"; return false; }
        int mat = d.triangles[m.first_triangle].material;
        for (uint32_t k = 0; k < m.num_triangles; ++k)
            if (d.triangles[m.first_triangle + k].material != mat) { err = "mesh with mixed materials"; return false; }
        if (mat < 0 || (uint32_t)mat >= d.num_materials || d.materials[mat].kind != RT_MAT_TEXTURED) {
            err = "mesh material must be TEXTURED (the reference builds Textured from the OBJ materials)"; return false;
        }
        mat_of_mesh.push_back(mat);
        int slot = -1;
        for (size_t s = 0; s < slot_tex.size(); ++s) if (slot_tex[s] == mat) slot = (int)s;
        if (slot < 0) {
            slot = (int)slot_tex.size(); slot_tex.push_back(mat);
            tinyobj::material_t mt;
            int tex = d.materials[mat].texture;
            mt.diffuse_texname = (tex >= 0 && (uint32_t)tex < d.num_textures) ? "tex" + std::to_string(tex) + ".tga" : std::string("missing.tga");
            out.materials.push_back(mt);
        }
        atlas::utils::Shape sh;
        sh.vertices.resize((size_t)m.num_triangles * 3);
        sh.indices.resize((size_t)m.num_triangles * 3);
        for (uint32_t k = 0; k < m.num_triangles; ++k) {
            const rt_triangle& t = d.triangles[m.first_triangle + k];
            const float* vp[3] = {t.v0, t.v1, t.v2};
            const float* up[3] = {t.uv0, t.uv1, t.uv2};
            for (int c = 0; c < 3; ++c) {
                auto& v = sh.vertices[(size_t)k * 3 + c];
                v.position = P3(vp[c]);
                v.texCoord = P2(up[c]);
                sh.indices[(size_t)k * 3 + c] = (size_t)k * 3 + c;
            }
        }
        sh.materialIds.assign(i + 1, slot); // read as materialIds[shape index] (main.cpp:871)
        out.shapes.push_back(std::move(sh));
    }
    return true;
}

std::shared_ptr<Shape> shape_of(Built& b, const rt_shape_ref& r)
{
    switch (r.kind) {
    case RT_SHAPE_PLANE: return r.index < b.planes.size() ? b.planes[r.index] : nullptr;
    case RT_SHAPE_SPHERE: return r.index < b.spheres.size() ? b.spheres[r.index] : nullptr;
    case RT_SHAPE_TRIANGLE: return r.index < b.tris.size() ? b.tris[r.index] : nullptr;
    case RT_SHAPE_MESH: return r.index < b.meshes.size() ? b.meshes[r.index] : nullptr;
    case RT_SHAPE_MULTIMESH: return r.index < b.multimeshes.size() ? b.multimeshes[r.index] : nullptr;
    case RT_SHAPE_BVH: return r.index < b.bvhs.size() ? b.bvhs[r.index] : nullptr;
    }
    return nullptr;
}

bool build_world(const rt_scene_desc& d, const rt_params& p, const char* tmpdir, Built& b)
{
    // textures -> <tmpdir>/t/tex<i>.tga, decoded again by the reference's own stbi_load
    const std::string subdir = "t";
    modelRoot = std::string(tmpdir) + "/";
    {
        std::string cmd = "mkdir -p '" + modelRoot + subdir + "'";
        if (std::system(cmd.c_str()) != 0) { b.error = "cannot create texture dir"; return false; }
    }
    for (uint32_t i = 0; i < d.num_textures; ++i)
        if (d.textures[i].rgba && !write_tga(modelRoot + subdir + "/tex" + std::to_string(i) + ".tga", d.textures[i])) {
            b.error = "cannot write texture"; return false;
        }

    b.world = std::make_shared<World>();
    World& w = *b.world;
    w.width = (size_t)p.width; w.height = (size_t)p.height; w.background = {0, 0, 0};

    for (uint32_t i = 0; i < d.num_materials; ++i) {
        auto m = make_material(d, d.materials[i], subdir);
        if (!m) { b.error = "bad material kind"; return false; }
        b.materials.push_back(m);
    }
    auto mat = [&](int32_t i) -> std::shared_ptr<Material> {
        return (i >= 0 && (size_t)i < b.materials.size()) ? b.materials[i] : nullptr;
    };
    for (uint32_t i = 0; i < d.num_planes; ++i) {
        auto s = std::make_shared<Plane>(P3(d.planes[i].point), P3(d.planes[i].normal));
        s->setMaterial(mat(d.planes[i].material));
        b.planes.push_back(s);
    }
    for (uint32_t i = 0; i < d.num_spheres; ++i) {
        auto s = std::make_shared<Sphere>(P3(d.spheres[i].centre), d.spheres[i].radius);
        s->setMaterial(mat(d.spheres[i].material));
        b.spheres.push_back(s);
    }
    // which triangles belong to meshes
    std::vector<char> in_mesh(d.num_triangles, 0);
    for (uint32_t i = 0; i < d.num_meshes; ++i)
        for (uint32_t k = 0; k < d.meshes[i].num_triangles; ++k) in_mesh[d.meshes[i].first_triangle + k] = 1;
    b.tris.resize(d.num_triangles);
    for (uint32_t i = 0; i < d.num_triangles; ++i) {
        if (in_mesh[i]) continue;
        const rt_triangle& t = d.triangles[i];
        auto s = std::make_shared<Triangle>(P3(t.v0), P3(t.v1), P3(t.v2), P2(t.uv0), P2(t.uv1), P2(t.uv2));
        s->setMaterial(mat(t.material));
        b.tris[i] = s;
    }
    // which meshes belong to multimeshes
    std::vector<char> in_mm(d.num_meshes, 0);
    for (uint32_t i = 0; i < d.num_multimeshes; ++i)
        for (uint32_t k = 0; k < d.multimeshes[i].num_meshes; ++k) in_mm[d.multimeshes[i].first_mesh + k] = 1;
    b.meshes.resize(d.num_meshes);
    for (uint32_t i = 0; i < d.num_meshes; ++i) {
        if (in_mm[i]) continue;
        atlas::utils::ObjMesh om; std::vector<int> mm;
        if (!make_objmesh(d, i, 1, om, mm, b.error)) return false;
        auto s = std::make_shared<Mesh>(om, subdir, Vector{0, 0, 0});
        for (auto& tri : s->mMeshTriangles)
            if (auto tx = std::dynamic_pointer_cast<Textured>(tri.getMaterial())) apply_textured_params(tx.get(), d.materials[mm[0]]);
        b.meshes[i] = s;
    }
    for (uint32_t i = 0; i < d.num_multimeshes; ++i) {
        atlas::utils::ObjMesh om; std::vector<int> mm;
        if (!make_objmesh(d, d.multimeshes[i].first_mesh, d.multimeshes[i].num_meshes, om, mm, b.error)) return false;
        auto s = std::make_shared<MultiMesh>(om, subdir, Vector{0, 0, 0});
        for (size_t k = 0; k < s->meshes.size(); ++k)
            if (auto tx = std::dynamic_pointer_cast<Textured>(s->meshes[k].mMeshTriangles[0].getMaterial()))
                apply_textured_params(tx.get(), d.materials[mm[k]]);
        b.multimeshes.push_back(s);
    }
    for (uint32_t i = 0; i < d.num_bvhs; ++i) {
        auto acc = std::make_shared<BVHAccel>();
        for (uint32_t k = 0; k < d.bvhs[i].num_leaves; ++k) {
            const rt_shape_ref& r = d.bvh_leaves[d.bvhs[i].first_leaf + k];
            switch (r.kind) {
            case RT_SHAPE_SPHERE: acc->addShape(b.spheres.at(r.index)->getBoundingBoxPoints(), b.spheres[r.index]); break;
            case RT_SHAPE_TRIANGLE:
                if (!b.tris.at(r.index)) { b.error = "bvh leaf triangle belongs to a mesh"; return false; }
                acc->addShape(b.tris[r.index]->getBoundingBoxPoints(), b.tris[r.index]); break;
            case RT_SHAPE_MESH:
                if (!b.meshes.at(r.index)) { b.error = "bvh leaf mesh belongs to a multimesh"; return false; }
                acc->addShape(b.meshes[r.index]->getBoundingBoxPoints(), b.meshes[r.index]); break;
            case RT_SHAPE_MULTIMESH: acc->addShape(b.multimeshes.at(r.index)->getBoundingBoxPoints(), b.multimeshes[r.index]); break;
            default: b.error = "unsupported bvh leaf kind"; return false;
            }
        }
        acc->generateBVH();
        b.bvhs.push_back(acc);
    }
    for (uint32_t i = 0; i < d.num_scene; ++i) {
        auto s = shape_of(b, d.scene[i]);
        if (!s) { b.error = "bad scene shape reference"; return false; }
        w.scene.push_back(s);
    }
    for (uint32_t i = 0; i < d.num_lights; ++i) {
        const rt_light& l = d.lights[i];
        std::shared_ptr<Light> lp;
        if (l.kind == RT_LIGHT_POINT) lp = std::make_shared<PointLight>(P3(l.v));
        else lp = std::make_shared<Directional>(P3(l.v));
        lp->setColour(Colour{l.colour[0], l.colour[1], l.colour[2]});
        lp->scaleRadiance(l.radiance);
        w.lights.push_back(lp);
    }
    if (d.ambient.occlusion_samples > 0) {
        auto a = std::make_shared<Ambient>();
        a->mOcclusionSamples = (unsigned)d.ambient.occlusion_samples; // private, no setter (Raytracer.hpp:639-640)
        a->mOcclusionRayDist = d.ambient.occlusion_dist;
        w.ambient = a;
    } else {
        w.ambient = std::make_shared<FlatAmbient>();
    }
    w.ambient->setColour(Colour{d.ambient.colour[0], d.ambient.colour[1], d.ambient.colour[2]});
    w.ambient->scaleRadiance(d.ambient.radiance);
    return true;
}

// give every primitive a unique colour id (Shape::setColour) / clear it again
void paint_ids(const rt_scene_desc& d, Built& b, bool on)
{
    auto col = [&](uint32_t id) { return on ? Colour{(float)(id & 0xFFF), (float)((id >> 12) & 0xFFF), (float)(id >> 24)} : Colour{0, 0, 0}; };
    uint32_t base = 0;
    for (uint32_t i = 0; i < d.num_planes; ++i) b.planes[i]->setColour(col(base + i));
    base += d.num_planes;
    for (uint32_t i = 0; i < d.num_spheres; ++i) b.spheres[i]->setColour(col(base + i));
    base += d.num_spheres;
    for (uint32_t i = 0; i < d.num_triangles; ++i) if (b.tris[i]) b.tris[i]->setColour(col(base + i));
    for (uint32_t i = 0; i < d.num_meshes; ++i) if (b.meshes[i])
        for (uint32_t k = 0; k < d.meshes[i].num_triangles; ++k)
            b.meshes[i]->mMeshTriangles[k].setColour(col(base + d.meshes[i].first_triangle + k));
    for (uint32_t i = 0; i < d.num_multimeshes; ++i)
        for (uint32_t m = 0; m < d.multimeshes[i].num_meshes; ++m) {
            const rt_mesh& me = d.meshes[d.multimeshes[i].first_mesh + m];
            for (uint32_t k = 0; k < me.num_triangles; ++k)
                b.multimeshes[i]->meshes[m].mMeshTriangles[k].setColour(col(base + me.first_triangle + k));
        }
}

std::shared_ptr<Sampler> make_sampler(const rt_params& p)
{
    switch (p.sampler) {
    case RT_SAMPLER_REGULAR: return std::make_shared<Regular>(p.num_samples, p.num_sets);
    case RT_SAMPLER_RANDOM: return std::make_shared<Random>(p.num_samples, p.num_sets);
    default: return std::make_shared<Jitter>(p.num_samples, p.num_sets);
    }
}

// sample point j of a pixel from the counter hash (include/rt_rng.h), using the reference's own
// conversion formulas (Regular main.cpp:986-987, Jitter :1010-1011)
Point hashed_sample(const rt_params& p, int j)
{
    int n = static_cast<int>(glm::sqrt(static_cast<float>(p.num_samples)));
    if (n < 1) n = 1;
    int jj = j % (n * n);
    int pp = jj / n, q = jj % n;
    if (p.sampler == RT_SAMPLER_REGULAR) return Point{(q + 0.5f) / n, (pp + 0.5f) / n, 0.0f};
    if (p.sampler == RT_SAMPLER_RANDOM) {
        float a = (float)(harness_rand() >> 7) * (1.0f / 16777216.0f);
        float b = (float)(harness_rand() >> 7) * (1.0f / 16777216.0f);
        return Point{a, b, 0.0f};
    }
    float gridSize = 1.0f / n;
    float a = q * gridSize + (static_cast<float>(harness_rand()) / RAND_MAX * gridSize);
    float b = pp * gridSize + (static_cast<float>(harness_rand()) / RAND_MAX * gridSize);
    return Point{a, b, 0.0f};
}

} // namespace

extern "C" {

struct ref_options {
    int32_t rng_mode;   // 0: libc rand() + the reference's own Sampler;  1: counter hash for rand() and
                        // for the pixel sample points;  2: rand() == RAND_MAX/2 + the reference's Sampler
    int32_t nthreads;   // row-interleaved worker threads (each with its own Sampler in modes 0/2)
    int32_t shade;      // 0: primary closest-hit query only (ids/t/normal/uv), 1: + full shading into rgb
    const char* tmpdir; // where textures are written for the reference's stbi_load
};

struct ref_outputs {
    float* rgb;        // W*H*3, row-major r*width+c
    uint32_t* primid;  // W*H   sample 0's primary hit (0xFFFFFFFF = miss)
    float* t;          // W*H
    float* normal;     // W*H*3
    float* uv;         // W*H*2
    uint64_t rays;     // closest-hit queries over world->scene during shading passes
    double seconds;    // wall time of the shading pass
};

static std::string g_err;
const char* ref_last_error() { return g_err.c_str(); }

int ref_render(const rt_scene_desc* d, const rt_camera* cam, const rt_params* p, const ref_options* opt, ref_outputs* out)
{
    Built b;
    g_max_depth = p->max_depth;
    g_seed = p->seed;
    g_rng_mode = 0;
    if (!build_world(*d, *p, opt->tmpdir ? opt->tmpdir : "/tmp/ref_oracle_tmp", b)) { g_err = b.error; return -1; }
    auto world = b.world;
    Camera camera{P3(cam->eye), P3(cam->look_at), P3(cam->up), cam->frustum_dist};
    const int W = p->width, H = p->height;
    const int spp = p->num_samples;
    const bool want_ids = out->primid || out->t || out->normal || out->uv;

    // ---- pass 1: primary closest-hit query of sample 0 with colour ids painted on -------
    if (want_ids) {
        paint_ids(*d, b, true);
        g_rng_mode = opt->rng_mode == 0 ? 2 : opt->rng_mode; // ids use a reproducible sample 0
        auto sampler = make_sampler(*p);
        for (int r = p->y0; r < p->y1; ++r)
            for (int c = p->x0; c < p->x1; ++c) {
                tl_pixel = (uint32_t)(r * W + c); tl_sample = 0; tl_counter = 0;
                Point sp = (opt->rng_mode == 1) ? hashed_sample(*p, 0)
                         : (p->sampler == RT_SAMPLER_REGULAR && spp == 1 ? sampler->sampleUnitSquare() : Point{0.5f, 0.5f, 0.0f});
                atlas::math::Ray<atlas::math::Vector> ray;
                camera.calculateRay((c - (0.5f * world->width)) + sp.x, (r - (0.5f * world->height)) + sp.y, ray);
                ShadeRec sr{};
                sr.world = world;
                sr.t = std::numeric_limits<float>::max();
                bool hit{};
                for (auto const& obj : world->scene) hit |= obj->hit(ray, sr);
                size_t i = (size_t)r * W + c;
                if (out->primid) out->primid[i] = hit ? ((uint32_t)sr.color.x | ((uint32_t)sr.color.y << 12) | ((uint32_t)sr.color.z << 24)) : 0xFFFFFFFFu;
                if (out->t) out->t[i] = sr.t;
                if (out->normal) { out->normal[i * 3] = hit ? sr.normal.x : 0; out->normal[i * 3 + 1] = hit ? sr.normal.y : 0; out->normal[i * 3 + 2] = hit ? sr.normal.z : 0; }
                if (out->uv) { out->uv[i * 2] = hit ? sr.uvCoord.x : 0; out->uv[i * 2 + 1] = hit ? sr.uvCoord.y : 0; }
            }
        paint_ids(*d, b, false);
    }
    if (!opt->shade || !out->rgb) return 0;

    // ---- pass 2: full shading -----------------------------------------------------------
    g_rng_mode = opt->rng_mode;
    world->scene.push_back(std::make_shared<RayCounter>());
    const int nthreads = opt->nthreads > 0 ? opt->nthreads : 1;
    std::atomic<uint64_t> rays{0};
    const float avg = 1.0f / spp;
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        tl_rays = 0;
        auto sampler = make_sampler(*p); // per-thread: the reference shares one racy Sampler
        for (int r = p->y0 + tid; r < p->y1; r += nthreads)
            for (int c = p->x0; c < p->x1; ++c) {
                Colour pixelAverage{0, 0, 0};
                for (int j = 0; j < spp; ++j) {
                    tl_pixel = (uint32_t)(r * W + c); tl_sample = (uint32_t)j; tl_counter = 0;
                    Point sp = (opt->rng_mode == 1) ? hashed_sample(*p, j) : sampler->sampleUnitSquare();
                    tl_counter = 2;
                    atlas::math::Ray<atlas::math::Vector> ray;
                    camera.calculateRay((c - (0.5f * world->width)) + sp.x, (r - (0.5f * world->height)) + sp.y, ray);
                    ShadeRec sr{};
                    sr.world = world;
                    sr.t = std::numeric_limits<float>::max();
                    bool hit{};
                    for (auto const& obj : world->scene) hit |= obj->hit(ray, sr);
                    if (hit) pixelAverage += sr.material->shade(sr);
                }
                size_t i = ((size_t)r * W + c) * 3;
                out->rgb[i] = pixelAverage.r * avg; out->rgb[i + 1] = pixelAverage.g * avg; out->rgb[i + 2] = pixelAverage.b * avg;
            }
        rays += tl_rays;
    };
    if (nthreads == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < nthreads; ++i) th.emplace_back(worker, i);
        for (auto& t : th) t.join();
    }
    out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out->rays = rays.load();
    world->scene.pop_back();
    return 0;
}

// ---- known-answer probes of single reference functions ----------------------------------
int ref_geq(float a, float b) { return atlas::core::geq(a, b) ? 1 : 0; }

void ref_sample_hemisphere(float a, float b, float* out3)
{
    Vector v = sampleHemisphere(a, b);
    out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
}

void ref_rotate_to_space(const float* up3, const float* v3, float* out3)
{
    glm::mat4 m = rotationMatToSpace(P3(up3));
    Vector r = m * glm::vec4{P3(v3), 0};
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

void ref_camera_ray(const rt_camera* cam, float x, float y, float* o3, float* d3)
{
    Camera camera{P3(cam->eye), P3(cam->look_at), P3(cam->up), cam->frustum_dist};
    atlas::math::Ray<atlas::math::Vector> ray;
    camera.calculateRay(x, y, ray);
    o3[0] = ray.o.x; o3[1] = ray.o.y; o3[2] = ray.o.z;
    d3[0] = ray.d.x; d3[1] = ray.d.y; d3[2] = ray.d.z;
}

// BoundingVolumeBox built from n points (first = constructor argument), then intersects(ray)
int ref_box_intersects(const float* pts, int n, const float* o3, const float* d3)
{
    BoundingVolumeBox box(P3(pts));
    for (int i = 1; i < n; ++i) box.addVolumePoint(P3(pts + 3 * i));
    atlas::math::Ray<atlas::math::Vector> ray(P3(o3), P3(d3));
    return box.intersects(ray) ? 1 : 0;
}

// SpecularReflection::fn with the given normal / wo / wi
float ref_specular_fn(float ks, int exp, const float* n3, const float* wo3, const float* wi3)
{
    SpecularReflection s;
    s.setSpecularCoefficient(ks);
    s.setSpecularExp(exp);
    ShadeRec sr{};
    sr.normal = P3(n3);
    return s.fn(sr, P3(wo3), P3(wi3)).x;
}

// Transparency::refraction: returns direction, writes the child's index of refraction
void ref_refraction(float material_ior, float sr_ior, const float* d3, const float* n3, float* out3, float* enter_ior)
{
    Transparency tr(material_ior);
    ShadeRec sr{};
    sr.ray.d = P3(d3);
    sr.normal = P3(n3);
    sr.indexOfRefraction = sr_ior;
    float e = material_ior;
    Vector r = tr.refraction(sr, e);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
    *enter_ior = e;
}

// ImageTexture: decode a file with the reference's stbi_load and copy the texels out
int ref_load_texture(const char* path, int* w, int* h, uint8_t* out_rgba, size_t cap)
{
    ImageTexture tex{std::string(path)};
    if (!tex.mSet) return -1;
    *w = tex.mWidth; *h = tex.mHeight;
    size_t need = (size_t)tex.mWidth * tex.mHeight * 4;
    if (out_rgba && cap >= need) std::memcpy(out_rgba, tex.mImage, need);
    return 0;
}

// saveToFile's float -> 8 bit conversion (main.cpp:1847-1849)
void ref_quantize(const float* rgb, size_t n, uint8_t* out)
{
    for (size_t i = 0; i < n; ++i) out[i] = static_cast<unsigned char>(rgb[i] * 255);
}

// The reference's own saveToFile (main.cpp:1837-1857: truncating conversion + stbi_write_bmp) on a caller's
// float image, so that the device encoder (rt_encode_bmp24) is checked against bytes the reference wrote.
int ref_save_bmp(const float* rgb, int width, int height, const char* path)
{
    std::vector<Colour> image((size_t)width * height);
    for (size_t i = 0; i < image.size(); ++i) image[i] = Colour{rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]};
    saveToFile(std::string(path), (std::size_t)width, (std::size_t)height, image);
    return 0;
}

// getBoundingBoxPoints() of shape #index of the given kind (RT_SHAPE_SPHERE / TRIANGLE / MESH / MULTIMESH): the six
// support points x-, x+, y-, y+, z-, z+ (main.cpp:303-313) -> out[18].  Returns the number of points (6), < 0 on error.
int ref_bbox_points(const rt_scene_desc* d, const rt_params* p, const char* tmpdir, int kind, uint32_t index, float* out)
{
    Built b;
    if (!build_world(*d, *p, tmpdir ? tmpdir : "/tmp/ref_oracle_tmp", b)) { g_err = b.error; return -1; }
    std::vector<Point> pts;
    switch (kind) {
    case RT_SHAPE_SPHERE: if (index >= b.spheres.size()) return -1; pts = b.spheres[index]->getBoundingBoxPoints(); break;
    case RT_SHAPE_TRIANGLE: if (index >= b.tris.size() || !b.tris[index]) return -1; pts = b.tris[index]->getBoundingBoxPoints(); break;
    case RT_SHAPE_MESH: if (index >= b.meshes.size() || !b.meshes[index]) return -1; pts = b.meshes[index]->getBoundingBoxPoints(); break;
    case RT_SHAPE_MULTIMESH: if (index >= b.multimeshes.size()) return -1; pts = b.multimeshes[index]->getBoundingBoxPoints(); break;
    default: return -1;
    }
    for (size_t i = 0; i < pts.size() && i < 6; ++i) { out[i * 3] = pts[i].x; out[i * 3 + 1] = pts[i].y; out[i * 3 + 2] = pts[i].z; }
    return (int)pts.size();
}

// BVHAccel hierarchy dump for a desc's bvh #which: per node {leaf, left, right, min[3], max[3]}
int ref_bvh_dump(const rt_scene_desc* d, const rt_params* p, const char* tmpdir, uint32_t which, float* out, int cap_nodes)
{
    Built b;
    if (!build_world(*d, *p, tmpdir ? tmpdir : "/tmp/ref_oracle_tmp", b)) { g_err = b.error; return -1; }
    if (which >= b.bvhs.size()) return -1;
    auto& h = b.bvhs[which]->mHeirarchy;
    int n = (int)h.size();
    for (int i = 0; i < n && i < cap_nodes; ++i) {
        float* o = out + (size_t)i * 9;
        o[0] = h[i].leaf ? 1.0f : 0.0f;
        o[1] = h[i].leaf ? -1.0f : (float)h[i].left;
        o[2] = h[i].leaf ? -1.0f : (float)h[i].right;
        o[3] = h[i].nodeBvb.xMin; o[4] = h[i].nodeBvb.yMin; o[5] = h[i].nodeBvb.zMin;
        o[6] = h[i].nodeBvb.xMax; o[7] = h[i].nodeBvb.yMax; o[8] = h[i].nodeBvb.zMax;
    }
    return n;
}

} // extern "C"
