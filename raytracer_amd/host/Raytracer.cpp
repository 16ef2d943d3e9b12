// raytracer_amd/host/Raytracer.cpp — implementation of the C++ host mirror: scene flattening into
// the arrays of include/rt_amd.h, the render calls, the thread pool, the BMP writer and the small
// OBJ / PPM / TGA readers.  Links against librt_amd.so only.
#include "Raytracer.hpp"
#include <climits>
#include <cstdlib>
#include <algorithm>
#include <cstring>

#include <algorithm>

namespace {

[[noreturn]] void fail(const std::string& what, rt_ctx* ctx = nullptr)
{
    throw std::runtime_error(what + ": " + rt_last_error(ctx));
}

// ---- flattening World -> rt_scene_desc arrays -----------------------------------------------------
struct Flat {
    std::vector<rt_plane> planes; std::vector<rt_sphere> spheres; std::vector<rt_triangle> triangles;
    std::vector<rt_mesh> meshes; std::vector<rt_multimesh> multimeshes; std::vector<rt_bvh> bvhs;
    std::vector<rt_shape_ref> leaves, scene; std::vector<rt_material> materials; std::vector<rt_texture> textures;
    std::vector<rt_light> lights; rt_ambient ambient{};
    std::map<const Material*, int> mat_ids; std::map<const ImageTexture*, int> tex_ids; std::map<const Shape*, rt_shape_ref> seen;

    int tex_id(const std::shared_ptr<ImageTexture>& t)
    {
        if (!t) return -1;
        auto it = tex_ids.find(t.get());
        if (it != tex_ids.end()) return it->second;
        int id = (int)textures.size();
        textures.push_back(rt_texture{t->mWidth, t->mHeight, t->mSet ? t->mImage.data() : nullptr});
        return tex_ids[t.get()] = id;
    }
    int mat_id(const std::shared_ptr<Material>& m)
    {
        if (!m) throw std::runtime_error("shape without a material");
        auto it = mat_ids.find(m.get());
        if (it != mat_ids.end()) return it->second;
        rt_material c = m->c;
        c.texture = tex_id(m->texture);
        int id = (int)materials.size();
        materials.push_back(c);
        return mat_ids[m.get()] = id;
    }
    uint32_t add_mesh(const Mesh& m)
    {
        uint32_t first = (uint32_t)triangles.size();
        int mat = mat_id(m.mMaterial);
        for (rt_triangle t : m.mTriangles) { t.material = mat; triangles.push_back(t); }
        meshes.push_back(rt_mesh{first, (uint32_t)m.mTriangles.size()});
        return (uint32_t)meshes.size() - 1;
    }
    rt_shape_ref ref_of(const std::shared_ptr<Shape>& s)
    {
        auto it = seen.find(s.get());
        if (it != seen.end()) return it->second;
        rt_shape_ref r{};
        if (auto p = dynamic_cast<Plane*>(s.get())) {
            planes.push_back(rt_plane{{p->mPoint.x, p->mPoint.y, p->mPoint.z}, {p->mNormal.x, p->mNormal.y, p->mNormal.z}, mat_id(p->mMaterial)});
            r = {RT_SHAPE_PLANE, (uint32_t)planes.size() - 1};
        } else if (auto sp = dynamic_cast<Sphere*>(s.get())) {
            spheres.push_back(rt_sphere{{sp->mCentre.x, sp->mCentre.y, sp->mCentre.z}, sp->mRadius, mat_id(sp->mMaterial)});
            r = {RT_SHAPE_SPHERE, (uint32_t)spheres.size() - 1};
        } else if (auto t = dynamic_cast<Triangle*>(s.get())) {
            rt_triangle c{{t->mV0.x, t->mV0.y, t->mV0.z}, {t->mV1.x, t->mV1.y, t->mV1.z}, {t->mV2.x, t->mV2.y, t->mV2.z},
                          {t->mUV0.x, t->mUV0.y}, {t->mUV1.x, t->mUV1.y}, {t->mUV2.x, t->mUV2.y}, mat_id(t->mMaterial)};
            triangles.push_back(c);
            r = {RT_SHAPE_TRIANGLE, (uint32_t)triangles.size() - 1};
        } else if (auto me = dynamic_cast<Mesh*>(s.get())) {
            r = {RT_SHAPE_MESH, add_mesh(*me)};
        } else if (auto mm = dynamic_cast<MultiMesh*>(s.get())) {
            uint32_t first = (uint32_t)meshes.size();
            for (auto& m : mm->meshes) add_mesh(m);
            multimeshes.push_back(rt_multimesh{first, (uint32_t)mm->meshes.size()});
            r = {RT_SHAPE_MULTIMESH, (uint32_t)multimeshes.size() - 1};
        } else if (auto b = dynamic_cast<BVHAccel*>(s.get())) {
            std::vector<rt_shape_ref> mine;
            for (auto& l : b->mLeaves) mine.push_back(ref_of(l));
            bvhs.push_back(rt_bvh{(uint32_t)leaves.size(), (uint32_t)mine.size()});
            leaves.insert(leaves.end(), mine.begin(), mine.end());
            r = {RT_SHAPE_BVH, (uint32_t)bvhs.size() - 1};
        } else {
            throw std::runtime_error("unknown Shape subclass");
        }
        return seen[s.get()] = r;
    }
};

// The GPUs of this process: one context (+ the uploaded scene of the World last rendered) per device of the
// list.  RT_GPU_DEVICES="0,1,2,3" names the devices (a device may appear more than once: contexts then share
// it, which is how the tests run the multi-GPU path on a one-GPU box); RT_NUM_GPUS=N means devices 0..N-1;
// default: device 0.  With more than one context a frame is rendered by rt_render_multi: tiles t % N == i on
// context i, slabs gathered on the first device (the reference's pool threads, main.cpp:1803-1827, become GPUs).
struct Gpu {
    std::vector<rt_ctx*> ctx; std::vector<rt_scene*> scene;
    const World* world = nullptr; unsigned long revision = ~0ul;
    std::mutex mu;
    ~Gpu() { for (auto s : scene) if (s) rt_scene_destroy(s); for (auto c : ctx) if (c) rt_shutdown(c); }
};
Gpu& gpu() { static Gpu g; return g; }

std::vector<int> device_list()
{
    std::vector<int> d;
    if (const char* e = std::getenv("RT_GPU_DEVICES")) {
        std::istringstream ss(e);
        std::string tok;
        while (std::getline(ss, tok, ',')) if (!tok.empty()) d.push_back(std::atoi(tok.c_str()));
    } else if (const char* n = std::getenv("RT_NUM_GPUS")) {
        for (int i = 0; i < std::max(1, std::atoi(n)); ++i) d.push_back(i);
    }
    if (d.empty()) d.push_back(0);
    return d;
}

void ensure_uploaded(Gpu& g, const World& w)
{
    if (g.ctx.empty()) {
        for (int dev : device_list()) {
            rt_ctx* c = nullptr;
            if (rt_init(dev, &c) != RT_OK) fail("rt_init(" + std::to_string(dev) + ")");
            g.ctx.push_back(c);
        }
        g.scene.assign(g.ctx.size(), nullptr);
    }
    if (g.world == &w && g.revision == w.mRevision && g.scene[0]) return;
    for (auto& s : g.scene) if (s) { rt_scene_destroy(s); s = nullptr; }
    Flat f;
    for (auto& s : w.scene) f.scene.push_back(f.ref_of(s));
    for (auto& l : w.lights) f.lights.push_back(rt_light{l->mKind, {l->mV.x, l->mV.y, l->mV.z}, {l->mColour.x, l->mColour.y, l->mColour.z}, l->mRadiance});
    if (w.ambient) f.ambient = rt_ambient{{w.ambient->mColour.x, w.ambient->mColour.y, w.ambient->mColour.z}, w.ambient->mRadiance,
                                          w.ambient->mOcclusionSamples, w.ambient->mOcclusionRayDist};
    rt_scene_desc d{};
    d.planes = f.planes.data(); d.num_planes = (uint32_t)f.planes.size();
    d.spheres = f.spheres.data(); d.num_spheres = (uint32_t)f.spheres.size();
    d.triangles = f.triangles.data(); d.num_triangles = (uint32_t)f.triangles.size();
    d.meshes = f.meshes.data(); d.num_meshes = (uint32_t)f.meshes.size();
    d.multimeshes = f.multimeshes.data(); d.num_multimeshes = (uint32_t)f.multimeshes.size();
    d.bvhs = f.bvhs.data(); d.num_bvhs = (uint32_t)f.bvhs.size();
    d.bvh_leaves = f.leaves.data(); d.num_bvh_leaves = (uint32_t)f.leaves.size();
    d.scene = f.scene.data(); d.num_scene = (uint32_t)f.scene.size();
    d.materials = f.materials.data(); d.num_materials = (uint32_t)f.materials.size();
    d.textures = f.textures.data(); d.num_textures = (uint32_t)f.textures.size();
    d.lights = f.lights.data(); d.num_lights = (uint32_t)f.lights.size();
    d.ambient = f.ambient;
    // the scene is replicated: every context gets its own copy (commits run side by side)
    std::vector<std::string> errs(g.ctx.size());
    auto upload = [&](size_t i) {
        if (rt_scene_create(g.ctx[i], &g.scene[i]) != RT_OK) { errs[i] = std::string("rt_scene_create: ") + rt_last_error(g.ctx[i]); return; }
        if (rt_scene_set_desc(g.scene[i], &d) != RT_OK) { errs[i] = std::string("rt_scene_set_desc: ") + rt_last_error(g.ctx[i]); return; }
        if (rt_scene_commit(g.scene[i]) != RT_OK) errs[i] = std::string("rt_scene_commit: ") + rt_last_error(g.ctx[i]);
    };
    std::vector<std::thread> th;
    for (size_t i = 1; i < g.ctx.size(); ++i) th.emplace_back(upload, i);
    upload(0);
    for (auto& t : th) t.join();
    for (auto& e : errs) if (!e.empty()) throw std::runtime_error(e);
    g.world = &w; g.revision = w.mRevision;
}

void render_rect(int y0, int y1, int x0, int x1, rt_camera cam, std::shared_ptr<World> const& world, rt_stats* stats)
{
    Gpu& g = gpu();
    std::lock_guard<std::mutex> lk(g.mu);   // pool threads calling raytrace() concurrently serialise here
    ensure_uploaded(g, *world);
    if (world->image.size() != world->width * world->height) world->image.assign(world->width * world->height, Colour{0, 0, 0});
    rt_params p;
    rt_params_default(&p);
    p.width = (int)world->width; p.height = (int)world->height;
    p.x0 = x0; p.x1 = x1; p.y0 = y0; p.y1 = y1;
    if (world->sampler) { p.num_samples = world->sampler->mNumSamples; p.num_sets = world->sampler->mNumSets; p.sampler = world->sampler->mKind; }
    p.max_depth = world->maxDepth; p.light_attenuation = (float)lightAttenuationFactor; p.seed = world->seed;
    if (stats) p.flags = RT_FLAG_TIMING;
    static_assert(sizeof(Colour) == 3 * sizeof(float), "Colour must be three packed floats");
    float* image = reinterpret_cast<float*>(world->image.data());
    if (g.ctx.size() == 1) {
        if (rt_render(g.scene[0], &cam, &p, image, nullptr, nullptr, stats) != RT_OK) fail("rt_render", g.ctx[0]);
    } else {
        std::vector<rt_stats> per(g.ctx.size());
        if (rt_render_multi(g.scene.data(), (int)g.scene.size(), &cam, &p, image, stats ? per.data() : nullptr) != RT_OK) fail("rt_render_multi", g.ctx[0]);
        if (stats) {   // the frame's record: counters summed over the ranks, times = the slowest rank's
            *stats = per[0];
            for (size_t r = 1; r < per.size(); ++r) {
                const rt_stats& q = per[r];
                stats->rays_primary += q.rays_primary; stats->rays_shadow += q.rays_shadow; stats->rays_ao += q.rays_ao;
                stats->rays_bounce += q.rays_bounce; stats->rays_skipped += q.rays_skipped;
                for (int k = 0; k < 3; ++k) { stats->nodes[k] += q.nodes[k]; stats->tris[k] += q.tris[k]; stats->gates[k] += q.gates[k]; stats->bvh_rays[k] += q.bvh_rays[k]; }
                if (q.ms_total > stats->ms_total) stats->ms_total = q.ms_total;
                for (int k = 0; k < RT_K_COUNT; ++k) { if (q.ms_kernel[k] > stats->ms_kernel[k]) stats->ms_kernel[k] = q.ms_kernel[k]; stats->launches[k] += q.launches[k]; }
                stats->batches += q.batches;
            }
        }
    }
}

} // namespace

void raytrace(int beginBlockY, int endBlockY, int beginBlockX, int endBlockX, Camera const& camera, std::shared_ptr<World> const& world)
{
    if (ThreadPool* pool = ThreadPool::current()) {   // inside a pool job: coalesced at waitDone()
        pool->defer(ThreadPool::DeferredTile{beginBlockY, endBlockY, beginBlockX, endBlockX, camera.toC(), world});
        return;
    }
    render_rect(beginBlockY, endBlockY, beginBlockX, endBlockX, camera.toC(), world, nullptr);
}

void render(Camera const& camera, std::shared_ptr<World> const& world, rt_stats* stats)
{
    render_rect(0, (int)world->height, 0, (int)world->width, camera.toC(), world, stats);
}

// ---- meshes ----------------------------------------------------------------------------------------
Mesh::Mesh(atlas::utils::Shape const& shape, unsigned int matIndex, std::vector<std::shared_ptr<Textured>> const& loadedMaterials,
           std::string, Vector offset)
{
    if (shape.indices.size() % 3 != 0) return;
    for (size_t k = 0; k + 2 < shape.indices.size(); k += 3) {
        rt_triangle t{};
        const atlas::utils::Vertex* v[3] = {&shape.vertices[shape.indices[k]], &shape.vertices[shape.indices[k + 1]], &shape.vertices[shape.indices[k + 2]]};
        float* pos[3] = {t.v0, t.v1, t.v2}; float* uv[3] = {t.uv0, t.uv1, t.uv2};
        for (int c = 0; c < 3; ++c) {
            pos[c][0] = v[c]->position.x + offset.x; pos[c][1] = v[c]->position.y + offset.y; pos[c][2] = v[c]->position.z + offset.z;
            uv[c][0] = v[c]->texCoord.x; uv[c][1] = v[c]->texCoord.y;
        }
        mTriangles.push_back(t);
    }
    if (!shape.materialIds.empty()) {   // reference main.cpp:868-872 (indexed by shape index)
        int id = shape.materialIds[std::min<size_t>(matIndex, shape.materialIds.size() - 1)];
        if (id >= 0 && (size_t)id < loadedMaterials.size()) mMaterial = loadedMaterials[id];
    }
}

std::vector<Point> Mesh::getBoundingBoxPoints() const
{
    if (mTriangles.empty()) return {};
    auto pt = [](const float* v) { return Point{v[0], v[1], v[2]}; };
    BoundingVolumeBox b(pt(mTriangles[0].v0));
    for (const rt_triangle& t : mTriangles) { b.addVolumePoint(pt(t.v0)); b.addVolumePoint(pt(t.v1)); b.addVolumePoint(pt(t.v2)); }
    return b.getBoundingBoxPoints();
}

std::vector<Point> MultiMesh::getBoundingBoxPoints() const
{
    if (meshes.empty()) return {};
    const std::vector<Point> first = meshes[0].getBoundingBoxPoints();
    if (first.empty()) return {};
    BoundingVolumeBox b(first[0]);
    for (const Mesh& m : meshes) for (const Point& q : m.getBoundingBoxPoints()) b.addVolumePoint(q);
    return b.getBoundingBoxPoints();
}

MultiMesh::MultiMesh(atlas::utils::ObjMesh const& mesh, std::string modelSubDirName, Vector offset)
{
    std::vector<std::shared_ptr<Textured>> loaded;
    for (auto& m : mesh.materials) loaded.push_back(std::make_shared<Textured>(m, modelSubDirName));
    for (unsigned int i = 0; i < mesh.shapes.size(); ++i) meshes.push_back(Mesh(mesh.shapes[i], i, loaded, modelSubDirName, offset));
}

// ---- thread pool (reference main.cpp:80-143; completion uses a counter + predicate) ----------------------
ThreadPool::ThreadPool(unsigned int const& numThreads)
{
    for (unsigned i = 0; i < std::max(1u, numThreads); ++i) mThreads.emplace_back(&ThreadPool::loopFunction, this);
}
namespace { thread_local ThreadPool* tl_pool = nullptr; }
ThreadPool* ThreadPool::current() { return tl_pool; }
void ThreadPool::defer(DeferredTile t)
{
    std::lock_guard<std::mutex> lk(mDeferredMutex);
    mDeferred.push_back(std::move(t));
}

// Renders what the finished jobs recorded.  Tiles of one (world, camera) whose rectangles are pairwise
// disjoint and exactly cover their bounding rectangle become one launch; anything else is rendered
// rectangle by rectangle, in submission order.
void ThreadPool::flushDeferred()
{
    std::vector<DeferredTile> all;
    {
        std::lock_guard<std::mutex> lk(mDeferredMutex);
        all.swap(mDeferred);
    }
    std::vector<char> done(all.size(), 0);
    for (size_t i = 0; i < all.size(); ++i) {
        if (done[i]) continue;
        std::vector<size_t> grp;
        for (size_t j = i; j < all.size(); ++j)
            if (!done[j] && all[j].world == all[i].world && std::memcmp(&all[j].camera, &all[i].camera, sizeof(rt_camera)) == 0) { grp.push_back(j); done[j] = 1; }
        int y0 = INT_MAX, y1 = INT_MIN, x0 = INT_MAX, x1 = INT_MIN;
        long long area = 0;
        bool ok = true;
        for (size_t j : grp) {
            const DeferredTile& t = all[j];
            if (t.y1 <= t.y0 || t.x1 <= t.x0) { ok = false; break; }
            y0 = std::min(y0, t.y0); y1 = std::max(y1, t.y1); x0 = std::min(x0, t.x0); x1 = std::max(x1, t.x1);
            area += (long long)(t.y1 - t.y0) * (t.x1 - t.x0);
        }
        ok = ok && area == (long long)(y1 - y0) * (x1 - x0);
        if (ok) {   // equal area + pairwise disjoint = exact tiling (sweep over the tiles sorted by row)
            std::vector<size_t> byrow(grp);
            std::sort(byrow.begin(), byrow.end(), [&](size_t a, size_t b) { return all[a].y0 < all[b].y0; });
            for (size_t a = 0; a < byrow.size() && ok; ++a)
                for (size_t b = a + 1; b < byrow.size() && all[byrow[b]].y0 < all[byrow[a]].y1; ++b) {
                    const DeferredTile &p = all[byrow[a]], &q = all[byrow[b]];
                    if (p.x0 < q.x1 && q.x0 < p.x1) { ok = false; break; }
                }
        }
        if (ok) { render_rect(y0, y1, x0, x1, all[i].camera, all[i].world, nullptr); ++mLaunches; }
        else for (size_t j : grp) { render_rect(all[j].y0, all[j].y1, all[j].x0, all[j].x1, all[j].camera, all[j].world, nullptr); ++mLaunches; }
    }
}

void ThreadPool::loopFunction()
{
    tl_pool = this;
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> lock(mQueueMutex);
            mCondition.wait(lock, [&] { return !mQueue.empty() || mTerminatePool; });
            if (mTerminatePool && mQueue.empty()) return;
            job = std::move(mQueue.front());
            mQueue.pop();
        }
        job();
        {
            std::unique_lock<std::mutex> lock(mQueueMutex);
            if (--mPending == 0) mComplCondition.notify_all();
        }
    }
}
void ThreadPool::startJob(std::function<void()> job)
{
    {
        std::unique_lock<std::mutex> lock(mQueueMutex);
        mQueue.push(std::move(job));
        ++mPending;
    }
    mCondition.notify_one();
}
void ThreadPool::waitDone()
{
    {
        std::unique_lock<std::mutex> lock(mQueueMutex);
        mComplCondition.wait(lock, [&] { return mPending == 0; });
    }
    flushDeferred();
}
ThreadPool::~ThreadPool()
{
    waitDone();   // nothing recorded is lost when the caller never waited
    {
        std::unique_lock<std::mutex> lock(mQueueMutex);
        mTerminatePool = true;
    }
    mCondition.notify_all();
    for (auto& t : mThreads) t.join();
}

// ---- saveToFile (reference main.cpp:1837-1857 + stbi_write_bmp's file layout) ------------------------------
void saveToFile(std::string const& filename, std::size_t width, std::size_t height, std::vector<Colour> const& image)
{
    const size_t pad = (4 - (width * 3) % 4) % 4, row = width * 3 + pad;
    std::vector<unsigned char> out(54 + row * height, 0);
    auto put32 = [&](size_t o, uint32_t v) { out[o] = v & 255; out[o + 1] = (v >> 8) & 255; out[o + 2] = (v >> 16) & 255; out[o + 3] = (v >> 24) & 255; };
    out[0] = 'B'; out[1] = 'M';
    put32(2, (uint32_t)out.size()); put32(10, 54); put32(14, 40); put32(18, (uint32_t)width); put32(22, (uint32_t)height);
    out[26] = 1; out[28] = 24;
    for (size_t r = 0; r < height; ++r)
        for (size_t c = 0; c < width; ++c) {
            const Colour& p = image[r * width + c];
            unsigned char* o = &out[54 + (height - 1 - r) * row + c * 3];
            o[2] = static_cast<unsigned char>(p.r * 255); o[1] = static_cast<unsigned char>(p.g * 255); o[0] = static_cast<unsigned char>(p.b * 255);
        }
    std::ofstream f(filename, std::ios::binary);
    f.write(reinterpret_cast<const char*>(out.data()), (std::streamsize)out.size());
}

// ---- small readers -------------------------------------------------------------------------------------
// ---- PNG -> RGBA8 (what stbi_load(path, .., STBI_rgb_alpha) hands the reference, main.cpp:147-165) --------
// DEFLATE (RFC 1951) with canonical-Huffman decoding by code length counts, zlib framing (RFC 1950), PNG
// chunks / filters (ISO 15948).  Non-interlaced images of every colour type; 16-bit samples keep their
// high byte, sub-byte grey samples are scaled to 0..255, palettes use tRNS for alpha.
namespace {
struct BitReader {
    const unsigned char* p; size_t n, pos = 0; uint32_t buf = 0; int cnt = 0; bool err = false;
    uint32_t bits(int k)
    {
        while (cnt < k) { if (pos >= n) { err = true; return 0; } buf |= (uint32_t)p[pos++] << cnt; cnt += 8; }
        uint32_t v = k ? (buf & ((1u << k) - 1u)) : 0u;
        buf >>= k; cnt -= k;
        return v;
    }
};
struct Huffman {
    uint16_t count[16] = {0}, symbol[288] = {0};
    void build(const unsigned char* len, int n)
    {
        std::fill(count, count + 16, (uint16_t)0);
        for (int i = 0; i < n; ++i) count[len[i]]++;
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; ++i) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        count[0] = 0;
    }
    int decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; ++l) {
            code |= (int)br.bits(1);
            if (br.err) return -1;
            int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};
bool inflate_zlib(const std::vector<unsigned char>& z, std::vector<unsigned char>& out)
{
    if (z.size() < 6 || (z[0] & 15) != 8 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 32)) return false;
    BitReader br{z.data() + 2, z.size() - 2};
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (int last = 0; !last;) {
        last = (int)br.bits(1);
        const int type = (int)br.bits(2);
        if (br.err) return false;
        if (type == 0) {
            br.buf = 0; br.cnt = 0;
            if (br.pos + 4 > br.n) return false;
            const unsigned len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xFFFFu) != nlen || br.pos + len > br.n) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
            continue;
        }
        if (type == 3) return false;
        Huffman lit, dist;
        unsigned char lens[320];
        if (type == 1) {
            for (int i = 0; i < 288; ++i) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            lit.build(lens, 288);
            for (int i = 0; i < 30; ++i) lens[i] = 5;
            dist.build(lens, 30);
        } else {
            const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
            static const unsigned char order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            if (br.err || nlen > 286 || ndist > 30) return false;
            unsigned char cl[19] = {0};
            for (int i = 0; i < ncode; ++i) cl[order[i]] = (unsigned char)br.bits(3);
            Huffman lencode; lencode.build(cl, 19);
            for (int i = 0; i < nlen + ndist;) {
                int sym = lencode.decode(br);
                if (sym < 0) return false;
                if (sym < 16) { lens[i++] = (unsigned char)sym; continue; }
                int rep, val = 0;
                if (sym == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (int)br.bits(2); }
                else if (sym == 17) rep = 3 + (int)br.bits(3);
                else rep = 11 + (int)br.bits(7);
                if (i + rep > nlen + ndist) return false;
                while (rep--) lens[i++] = (unsigned char)val;
            }
            if (lens[256] == 0) return false;
            lit.build(lens, nlen);
            dist.build(lens + nlen, ndist);
        }
        for (;;) {
            int sym = lit.decode(br);
            if (sym < 0 || br.err) return false;
            if (sym < 256) { out.push_back((unsigned char)sym); continue; }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) return false;
            const int len = lbase[sym] + (int)br.bits(lext[sym]);
            const int ds = dist.decode(br);
            if (ds < 0 || ds >= 30) return false;
            const size_t d = dbase[ds] + br.bits(dext[ds]);
            if (br.err || d > out.size()) return false;
            for (int k = 0; k < len; ++k) out.push_back(out[out.size() - d]);
        }
    }
    return true;
}

bool decode_png(const std::vector<unsigned char>& f, int& W, int& H, std::vector<unsigned char>& rgba)
{
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (f.size() < 8 || std::memcmp(f.data(), sig, 8) != 0) return false;
    auto be32 = [&](size_t o) { return ((uint32_t)f[o] << 24) | ((uint32_t)f[o + 1] << 16) | ((uint32_t)f[o + 2] << 8) | f[o + 3]; };
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte, trns;
    for (size_t o = 8; o + 12 <= f.size();) {
        const uint32_t len = be32(o);
        const std::string tag(f.begin() + o + 4, f.begin() + o + 8);
        if (o + 12 + (size_t)len > f.size()) return false;
        const unsigned char* d = f.data() + o + 8;
        if (tag == "IHDR" && len >= 13) { w = be32(o + 8); h = be32(o + 12); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (tag == "PLTE") plte.assign(d, d + len);
        else if (tag == "tRNS") trns.assign(d, d + len);
        else if (tag == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (tag == "IEND") break;
        o += 12 + (size_t)len;
    }
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!w || !h || w > 32768 || h > 32768 || !channels || interlace != 0) return false;
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return false;
    if (ctype == 3 && (depth == 16 || plte.size() < 3)) return false;
    std::vector<unsigned char> raw;
    if (!inflate_zlib(idat, raw)) return false;
    const size_t bpp = std::max<size_t>(1, (size_t)channels * depth / 8), stride = ((size_t)w * channels * depth + 7) / 8;
    if (raw.size() < (stride + 1) * h) return false;
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    rgba.assign((size_t)w * h * 4, 255);
    for (uint32_t y = 0; y < h; ++y) {
        const unsigned char* src = &raw[(stride + 1) * y];
        const int ft = src[0];
        if (ft > 4) return false;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            cur[i] = (unsigned char)(src[1 + i] + pred);
        }
        for (uint32_t x = 0; x < w; ++x) {
            unsigned char* o = &rgba[((size_t)y * w + x) * 4];
            auto sample = [&](int ch) -> unsigned {   // 8-bit view of sample ch of pixel x (raw index for palettes)
                if (depth == 8) return cur[(size_t)x * channels + ch];
                if (depth == 16) return cur[((size_t)x * channels + ch) * 2];
                const unsigned bit = x * (unsigned)depth, v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                return ctype == 3 ? v : v * (255u / ((1u << depth) - 1u));
            };
            if (ctype == 0) { o[0] = o[1] = o[2] = (unsigned char)sample(0); }
            else if (ctype == 2) { o[0] = (unsigned char)sample(0); o[1] = (unsigned char)sample(1); o[2] = (unsigned char)sample(2); }
            else if (ctype == 3) { const unsigned i = sample(0); if (i * 3 + 2 < plte.size()) { o[0] = plte[i * 3]; o[1] = plte[i * 3 + 1]; o[2] = plte[i * 3 + 2]; } else { o[0] = o[1] = o[2] = 0; } o[3] = i < trns.size() ? trns[i] : 255; }
            else if (ctype == 4) { o[0] = o[1] = o[2] = (unsigned char)sample(0); o[3] = (unsigned char)sample(1); }
            else { o[0] = (unsigned char)sample(0); o[1] = (unsigned char)sample(1); o[2] = (unsigned char)sample(2); o[3] = (unsigned char)sample(3); }
        }
        prev.swap(cur);
    }
    W = (int)w; H = (int)h;
    return true;
}
} // namespace

ImageTexture::ImageTexture(std::string const& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return;
    std::vector<unsigned char> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (d.size() > 8 && d[0] == 0x89 && d[1] == 'P' && d[2] == 'N' && d[3] == 'G') {
        int w = 0, h = 0;
        if (decode_png(d, w, h, mImage)) { mWidth = w; mHeight = h; mSet = true; } else mImage.clear();
        return;
    }
    if (d.size() > 2 && d[0] == 'P' && d[1] == '6') {
        std::istringstream hs(std::string(d.begin(), d.begin() + std::min<size_t>(d.size(), 64)));
        std::string magic; int w = 0, h = 0, mx = 0;
        hs >> magic >> w >> h >> mx;
        size_t off = (size_t)hs.tellg() + 1;
        if (w <= 0 || h <= 0 || mx != 255 || d.size() < off + (size_t)w * h * 3) return;
        mImage.resize((size_t)w * h * 4);
        for (size_t i = 0; i < (size_t)w * h; ++i) { mImage[i * 4] = d[off + i * 3]; mImage[i * 4 + 1] = d[off + i * 3 + 1]; mImage[i * 4 + 2] = d[off + i * 3 + 2]; mImage[i * 4 + 3] = 255; }
        mWidth = w; mHeight = h; mSet = true;
    } else if (d.size() > 18 && d[2] == 2 && (d[16] == 24 || d[16] == 32)) {   // uncompressed true-colour TGA
        int w = d[12] | (d[13] << 8), h = d[14] | (d[15] << 8), bpp = d[16] / 8;
        size_t off = 18 + d[0];
        if (w <= 0 || h <= 0 || d.size() < off + (size_t)w * h * bpp) return;
        bool top = (d[17] & 0x20) != 0;
        mImage.resize((size_t)w * h * 4);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const unsigned char* s = &d[off + ((size_t)(top ? y : h - 1 - y) * w + x) * bpp];
                unsigned char* o = &mImage[((size_t)y * w + x) * 4];
                o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; o[3] = bpp == 4 ? s[3] : 255;
            }
        mWidth = w; mHeight = h; mSet = true;
    }
}

std::optional<atlas::utils::ObjMesh> atlas::utils::loadObjMesh(std::string const& filename, std::string const& materialPath)
{
    std::ifstream f(filename);
    if (!f) return std::nullopt;
    std::string dir = materialPath.empty() ? filename.substr(0, filename.find_last_of("/\\") + 1) : materialPath;
    ObjMesh out;
    std::vector<math::Point> pos; std::vector<math::Point2> tex;
    std::map<std::string, int> mat_index;
    int cur_mat = -1;
    auto shape = [&]() -> Shape& { if (out.shapes.empty()) out.shapes.emplace_back(); return out.shapes.back(); };
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag)) continue;
        if (tag == "v") { math::Point p; ls >> p.x >> p.y >> p.z; pos.push_back(p); }
        else if (tag == "vt") { math::Point2 t; ls >> t.x >> t.y; tex.push_back(t); }
        else if (tag == "o" || tag == "g") { if (out.shapes.empty() || !out.shapes.back().indices.empty()) out.shapes.emplace_back(); }
        else if (tag == "mtllib") {
            std::string name; ls >> name;
            std::ifstream m(dir + name);
            std::string ml;
            while (std::getline(m, ml)) {
                std::istringstream ms(ml);
                std::string t2;
                if (!(ms >> t2)) continue;
                if (t2 == "newmtl") { tinyobj::material_t mt; ms >> mt.name; mat_index[mt.name] = (int)out.materials.size(); out.materials.push_back(mt); }
                else if (t2 == "map_Kd" && !out.materials.empty()) { ms >> out.materials.back().diffuse_texname; std::replace(out.materials.back().diffuse_texname.begin(), out.materials.back().diffuse_texname.end(), '\\', '/'); }
            }
        } else if (tag == "usemtl") { std::string name; ls >> name; auto it = mat_index.find(name); cur_mat = it == mat_index.end() ? -1 : it->second; }
        else if (tag == "f") {
            Shape& s = shape();
            std::vector<std::size_t> face;
            std::string vtx;
            while (ls >> vtx) {
                int vi = 0, ti = 0;
                std::sscanf(vtx.c_str(), "%d/%d", &vi, &ti);
                Vertex v;
                if (vi < 0) vi = (int)pos.size() + vi + 1;
                if (ti < 0) ti = (int)tex.size() + ti + 1;
                if (vi >= 1 && (size_t)vi <= pos.size()) v.position = pos[vi - 1];
                if (ti >= 1 && (size_t)ti <= tex.size()) v.texCoord = tex[ti - 1];
                v.index = s.vertices.size();
                s.vertices.push_back(v);
                face.push_back(v.index);
            }
            for (size_t k = 1; k + 1 < face.size(); ++k) {   // fan-triangulate
                s.indices.push_back(face[0]); s.indices.push_back(face[k]); s.indices.push_back(face[k + 1]);
                s.materialIds.push_back(cur_mat);
            }
        }
    }
    // the reference reads materialIds[shape index]: make every entry of a shape its material
    for (auto& s : out.shapes) if (!s.materialIds.empty()) { int m = s.materialIds[0]; s.materialIds.assign(std::max(s.materialIds.size(), out.shapes.size()), m); }
    return out;
}
