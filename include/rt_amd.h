/*
 * rt_amd.h — C ABI of the MI355X (gfx950) renderer that replaces the per-pixel
 * ray/scene intersection + shading path of andrewyung/Raytracer.
 *
 * The reference has no FFI: its "API" is the C++ class surface of Raytracer.hpp used by
 * main() (reference Raytracer/Raytracer.hpp:75-641, Raytracer/main.cpp:1620-1857).  This header
 * is the flat, plain-pointer boundary a maintainer binds that class surface to
 * (see INTEGRATION.md).  Every entry point cites the reference interface it stands in for.
 *
 * Conventions
 *   - every function returns an rt_status (0 = ok, negative = error); no C++ exception and no
 *     torch type crosses this boundary; rt_last_error() gives the text of the last failure.
 *   - the caller owns every host buffer it passes; the library copies during rt_scene_set_*.
 *   - "device" pointers are HIP device pointers on the context's GPU (e.g. tensor.data_ptr()).
 *   - image layout is row-major  rgb[(r*width + c)*3 + ch]  (the reference writes
 *     image[r*height+c], main.cpp:1656, which is only correct for square images — SURVEY Q14).
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_AMD_VERSION 1

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument / inconsistent scene description            */
    RT_ERR_UNSUPPORTED = -2, /* describable by the reference API but not by this path    */
    RT_ERR_HIP = -3,         /* a HIP runtime call failed (text in rt_last_error)        */
    RT_ERR_NOMEM = -4,
    RT_ERR_STATE = -5        /* e.g. rt_render before rt_scene_commit                    */
} rt_status;

/* ---- scene description: the flattened form of Raytracer.hpp's object graph ------------- */

/* Shape kinds (reference Raytracer.hpp:255-413) */
enum {
    RT_SHAPE_PLANE = 0,     /* Plane(point, normal)                      Raytracer.hpp:331 */
    RT_SHAPE_SPHERE = 1,    /* Sphere(centre, radius)                    Raytracer.hpp:255 */
    RT_SHAPE_TRIANGLE = 2,  /* Triangle(p0,p1,p2,uv0,uv1,uv2)            Raytracer.hpp:273 */
    RT_SHAPE_MESH = 3,      /* Mesh: AABB gate + linear triangle run     Raytracer.hpp:313 */
    RT_SHAPE_MULTIMESH = 4, /* MultiMesh: AABB gate + run of Mesh        Raytracer.hpp:296 */
    RT_SHAPE_BVH = 5        /* BVHAccel over shapes                      Raytracer.hpp:381 */
};

/* Material kinds (reference Raytracer.hpp:516-600) */
enum {
    RT_MAT_MATTE = 0,
    RT_MAT_SPECULAR = 1,
    RT_MAT_TEXTURED = 2,
    RT_MAT_MIRROR = 3,
    RT_MAT_REFRACTION = 4 /* the only kind with Material::mTransparent == true (main.cpp:1599) */
};

enum { RT_LIGHT_POINT = 0, RT_LIGHT_DIRECTIONAL = 1 };          /* Raytracer.hpp:602-627 */
enum { RT_SAMPLER_REGULAR = 0, RT_SAMPLER_JITTER = 1, RT_SAMPLER_RANDOM = 2 }; /* :415-437 */

typedef struct rt_plane    { float point[3]; float normal[3]; int32_t material; } rt_plane;
typedef struct rt_sphere   { float centre[3]; float radius; int32_t material; } rt_sphere;
/* Mesh triangles are stored already translated by the MultiMesh offset, exactly as the
 * reference bakes the offset into its vertex copies (main.cpp:859-864). */
typedef struct rt_triangle {
    float v0[3], v1[3], v2[3];
    float uv0[2], uv1[2], uv2[2];
    int32_t material;
} rt_triangle;
typedef struct rt_mesh      { uint32_t first_triangle, num_triangles; } rt_mesh;
typedef struct rt_multimesh { uint32_t first_mesh, num_meshes; } rt_multimesh;
typedef struct rt_shape_ref { int32_t kind; uint32_t index; } rt_shape_ref;
/* One BVHAccel: its leaves, in BVHAccel::addShape order (main.cpp:1763-1779), are
 * bvh_leaves[first_leaf .. first_leaf+num_leaves).  The library rebuilds the reference's
 * bottom-up nearest-centroid hierarchy (main.cpp:341-423) from them. */
typedef struct rt_bvh       { uint32_t first_leaf, num_leaves; } rt_bvh;

/* One record covers all five material classes; unused fields are ignored.
 *   MATTE      colour, kd, ka                      (Raytracer.hpp:516-533)
 *   SPECULAR   colour, kd, ka, ks, spec_exp        (:535-556; defaults kd=ka=0.5 ks=0.3 exp=16)
 *   TEXTURED   texture, ka, ks, spec_exp           (:585-600)
 *   MIRROR     colour (= ambient colour), ka, ks, spec_exp   (:558-571)
 *   REFRACTION ior                                 (:573-583; default 1.7, :499)        */
typedef struct rt_material {
    int32_t kind;
    float colour[3];
    float kd, ka, ks;
    int32_t spec_exp;
    float ior;
    int32_t texture; /* index into textures[], -1 = none (ImageTexture::mSet == false) */
} rt_material;

/* Decoded RGBA8 texels as stbi_load(..., STBI_rgb_alpha) yields them (main.cpp:147-165). */
typedef struct rt_texture { int32_t width, height; const uint8_t* rgba; } rt_texture;

/* v = position for POINT, direction for DIRECTIONAL (normalised by the library like
 * Directional::setDirection, main.cpp:1315-1318). */
typedef struct rt_light { int32_t kind; float v[3]; float colour[3]; float radiance; } rt_light;

/* world->ambient.  occlusion_samples = n of the n x n ambient-occlusion grid
 * (Ambient::mOcclusionSamples, Raytracer.hpp:639; reference value 4); 0 selects the plain
 * Light::L = radiance*colour (main.cpp:551-554), i.e. "AO off". */
typedef struct rt_ambient {
    float colour[3];
    float radiance;
    int32_t occlusion_samples;
    float occlusion_dist; /* Ambient::mOcclusionRayDist, Raytracer.hpp:640 (1000) */
} rt_ambient;

/* Camera(position, lookAt, up, frustrumDist)  (Raytracer.hpp:150, main.cpp:1407-1424) */
typedef struct rt_camera { float eye[3], look_at[3], up[3]; float frustum_dist; } rt_camera;

/* Whole-scene aggregate (used by rt_scene_set_desc, by the oracle and by the reference
 * harness under oracle/). */
typedef struct rt_scene_desc {
    const rt_plane* planes;           uint32_t num_planes;
    const rt_sphere* spheres;         uint32_t num_spheres;
    const rt_triangle* triangles;     uint32_t num_triangles;
    const rt_mesh* meshes;            uint32_t num_meshes;
    const rt_multimesh* multimeshes;  uint32_t num_multimeshes;
    const rt_bvh* bvhs;               uint32_t num_bvhs;
    const rt_shape_ref* bvh_leaves;   uint32_t num_bvh_leaves;
    const rt_shape_ref* scene;        uint32_t num_scene;   /* world->scene, in order */
    const rt_material* materials;     uint32_t num_materials;
    const rt_texture* textures;       uint32_t num_textures;
    const rt_light* lights;           uint32_t num_lights;
    rt_ambient ambient;
} rt_scene_desc;

/* Render parameters.  Defaults (rt_params_default) reproduce the reference constants:
 * maxBounceDepth=3, lightAttenuationFactor=500000 (Raytracer.hpp:52-53), Jitter(4,83),
 * 1000x1000 (main.cpp:1680-1683). */
typedef struct rt_params {
    int32_t width, height;        /* world->width / world->height                          */
    int32_t x0, y0, x1, y1;       /* raytrace(beginY=y0,endY=y1,beginX=x0,endX=x1,..) rect  */
    int32_t num_samples;          /* Sampler numSamples = samples per pixel                 */
    int32_t num_sets;             /* Sampler numSets (kept for API parity; the counter-based
                                     generator does not need precomputed sets)              */
    int32_t sampler;              /* RT_SAMPLER_*                                           */
    int32_t max_depth;            /* maxBounceDepth                                         */
    float   light_attenuation;    /* lightAttenuationFactor                                 */
    uint32_t seed;                /* key of the counter-based generator (include/rt_rng.h)  */
    int32_t tile_rank, tile_world;/* multi-GPU: this call renders the 64x64 tiles t with
                                     t % tile_world == tile_rank (0,1 = everything)         */
    uint32_t batch_paths;         /* paths in flight per wavefront batch, 0 = default: as
                                     many as the queues' 31-bit ray ids allow, at most 85 % of
                                     the free device memory (a 4K 16-spp frame with 4x4 AO is
                                     one batch of ~208 GB); set it to bound the memory taken  */
    uint32_t flags;               /* RT_FLAG_*                                              */
} rt_params;

#define RT_FLAG_STATS      1u  /* run the counting variants of the traversal kernels        */
#define RT_FLAG_TIMING     2u  /* bracket every launch with HIP events (rt_stats.ms_*)      */
#define RT_TILE 64             /* edge of the tile that is the unit of the multi-GPU split  */

enum { RT_K_RAYGEN = 0, RT_K_TRACE, RT_K_TRACE_BVH, RT_K_PREP, RT_K_SHADOW, RT_K_AO, RT_K_AO_BVH, RT_K_SHADE, RT_K_RESOLVE, RT_K_COUNT };

typedef struct rt_stats {
    uint64_t rays_primary, rays_shadow, rays_ao, rays_bounce; /* closest-hit queries traced */
    /* RT_FLAG_STATS only: BVH nodes fetched / triangles tested / reference-level gate boxes
     * tested, per ray class {closest(primary+bounce), shadow, ao} */
    uint64_t nodes[3], tris[3], gates[3];
    uint64_t bvh_rays[3];            /* rays of each class that had at least one Mesh gate pass and were
                                        traversed by the compacted triangle-BVH kernel                */
    double   ms_total;               /* hipEvent time of the whole render on its stream      */
    double   ms_kernel[RT_K_COUNT];  /* RT_FLAG_TIMING: summed per kernel class              */
    uint32_t launches[RT_K_COUNT];
    uint32_t batches;
    /* Closest-hit queries the reference traces (its unit: one `for obj in world->scene: obj->hit()` loop,
     * main.cpp:1645-1648, 51-56, 1390-1393, 1457-1460, 1548-1551) that this path proves unnecessary and does
     * not trace: the shadow ray of a light the surface faces away from (the reference calls shadowed() before
     * it tests n.wi > 0, main.cpp:1227, 1281) and the n x n occlusion rays of a surface whose ka is 0
     * (main.cpp:1497; the reference's mirror, main.cpp:1714).  rays_primary + rays_shadow + rays_ao + rays_bounce
     * + rays_skipped == the reference's own query count for the same frame, exactly. */
    uint64_t rays_skipped;
} rt_stats;

typedef struct rt_ctx rt_ctx;
typedef struct rt_scene rt_scene;

/* ---- lifetime ------------------------------------------------------------------------- */
/* Binds the context to HIP device `device` (one context per GPU, one process per GPU).
 * Fails with RT_ERR_HIP when no gfx950 device / kernel image is usable: there is no CPU
 * fallback behind this ABI.  [replaces: ThreadPool(numThreads) set-up, main.cpp:1803-1805] */
int rt_init(int device, rt_ctx** out);
int rt_shutdown(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx); /* ctx may be NULL: last error of rt_init */
int rt_version(void);
void rt_params_default(rt_params* p);

/* ---- scene upload --------------------------------------------------------------------- */
/* [replaces: building World / world->scene / BVHAccel in main.cpp:1678-1801] */
int rt_scene_create(rt_ctx* ctx, rt_scene** out);
int rt_scene_set_planes(rt_scene* s, const rt_plane* v, size_t n);            /* Plane ctor, main.cpp:921 */
int rt_scene_set_spheres(rt_scene* s, const rt_sphere* v, size_t n);          /* Sphere ctor, :567 */
int rt_scene_set_triangles(rt_scene* s, const rt_triangle* v, size_t n);      /* Triangle ctor, :637 */
int rt_scene_set_meshes(rt_scene* s, const rt_mesh* v, size_t n);             /* Mesh ctor, :851-882 */
int rt_scene_set_multimeshes(rt_scene* s, const rt_multimesh* v, size_t n);   /* MultiMesh ctor, :742-766 */
int rt_scene_set_bvhs(rt_scene* s, const rt_bvh* v, size_t n,
                      const rt_shape_ref* leaves, size_t num_leaves);         /* addShape, :323-339 */
int rt_scene_set_scene(rt_scene* s, const rt_shape_ref* v, size_t n);         /* world->scene.push_back, :1782-1789 */
int rt_scene_set_materials(rt_scene* s, const rt_material* v, size_t n);      /* material setters, Raytracer.hpp:522-593 */
int rt_scene_set_textures(rt_scene* s, const rt_texture* v, size_t n);        /* ImageTexture ctor, main.cpp:147 */
int rt_scene_set_lights(rt_scene* s, const rt_light* v, size_t n);            /* world->lights, :1792-1795 */
int rt_scene_set_ambient(rt_scene* s, const rt_ambient* a);                   /* world->ambient, :1791,1797-1798 */
int rt_scene_set_desc(rt_scene* s, const rt_scene_desc* d);                   /* all of the above in one call */
/* Validates, rebuilds the reference's shape-level hierarchy and gate boxes
 * (BVHAccel::generateBVH main.cpp:341-423, BoundingVolumeBox :191-243), builds one triangle
 * BVH per Mesh, and uploads everything to the context's GPU. */
int rt_scene_commit(rt_scene* s);
int rt_scene_destroy(rt_scene* s);

/* ---- render --------------------------------------------------------------------------- */
/* [replaces: the job loop over raytrace() main.cpp:1809-1827 + raytrace() itself :1620-1665]
 * Blocking; serialised per context.  Outputs (each nullable) are host buffers:
 *   rgb     width*height*3 floats, pixels outside [x0,x1)x[y0,y1) or not owned by
 *           (tile_rank,tile_world) are left untouched
 *   primid  width*height: id of the primitive hit by sample 0's camera ray
 *           (planes, then spheres, then triangles, in description order; 0xFFFFFFFF = miss)
 *   t       width*height: that hit's ray parameter (FLT_MAX on a miss)                    */
int rt_render(rt_scene* s, const rt_camera* cam, const rt_params* p,
              float* rgb, uint32_t* primid, float* t, rt_stats* stats);

/* Same, writing to device memory and returning as soon as the work is queued on `stream`
 * (a hipStream_t passed as void*; NULL = the context's own stream, in which case the call
 * also waits).  With tile_world > 1, d_rgb is this rank's compact tile slab
 * [rt_owned_tiles()][RT_TILE*RT_TILE][3] instead of a full frame.
 *
 * Stream contract of every entry point that takes a `stream` (rt_render_device, rt_unpack_tiles,
 * rt_quantize_rgb8, rt_encode_bmp24):
 *   - stream != NULL: the work is queued on that stream, behind whatever the caller queued there
 *     (fills, copies, collectives that produce or consume the buffers); the call does not wait.
 *   - stream == NULL: the work runs on the context's own stream, ordered after everything the
 *     caller had queued on the legacy default (null) stream when the call was made, and the call
 *     returns when it is complete.  Work the caller queued on any OTHER stream is NOT ordered
 *     against it: pass that stream (e.g. torch.cuda.current_stream().cuda_stream) instead.
 *   - one context carries one render at a time: a render on another stream waits (on the device)
 *     for the previous one, since they share the context's ray queues and counters. */
int rt_render_device(rt_scene* s, const rt_camera* cam, const rt_params* p,
                     float* d_rgb, uint32_t* d_primid, float* d_t, rt_stats* stats, void* stream);

/* Allocation-free variant of rt_render_device.  It has NO counterpart in the reference; it exists for
 * callers that cannot allocate device memory: code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  Same arguments and results as rt_render_device (without the statistics), but the
 * wavefront pipeline's scratch (ray / hit records, path, shadow and BVH work queues, AO distances,
 * per-level shading terms, queue counters) lives in a fixed RT_NOALLOC_SCRATCH_BYTES array inside the
 * library's device image instead of hipMalloc'ed memory, and the call makes no hipMalloc / hipFree /
 * event or stream creation and no read-back.  The frame is rendered in as many batches as that array
 * requires (more, smaller batches than rt_render_device: slower, same pixels, bit for bit).
 *   - limit: one RT_TILE x RT_TILE tile x num_samples paths must fit; with 4x4 ambient occlusion and one
 *     light that is num_samples <= 32 (about 1.7 KB of queues per path + 143 MB of fixed spill stacks).
 *     Larger requests fail with RT_ERR_UNSUPPORTED and a message naming the sizes; flags must be 0.
 *   - each GPU holds one such array for the life of the process (every process that loads the library pays
 *     RT_NOALLOC_SCRATCH_BYTES per GPU it initialises), shared by every context and call on that GPU.  Eager
 *     calls are ordered by the library, across contexts and streams (one "last user" event per device).
 *   - under stream capture (pass the capturing stream; NULL is rejected by HIP's capture rules) the call adds
 *     only kernel and memset nodes and waits on / records no event: order replays of the graph against other
 *     users of that GPU's scratch yourself — overlapping ones race.
 *   - stream semantics otherwise as for rt_render_device. */
#ifndef RT_NOALLOC_SCRATCH_BYTES
#define RT_NOALLOC_SCRATCH_BYTES (384ull << 20)
#endif
int rt_render_device_noalloc(rt_scene* s, const rt_camera* cam, const rt_params* p,
                             float* d_rgb, uint32_t* d_primid, float* d_t, void* stream);

/* Multi-GPU render from ONE process (the C++ drop-in's ThreadPool::waitDone, main.cpp:1809-1827: the reference's
 * pool threads become GPUs).  scenes[0..num_scenes) hold the SAME committed scene description, each on a context
 * of its own (normally one per GPU: rt_init(i)); rank i renders the RT_TILE tiles t with t % num_scenes == i on
 * its context's stream from a host thread of its own, sends its compact slab to scenes[0]'s GPU (one peer copy
 * per rank: each peer has its own xGMI link to GPU 0), where the slabs are scattered into the frame
 * (rt_unpack_tiles' kernel) and copied to `rgb` (host, width*height*3 floats; pixels outside the render rectangle
 * are left untouched).  The frame is bit-identical to a single rt_render.  p->tile_rank / tile_world are
 * ignored.  stats: NULL or num_scenes records (one per rank).  Contexts may share a device (how the tests run it
 * on a one-GPU box).  The torch.distributed path of bench.py (one process per GPU, RCCL gather) is the
 * alternative for Python hosts. */
int rt_render_multi(rt_scene* const* scenes, int num_scenes, const rt_camera* cam, const rt_params* p,
                    float* rgb, rt_stats* stats);

/* Number of RT_TILE tiles of the render rectangle owned by (rank, world). */
int rt_owned_tiles(const rt_params* p, int rank, int world, uint32_t* out);

/* Scatters one rank's tile slab (as gathered over RCCL) into a full row-major frame on the
 * context's GPU.  [no reference counterpart: the reference's threads share one std::vector,
 * main.cpp:1656] */
int rt_unpack_tiles(rt_ctx* ctx, const float* d_slab, const rt_params* p, int rank, int world,
                    float* d_frame, void* stream);

/* Where rt_scene_commit builds the triangle BVHs of the Meshes (no reference counterpart:
 * Mesh::hit scans all triangles, main.cpp:890-912).  RT_BUILDER_HOST_SAH (default): binned SAH on the host,
 * multi-threaded — the better trees.  RT_BUILDER_GPU_LBVH: Morton-code radix trees built on the device — the
 * faster build; falls back to the host builder when a tree would be deeper than the traversal stack.  A BVH
 * only culls: both give bit-identical frames.  rt_scene_builder_used reports what the last commit used. */
enum { RT_BUILDER_HOST_SAH = 0, RT_BUILDER_GPU_LBVH = 1 };
int rt_scene_set_builder(rt_scene* scene, int builder);
int rt_scene_builder_used(rt_scene* scene, int* builder, double* build_seconds);

/* Self-test of the hand-written device arithmetic behind Ambient::L's sample directions (main.cpp:1361-1393, :4-19): for n
 * pseudo-random inputs it compares, bit for bit and on the device, (0) the exact square root without small-input rescaling
 * against sqrtf, (1, 2) sin / cos with scalar-operand fma steps against the same polynomial written plainly, (3) the
 * division-free r / RAND_MAX against the fp64 division; mismatches[k] counts the inputs that differ (all 0 when sound).
 * [no reference counterpart: a check of this library's own arithmetic, for its GPU tests] */
int rt_selftest_device_math(rt_ctx* ctx, uint64_t n, uint64_t seed, uint64_t mismatches[4]);

/* Resolve step of saveToFile (main.cpp:1842-1850): uchar(pixel*255), truncating, on device. */
int rt_quantize_rgb8(rt_ctx* ctx, const float* d_rgb, size_t num_pixels, uint8_t* d_rgb8, void* stream);

/* saveToFile in one device pass (main.cpp:1837-1857 + stb_image_write.h's stbi_write_bmp layout):
 * truncating 8-bit conversion and the complete 24-bit BMP file image (54-byte header, BGR,
 * bottom-up rows padded to 4 bytes) into d_file, which holds rt_bmp24_bytes(width, height) bytes.
 * The caller copies those bytes to the host (a quarter of the f32 frame) and writes them out. */
size_t rt_bmp24_bytes(int width, int height);
int rt_encode_bmp24(rt_ctx* ctx, const float* d_rgb, int width, int height, uint8_t* d_file, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
