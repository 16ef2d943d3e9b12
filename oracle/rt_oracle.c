/*
 * oracle/rt_oracle.c — TEST INFRASTRUCTURE: a plain-C CPU restatement of the reference's
 * per-pixel ray/scene intersection + shading path (andrewyung/Raytracer, Raytracer/main.cpp).
 *
 * It is the checker for the HIP kernels (tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg) and is never imported, linked or executed by the product (raytracer_amd/).
 * It is itself pinned against the real reference compiled from /root/reference
 * (oracle/ref_harness.cpp -> oracle/_ref/libref_oracle.so): tests/test_oracle_ref.py requires
 * bit-identical t / normal / uv / primitive ids / RGB in every deterministic mode, and
 * tests/golden/ holds vectors produced by that reference build.
 *
 * Every function names the reference lines it follows.  Float operation order follows the
 * vendored glm the reference uses (include/glm/detail/func_geometric.inl: dot = x+y+z left to
 * right :54-61, cross :74-85, normalize = v * (1/sqrt(dot)) :88-96 with inversesqrt =
 * 1/sqrt func_exponential.inl:131-135, reflect = I - N*dot(N,I)*2 :110-116).  Build with
 * -ffp-contract=off (x86-64 g++ -O2 emits no FMA for the reference either).
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rt_amd.h"
#include "rt_rng.h"

/* ---- vector helpers (glm operation order) ---------------------------------------------- */
typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vld(const float* p) { return V(p[0], p[1], p[2]); }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 x, v3 y) { return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
static inline v3 vnormalize(v3 v) { return vscale(v, 1.0f / sqrtf(vdot(v, v))); }
static inline float vlength(v3 v) { return sqrtf(vdot(v, v)); }
static inline v3 vreflect(v3 I, v3 N) { return vsub(I, vscale(vscale(N, vdot(N, I)), 2.0f)); }
static inline float clamp01(float x) { float m = (x < 0.0f) ? 0.0f : x; return (1.0f < m) ? 1.0f : m; } /* glm::clamp = min(max(x,lo),hi) */
static inline float clampf(float x, float lo, float hi) { float m = (x < lo) ? lo : x; return (hi < m) ? hi : m; }

/* ---- rand() hook: same three modes as oracle/ref_harness.cpp ----------------------------- */
typedef struct {
    int mode;                 /* 0 libc rand(), 1 counter hash, 2 constant RAND_MAX/2 */
    uint32_t seed, pixel, sample, counter;
} Rng;

static int rng_rand(Rng* g)
{
    if (g->mode == 1) return (int)rt_rand31(g->seed, g->pixel, g->sample, g->counter++);
    if (g->mode == 2) return RAND_MAX / 2;
    return rand();
}

/* ---- scene ------------------------------------------------------------------------------- */
typedef struct { float mn[3], mx[3]; } Box; /* BoundingVolumeBox: xMin..zMax (main.cpp:229-243) */

typedef struct {
    int leaf;
    rt_shape_ref shape;
    Box box;
    uint32_t left, right;
    int accessed;
} BvhNode; /* BVHAccel::BVHNode (Raytracer.hpp:398-408) */

typedef struct { BvhNode* nodes; uint32_t n; } Bvh;

typedef struct {
    const rt_scene_desc* d;
    Box* mesh_box;      /* per rt_mesh       (main.cpp:875-881) */
    Box* mm_box;        /* per rt_multimesh  (main.cpp:757-765) */
    Bvh* bvh;           /* per rt_bvh        (main.cpp:323-423) */
    v3* light_dir;      /* Directional::setDirection normalises (main.cpp:1315-1318) */
    uint32_t prim_base_sphere, prim_base_tri;
    int max_depth;
    float light_atten;
} Scene;

typedef struct {
    v3 color; float t; v3 normal; v3 ro, rd; int material; float u, v;
    unsigned depth; float ior; int shadowRay;
    uint32_t prim;
} ShadeRec; /* Raytracer.hpp:89-102 (+ prim: bookkeeping for the id output) */

static void rec_init(ShadeRec* r)
{
    memset(r, 0, sizeof *r);
    r->material = -1; r->ior = 1.0f; r->prim = 0xFFFFFFFFu;
}

/* BoundingVolumeBox(start) + addVolumePoint(...) only ever yield the min/max of the points
 * (main.cpp:191-243); the anchor points' other coordinates never reach a result. */
static void box_start(Box* b, v3 p) { b->mn[0] = b->mx[0] = p.x; b->mn[1] = b->mx[1] = p.y; b->mn[2] = b->mx[2] = p.z; }
static void box_add(Box* b, v3 p)
{
    if (p.x < b->mn[0]) b->mn[0] = p.x;
    if (p.x > b->mx[0]) b->mx[0] = p.x;
    if (p.y < b->mn[1]) b->mn[1] = p.y;
    if (p.y > b->mx[1]) b->mx[1] = p.y;
    if (p.z < b->mn[2]) b->mn[2] = p.z;
    if (p.z > b->mx[2]) b->mx[2] = p.z;
}
static void box_union(Box* o, const Box* a, const Box* b)
{
    *o = *a;
    box_add(o, V(b->mn[0], b->mn[1], b->mn[2]));
    box_add(o, V(b->mx[0], b->mx[1], b->mx[2]));
}
/* BoundingVolumeBox::updateMinMaxCentroid (main.cpp:240-242) */
static v3 box_centroid(const Box* b) { return V((b->mn[0] + b->mx[0]) / 2.0f, (b->mn[1] + b->mx[1]) / 2.0f, (b->mn[2] + b->mx[2]) / 2.0f); }

/* atlas::core::geq (include/atlas/core/Float.hpp:17-21,49-52) */
static int geq(float a, float b)
{
    if (a > b) return 1;
    float scale = (fabsf(a) + fabsf(b)) / 2.0f;
    return fabsf(a - b) <= FLT_EPSILON * scale;
}

/* Plane::intersectRay (main.cpp:943-957) */
static int plane_intersect(v3 point, v3 normal, v3 o, v3 d, float* tMin)
{
    float denom = vdot(normal, d);
    if (fabsf(denom) < 0.00001f) return 0;
    float t = vdot(vsub(point, o), normal) / denom;
    if (t >= 0 && t < *tMin) { *tMin = t; return 1; }
    return 0;
}

/* Plane::hit (main.cpp:924-941) */
static int plane_hit(const Scene* s, uint32_t i, v3 o, v3 d, ShadeRec* sr)
{
    const rt_plane* p = &s->d->planes[i];
    float t = FLT_MAX;
    int inter = plane_intersect(vld(p->point), vld(p->normal), o, d, &t);
    if (inter && t < sr->t) {
        sr->normal = vld(p->normal); sr->t = t; sr->ro = o; sr->rd = d;
        sr->color = V(0, 0, 0); sr->material = p->material; sr->prim = i;
        return 1;
    }
    return 0;
}

/* BoundingVolumeBox::intersects (main.cpp:245-296): six Plane::hit, then strict in-face tests */
static int box_face(int axis, float plane_c, const Box* b, v3 o, v3 d)
{
    v3 n = V(axis == 0, axis == 1, axis == 2);
    v3 p = V(axis == 0 ? plane_c : 0.0f, axis == 1 ? plane_c : 0.0f, axis == 2 ? plane_c : 0.0f);
    /* the anchor's other coordinates only ever meet a zero normal component; use the ray
     * origin's so that (p-o)*0 is an exact 0 as it is for any finite anchor */
    if (axis != 0) p.x = o.x;
    if (axis != 1) p.y = o.y;
    if (axis != 2) p.z = o.z;
    float t = FLT_MAX;
    if (!plane_intersect(p, n, o, d, &t)) return 0;
    v3 h = vadd(o, vscale(d, t));
    if (axis == 0) return h.y > b->mn[1] && h.y < b->mx[1] && h.z > b->mn[2] && h.z < b->mx[2];
    if (axis == 1) return h.x > b->mn[0] && h.x < b->mx[0] && h.z > b->mn[2] && h.z < b->mx[2];
    return h.x > b->mn[0] && h.x < b->mx[0] && h.y > b->mn[1] && h.y < b->mx[1];
}
static int box_intersects(const Box* b, v3 o, v3 d)
{
    if (box_face(0, b->mn[0], b, o, d)) return 1;
    if (box_face(0, b->mx[0], b, o, d)) return 1;
    if (box_face(1, b->mn[1], b, o, d)) return 1;
    if (box_face(1, b->mx[1], b, o, d)) return 1;
    if (box_face(2, b->mn[2], b, o, d)) return 1;
    if (box_face(2, b->mx[2], b, o, d)) return 1;
    return 0;
}

/* Sphere::intersectRay (main.cpp:600-633) — far root is NOT divided by 2a */
static int sphere_intersect(const rt_sphere* sp, v3 o, v3 d, float* tMin)
{
    v3 tmp = vsub(o, vld(sp->centre));
    float a = vdot(d, d);
    float b = 2.0f * vdot(d, tmp);
    float c = vdot(tmp, tmp) - sp->radius * sp->radius;
    float disc = (b * b) - (4.0f * a * c);
    if (geq(disc, 0.0f)) {
        const float kEpsilon = 0.01f;
        float e = sqrtf(disc);
        float denom = 2.0f * a;
        float t = (-b - e) / denom;
        if (geq(t, kEpsilon)) { *tMin = t; return 1; }
        t = (-b + e);
        if (geq(t, kEpsilon)) { *tMin = t; return 1; }
    }
    return 0;
}

/* Sphere::hit (main.cpp:578-598) */
static int sphere_hit(const Scene* s, uint32_t i, v3 o, v3 d, ShadeRec* sr)
{
    const rt_sphere* sp = &s->d->spheres[i];
    v3 tmp = vsub(o, vld(sp->centre));
    float t = FLT_MAX;
    int inter = sphere_intersect(sp, o, d, &t);
    if (inter && t < sr->t) {
        sr->normal = vdivs(vadd(tmp, vscale(d, t)), sp->radius);
        sr->ro = o; sr->rd = d; sr->color = V(0, 0, 0); sr->t = t; sr->material = sp->material;
        sr->prim = s->prim_base_sphere + i;
        return 1;
    }
    return 0;
}

/* Triangle::intersectRay, 3-argument overload (main.cpp:713-739): single-sided, no t sign test */
static int tri_intersect(const rt_triangle* T, v3 o, v3 d, float* tMin, v3* bary)
{
    v3 v0 = vld(T->v0);
    v3 v0v1 = vsub(vld(T->v1), v0);
    v3 v0v2 = vsub(vld(T->v2), v0);
    v3 pvec = vcross(d, v0v2);
    float det = vdot(v0v1, pvec);
    if (det < 0.000001f) return 0;
    float invDet = 1.0f / det;
    v3 tvec = vsub(o, v0);
    float u = vdot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return 0;
    v3 qvec = vcross(tvec, v0v1);
    float v = vdot(d, qvec) * invDet;
    if (v < 0 || u + v > 1) return 0;
    float t = vdot(v0v2, qvec) * invDet;
    *tMin = t;
    *bary = V(u, v, 1.0f - u - v);
    return 1;
}

/* Triangle::hit (main.cpp:641-663) */
static int tri_hit(const Scene* s, uint32_t i, v3 o, v3 d, ShadeRec* sr)
{
    const rt_triangle* T = &s->d->triangles[i];
    float t = FLT_MAX;
    v3 bc;
    int inter = tri_intersect(T, o, d, &t, &bc);
    if (inter && t < sr->t) {
        v3 v2v0 = vsub(vld(T->v2), vld(T->v0));
        v3 v1v0 = vsub(vld(T->v1), vld(T->v0));
        sr->normal = vnormalize(vcross(v1v0, v2v0));
        sr->t = t; sr->ro = o; sr->rd = d; sr->color = V(0, 0, 0); sr->material = T->material;
        /* uv = uv1*b.x + uv2*b.y + uv0*b.z, summed left to right */
        sr->u = (T->uv1[0] * bc.x + T->uv2[0] * bc.y) + T->uv0[0] * bc.z;
        sr->v = (T->uv1[1] * bc.x + T->uv2[1] * bc.y) + T->uv0[1] * bc.z;
        sr->prim = s->prim_base_tri + i;
        return 1;
    }
    return 0;
}

/* Mesh::hit (main.cpp:890-912): box gate, then every triangle */
static int mesh_hit(const Scene* s, uint32_t mi, v3 o, v3 d, ShadeRec* sr)
{
    float t = FLT_MAX;
    int intersected = 0;
    if (!box_intersects(&s->mesh_box[mi], o, d)) return 0;
    const rt_mesh* m = &s->d->meshes[mi];
    for (uint32_t i = 0; i < m->num_triangles; ++i) {
        int inter = tri_hit(s, m->first_triangle + i, o, d, sr);
        if (inter && sr->t < t) { t = sr->t; intersected = 1; }
    }
    return intersected;
}

/* MultiMesh::hit (main.cpp:768-790) */
static int multimesh_hit(const Scene* s, uint32_t mmi, v3 o, v3 d, ShadeRec* sr)
{
    float t = FLT_MAX;
    int intersected = 0;
    if (!box_intersects(&s->mm_box[mmi], o, d)) return 0;
    const rt_multimesh* mm = &s->d->multimeshes[mmi];
    for (uint32_t i = 0; i < mm->num_meshes; ++i) {
        int inter = mesh_hit(s, mm->first_mesh + i, o, d, sr);
        if (inter && sr->t < t) { t = sr->t; intersected = 1; }
    }
    return intersected;
}

static int shape_hit(const Scene* s, rt_shape_ref r, v3 o, v3 d, ShadeRec* sr);

/* BVHAccel::hit (main.cpp:425-470): breadth-first, internal nodes box-tested, leaves not */
static int bvh_hit(const Scene* s, uint32_t bi, v3 o, v3 d, ShadeRec* sr)
{
    const Bvh* b = &s->bvh[bi];
    const BvhNode* root = &b->nodes[b->n - 1];
    if (root->leaf) return shape_hit(s, root->shape, o, d, sr);
    uint32_t* queue = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)b->n * 2 + 8);
    size_t head = 0, tail = 0;
    if (box_intersects(&root->box, o, d)) { queue[tail++] = root->left; queue[tail++] = root->right; }
    int intersects = 0;
    while (head < tail) {
        const BvhNode* node = &b->nodes[queue[head]];
        if (node->leaf) {
            if (shape_hit(s, node->shape, o, d, sr)) intersects = 1;
        } else if (box_intersects(&node->box, o, d)) {
            queue[tail++] = node->left; queue[tail++] = node->right;
        }
        ++head;
    }
    free(queue);
    return intersects;
}

static int shape_hit(const Scene* s, rt_shape_ref r, v3 o, v3 d, ShadeRec* sr)
{
    switch (r.kind) {
    case RT_SHAPE_PLANE: return plane_hit(s, r.index, o, d, sr);
    case RT_SHAPE_SPHERE: return sphere_hit(s, r.index, o, d, sr);
    case RT_SHAPE_TRIANGLE: return tri_hit(s, r.index, o, d, sr);
    case RT_SHAPE_MESH: return mesh_hit(s, r.index, o, d, sr);
    case RT_SHAPE_MULTIMESH: return multimesh_hit(s, r.index, o, d, sr);
    case RT_SHAPE_BVH: return bvh_hit(s, r.index, o, d, sr);
    }
    return 0;
}

/* `for (auto obj : world->scene) hit |= obj->hit(ray, rec)` — one closest-hit query */
typedef struct { uint64_t rays; Rng rng; } Ctx;
static int scene_hit(const Scene* s, Ctx* cx, v3 o, v3 d, ShadeRec* sr)
{
    int hit = 0;
    for (uint32_t i = 0; i < s->d->num_scene; ++i) hit |= shape_hit(s, s->d->scene[i], o, d, sr);
    cx->rays++;
    return hit;
}

/* ---- shape-level hierarchy: BVHAccel::addShape + generateBVH (main.cpp:323-423) ------------ */
static int shape_box(const Scene* s, rt_shape_ref r, Box* out)
{
    const rt_scene_desc* d = s->d;
    switch (r.kind) {
    case RT_SHAPE_SPHERE: { /* Sphere::getBoundingBoxPoints (main.cpp:571-576) */
        const rt_sphere* sp = &d->spheres[r.index];
        v3 c = vld(sp->centre), rr = V(sp->radius, sp->radius, sp->radius);
        box_start(out, vadd(c, rr)); box_add(out, vsub(c, rr));
        return 1;
    }
    case RT_SHAPE_TRIANGLE: { /* Triangle::getBoundingBoxPoints (main.cpp:705-711) */
        const rt_triangle* T = &d->triangles[r.index];
        box_start(out, vld(T->v0)); box_add(out, vld(T->v1)); box_add(out, vld(T->v2));
        return 1;
    }
    case RT_SHAPE_MESH: *out = s->mesh_box[r.index]; return 1;
    case RT_SHAPE_MULTIMESH: *out = s->mm_box[r.index]; return 1;
    }
    return 0;
}

static int bvh_build(const Scene* s, const rt_bvh* in, Bvh* out)
{
    uint32_t nl = in->num_leaves;
    if (nl == 0) return 0;
    out->nodes = (BvhNode*)calloc((size_t)nl * 2 + 64, sizeof(BvhNode));
    out->n = 0;
    for (uint32_t i = 0; i < nl; ++i) {
        BvhNode* n = &out->nodes[out->n++];
        n->leaf = 1; n->shape = s->d->bvh_leaves[in->first_leaf + i];
        if (!shape_box(s, n->shape, &n->box)) return 0;
    }
    BvhNode* H = out->nodes;
    size_t leaves = nl;
    size_t nextLayerLeaves = (size_t)ceilf(leaves / 2.0f);
    while (leaves != 1) {
        int layerEndRange = (int)out->n;
        int layerStartRange = (int)(out->n - leaves);
        for (size_t y = 0; y < nextLayerLeaves; y++) {
            size_t leaf1Index = 0, leaf2Index = 0;
            int i, k, set = 0;
            float minCentroidDist = FLT_MAX;
            for (i = layerEndRange - 1; i > layerStartRange; i--) {
                if (H[i].accessed) continue;
                for (k = i - 1; k >= layerStartRange; k--) {
                    if (k < 0 || k < layerStartRange) break;
                    if (H[k].accessed) continue;
                    /* glm::distance(p0,p1) = length(p1 - p0) */
                    float centroidDist = vlength(vsub(box_centroid(&H[k].box), box_centroid(&H[i].box)));
                    if (centroidDist < minCentroidDist) { minCentroidDist = centroidDist; leaf1Index = (size_t)i; leaf2Index = (size_t)k; set = 1; }
                }
            }
            if (!set) {
                for (i = layerEndRange - 1; i >= layerStartRange; i--) {
                    if (i < 0 || i < layerStartRange) break;
                    if (!H[i].accessed) { H[out->n++] = H[i]; H[i].accessed = 1; break; }
                }
            } else {
                BvhNode* n = &H[out->n++];
                memset(n, 0, sizeof *n);
                box_union(&n->box, &H[leaf1Index].box, &H[leaf2Index].box);
                n->leaf = 0; n->left = (uint32_t)leaf1Index; n->right = (uint32_t)leaf2Index;
                H[leaf1Index].accessed = 1; H[leaf2Index].accessed = 1;
            }
        }
        leaves = nextLayerLeaves;
        nextLayerLeaves = (size_t)ceilf(leaves / 2.0f);
    }
    return 1;
}

static void scene_free(Scene* s)
{
    if (s->bvh) for (uint32_t i = 0; i < s->d->num_bvhs; ++i) free(s->bvh[i].nodes);
    free(s->bvh); free(s->mesh_box); free(s->mm_box); free(s->light_dir);
}

static const char* g_err = "";
const char* oracle_last_error(void) { return g_err; }

static int scene_build(Scene* s, const rt_scene_desc* d, const rt_params* p)
{
    memset(s, 0, sizeof *s);
    s->d = d;
    s->prim_base_sphere = d->num_planes;
    s->prim_base_tri = d->num_planes + d->num_spheres;
    s->max_depth = p ? p->max_depth : 3;
    s->light_atten = p ? p->light_attenuation : 500000.0f;
    s->mesh_box = (Box*)calloc(d->num_meshes + 1, sizeof(Box));
    s->mm_box = (Box*)calloc(d->num_multimeshes + 1, sizeof(Box));
    s->bvh = (Bvh*)calloc(d->num_bvhs + 1, sizeof(Bvh));
    s->light_dir = (v3*)calloc(d->num_lights + 1, sizeof(v3));
    for (uint32_t i = 0; i < d->num_meshes; ++i) { /* Mesh ctor (main.cpp:875-881) */
        const rt_mesh* m = &d->meshes[i];
        if (m->num_triangles == 0) { g_err = "empty mesh"; return 0; }
        box_start(&s->mesh_box[i], vld(d->triangles[m->first_triangle].v0));
        for (uint32_t k = 0; k < m->num_triangles; ++k) {
            const rt_triangle* T = &d->triangles[m->first_triangle + k];
            box_add(&s->mesh_box[i], vld(T->v0)); box_add(&s->mesh_box[i], vld(T->v1)); box_add(&s->mesh_box[i], vld(T->v2));
        }
    }
    for (uint32_t i = 0; i < d->num_multimeshes; ++i) { /* MultiMesh ctor (main.cpp:757-765) */
        const rt_multimesh* mm = &d->multimeshes[i];
        if (mm->num_meshes == 0) { g_err = "empty multimesh"; return 0; }
        s->mm_box[i] = s->mesh_box[mm->first_mesh];
        for (uint32_t k = 0; k < mm->num_meshes; ++k) box_union(&s->mm_box[i], &s->mm_box[i], &s->mesh_box[mm->first_mesh + k]);
    }
    for (uint32_t i = 0; i < d->num_bvhs; ++i)
        if (!bvh_build(s, &d->bvhs[i], &s->bvh[i])) { g_err = "bad bvh"; return 0; }
    for (uint32_t i = 0; i < d->num_lights; ++i)
        s->light_dir[i] = d->lights[i].kind == RT_LIGHT_DIRECTIONAL ? vnormalize(vld(d->lights[i].v)) : vld(d->lights[i].v);
    return 1;
}

/* ---- lights ---------------------------------------------------------------------------------- */
static v3 ray_at(const ShadeRec* sr, float t) { return vadd(sr->ro, vscale(sr->rd, t)); } /* Ray::operator() */

/* PointLight / Directional getDirection, getSourcePoint (main.cpp:1320-1345) */
static v3 light_direction(const Scene* s, uint32_t li, const ShadeRec* sr)
{
    if (s->d->lights[li].kind == RT_LIGHT_POINT) return vsub(s->light_dir[li], ray_at(sr, sr->t));
    return s->light_dir[li];
}
static v3 light_source(const Scene* s, uint32_t li, const ShadeRec* sr)
{
    if (s->d->lights[li].kind == RT_LIGHT_POINT) return s->light_dir[li];
    return vadd(ray_at(sr, sr->t), vscale(s->light_dir[li], 1000000.0f));
}
/* Light::L (main.cpp:551-554) */
static v3 light_L(const Scene* s, uint32_t li) { return vscale(vld(s->d->lights[li].colour), s->d->lights[li].radiance); }

/* sampleHemisphere (main.cpp:4-19) */
static v3 sample_hemisphere(float inX, float inY)
{
    const float r = sqrtf(1.0f - inX * inX);
    const float phi = 2 * 3.14159265358979323846264338327950288f * inY;
    return V(cosf(phi) * r, inX, sinf(phi) * r);
}

/* rotationMatToSpace (main.cpp:21-34) followed by mat4*vec4{r,0} (type_mat4x4.inl:512-523:
 * (col0*x + col1*y) + (col2*z + col3*0)) */
static v3 rotate_to_space(v3 up, v3 r)
{
    v3 w = vnormalize(up);
    v3 u = vnormalize(vcross(w, V(0.053f, 1.0f, 0.0332f)));
    v3 v = vcross(w, u);
    v3 a = vadd(vscale(u, r.x), vscale(w, r.y));
    v3 b = vadd(vscale(v, r.z), V(0.0f * 0.0f, 0.0f * 0.0f, 0.0f * 0.0f));
    return vadd(a, b);
}

/* Ambient::L (main.cpp:1361-1404); occlusion_samples == 0 -> Light::L */
static v3 ambient_L(const Scene* s, Ctx* cx, const ShadeRec* sr)
{
    const rt_ambient* A = &s->d->ambient;
    v3 base = vscale(vld(A->colour), A->radiance);
    if (A->occlusion_samples <= 0) return base;
    unsigned n = (unsigned)A->occlusion_samples;
    float dist = A->occlusion_dist;
    float gridBlockSize = 1.0f / n;
    float occlusionMaxSumDist = (dist * n * n);
    float occludedDistance = occlusionMaxSumDist;
    for (size_t x = 0; x < n; x++)
        for (size_t y = 0; y < n; y++) {
            float sampleX = (float)x * gridBlockSize + (float)(((double)rng_rand(&cx->rng) / RAND_MAX) * gridBlockSize);
            float sampleY = (float)y * gridBlockSize + (float)(((double)rng_rand(&cx->rng) / RAND_MAX) * gridBlockSize);
            v3 r = sample_hemisphere(sampleX, sampleY);
            r = rotate_to_space(sr->normal, r);
            r = vnormalize(r);
            v3 o = vadd(ray_at(sr, sr->t), vscale(sr->normal, 0.01f));
            ShadeRec rec; rec_init(&rec);
            rec.depth = sr->depth + 1;
            rec.t = dist;
            if (scene_hit(s, cx, o, r, &rec)) occludedDistance -= (dist - rec.t);
        }
    float occlusionFactor = (occludedDistance / occlusionMaxSumDist);
    return vscale(vscale(base, occlusionFactor), occlusionFactor);
}

/* Material::shadowed (main.cpp:40-76) */
static float shadowed(const Scene* s, Ctx* cx, const ShadeRec* sr, uint32_t li)
{
    v3 o = light_source(s, li, sr);
    v3 ip = ray_at(sr, sr->t);
    v3 d = vnormalize(vsub(ip, o));
    float distToLight = vlength(vsub(ip, o)) - 0.01f;
    ShadeRec rec; rec_init(&rec);
    rec.shadowRay = 1;
    int shadow = 0, transparentOnly = 1;
    for (uint32_t i = 0; i < s->d->num_scene; ++i) {
        rec.t = distToLight;
        int hit = shape_hit(s, s->d->scene[i], o, d, &rec);
        if (hit && rec.t < distToLight) {
            transparentOnly &= (s->d->materials[rec.material].kind == RT_MAT_REFRACTION);
            shadow = 1;
        }
    }
    cx->rays++;
    if (shadow) return transparentOnly ? 0.8f : 0.0f;
    return 1.0f;
}

/* SpecularReflection::fn (main.cpp:1090-1105): ks * pow() evaluated in double */
static float specular_fn(float ks, int exp, v3 normal, v3 wo, v3 wi)
{
    v3 r = vreflect(wi, normal);
    float vDotr = vdot(r, vneg(wo));
    float L = 0.0f;
    if (vDotr > 0.0) L = (float)(ks * pow((double)vDotr, (double)exp));
    return L;
}

/* ImageTexture::getColour (main.cpp:173-188).  The reference indexes (x + mHeight*y)*4 with no
 * bounds check; reads that fall outside the image are clamped to it here. */
static void texture_colour(const Scene* s, int tex, float u, float v, float out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    if (tex < 0 || (uint32_t)tex >= s->d->num_textures || !s->d->textures[tex].rgba) return;
    const rt_texture* T = &s->d->textures[tex];
    int x = (int)floorf(u * (T->width));
    int y = (int)floorf(v * (T->height));
    long long texel = (long long)x + (long long)T->height * y;
    long long last = (long long)T->width * T->height - 1;
    if (texel < 0) texel = 0;
    if (texel > last) texel = last;
    const uint8_t* px = T->rgba + texel * 4;
    out[0] = px[0]; out[1] = px[1]; out[2] = px[2]; out[3] = px[3];
}

static v3 shade(const Scene* s, Ctx* cx, ShadeRec* sr);

/* Reflective::rho (main.cpp:1437-1468) */
static v3 reflective_rho(const Scene* s, Ctx* cx, const ShadeRec* sr, v3 reflected)
{
    if (sr->depth > (unsigned)s->max_depth) return sr->color;
    if (sr->shadowRay) return V(0, 0, 0);
    v3 r = vreflect(vneg(reflected), sr->normal);
    v3 o = vadd(vadd(sr->ro, vscale(sr->rd, sr->t)), vscale(r, 0.1f));
    ShadeRec rec; rec_init(&rec);
    rec.depth = sr->depth + 1;
    rec.t = FLT_MAX;
    if (scene_hit(s, cx, o, r, &rec)) rec.color = shade(s, cx, &rec);
    return rec.color;
}

/* Transparency::refraction (main.cpp:1566-1592) */
static v3 refraction_dir(const ShadeRec* sr, float* enterIndexOfRefract)
{
    float cosi = clampf(vdot(sr->rd, sr->normal), -1.0f, 1.0f);
    v3 n = sr->normal;
    if (cosi < 0) cosi = -cosi; else n = vneg(sr->normal);
    float eta = sr->ior / *enterIndexOfRefract;
    float k = 1 - eta * eta * (1 - cosi * cosi);
    if (k < 0) { *enterIndexOfRefract = sr->ior; return vreflect(sr->rd, n); }
    return vnormalize(vadd(vscale(sr->rd, eta), vscale(n, (eta * cosi - sqrtf(k)))));
}

/* Transparency::rho (main.cpp:1525-1559) */
static v3 transparency_rho(const Scene* s, Ctx* cx, const ShadeRec* sr, float material_ior)
{
    if (sr->depth > (unsigned)s->max_depth) return sr->color;
    if (sr->shadowRay) return V(0, 0, 0);
    ShadeRec rec; rec_init(&rec);
    float enterIndexOfRefract = material_ior;
    v3 r = refraction_dir(sr, &enterIndexOfRefract);
    rec.ior = enterIndexOfRefract;
    v3 o = vadd(vadd(sr->ro, vscale(sr->rd, sr->t)), vscale(r, 0.1f));
    rec.depth = sr->depth + 1;
    rec.t = FLT_MAX;
    if (scene_hit(s, cx, o, r, &rec)) rec.color = shade(s, cx, &rec);
    return rec.color;
}

static v3 clamp3(v3 L) { return V(clamp01(L.x), clamp01(L.y), clamp01(L.z)); }

/* Material::shade for the five classes */
static v3 shade(const Scene* s, Ctx* cx, ShadeRec* sr)
{
    const rt_material* M = &s->d->materials[sr->material];
    uint32_t numLights = s->d->num_lights;
    switch (M->kind) {
    case RT_MAT_MATTE: { /* main.cpp:1143-1165: wi not normalised, no shadows, no clamp */
        v3 L = vmul(vscale(vld(M->colour), M->ka), ambient_L(s, cx, sr));
        for (uint32_t i = 0; i < numLights; ++i) {
            v3 wi = light_direction(s, i, sr);
            float nDotWi = vdot(sr->normal, wi);
            if (nDotWi > 0.0f) {
                v3 fn = vscale(vscale(vld(M->colour), M->kd), 0.318309886183790671537767526745028724f);
                L = vadd(L, vscale(vmul(fn, light_L(s, i)), nDotWi));
            }
        }
        return L;
    }
    case RT_MAT_SPECULAR: { /* main.cpp:1215-1244 */
        v3 wo = vnormalize(vneg(sr->rd));
        v3 L = vmul(vscale(vld(M->colour), M->ka), ambient_L(s, cx, sr));
        for (uint32_t i = 0; i < numLights; ++i) {
            float shadowAttenuation = shadowed(s, cx, sr, i);
            v3 wi = vnormalize(light_direction(s, i, sr));
            float nDotWi = vdot(sr->normal, wi);
            if (nDotWi > 0.0f) {
                v3 fn = vscale(vscale(vld(M->colour), M->kd), 0.318309886183790671537767526745028724f);
                L = vadd(L, vscale(vscale(vmul(fn, light_L(s, i)), nDotWi), shadowAttenuation));
                float sp = specular_fn(M->ks, M->spec_exp, sr->normal, wo, wi);
                L = vadd(L, vscale(V(sp, sp, sp), shadowAttenuation));
            }
        }
        return clamp3(L);
    }
    case RT_MAT_TEXTURED: { /* main.cpp:1269-1299 */
        float tc[4];
        texture_colour(s, M->texture, sr->u, sr->v, tc);
        v3 texColour = V(tc[0] / 255.0f, tc[1] / 255.0f, tc[2] / 255.0f);
        v3 wo = vnormalize(vneg(sr->rd));
        v3 L = vmul(vmul(vscale(V(1, 1, 1), M->ka), ambient_L(s, cx, sr)), texColour);
        for (uint32_t i = 0; i < numLights; ++i) {
            float shadowAttenuation = shadowed(s, cx, sr, i);
            v3 wi = vnormalize(light_direction(s, i, sr));
            float nDotWi = vdot(sr->normal, wi);
            if (nDotWi > 0.0f) {
                float lightToSurfaceDist = vlength(vsub(light_source(s, i, sr), ray_at(sr, sr->t)));
                float attenuation = s->light_atten / (lightToSurfaceDist * lightToSurfaceDist);
                L = vadd(L, vscale(vscale(vscale(vmul(texColour, light_L(s, i)), nDotWi), attenuation), shadowAttenuation));
                float sp = specular_fn(M->ks, M->spec_exp, sr->normal, wo, wi);
                L = vadd(L, vscale(vscale(V(sp, sp, sp), attenuation), shadowAttenuation));
            }
        }
        return clamp3(L);
    }
    case RT_MAT_MIRROR: { /* main.cpp:1487-1511 */
        v3 wo = vnormalize(vneg(sr->rd));
        v3 L = vmul(vscale(vld(M->colour), M->ka), ambient_L(s, cx, sr));
        L = vadd(L, reflective_rho(s, cx, sr, wo));
        for (uint32_t i = 0; i < numLights; ++i) {
            v3 wi = vnormalize(light_direction(s, i, sr));
            float nDotWi = vdot(sr->normal, wi);
            if (nDotWi > 0.0f) {
                float sp = specular_fn(M->ks, M->spec_exp, sr->normal, wo, wi);
                L = vadd(L, V(sp, sp, sp));
            }
        }
        return clamp3(L);
    }
    case RT_MAT_REFRACTION: /* main.cpp:1607-1616 */
        return clamp3(transparency_rho(s, cx, sr, M->ior));
    }
    return V(0, 0, 0);
}

/* ---- camera (main.cpp:1407-1424) ------------------------------------------------------------------ */
typedef struct { v3 eye, U, Vv, W; float dist; } Cam;
static void cam_init(Cam* c, const rt_camera* in)
{
    c->eye = vld(in->eye); c->dist = in->frustum_dist;
    c->W = vnormalize(vsub(vld(in->eye), vld(in->look_at)));
    c->Vv = vnormalize(vcross(c->W, vld(in->up)));
    c->U = vnormalize(vcross(c->W, c->Vv));
}
static void cam_ray(const Cam* c, float x, float y, v3* o, v3* d)
{
    *o = c->eye;
    *d = vnormalize(vsub(vadd(vscale(c->Vv, x), vscale(c->U, y)), vscale(c->W, c->dist)));
}

/* ---- samplers (main.cpp:503-548, 971-1035) ---------------------------------------------------------- */
typedef struct { float* sx; float* sy; int* shuffled; int numSamples, numSets, jump; unsigned long count; uint64_t st; } Sampler;

static uint32_t sampler_u32(Sampler* s) { s->st = s->st * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(s->st >> 32); }

static void sampler_init(Sampler* s, const rt_params* p, Rng* g, uint64_t stream)
{
    int ns = p->num_samples, sets = p->num_sets > 0 ? p->num_sets : 1;
    s->numSamples = ns; s->numSets = sets; s->count = 0; s->jump = 0;
    s->st = 0x853c49e6748fea9bULL ^ ((uint64_t)p->seed << 17) ^ (stream * 0x9E3779B97F4A7C15ULL);
    s->sx = (float*)calloc((size_t)ns * sets + 1, sizeof(float));
    s->sy = (float*)calloc((size_t)ns * sets + 1, sizeof(float));
    s->shuffled = (int*)calloc((size_t)ns * sets + 1, sizeof(int));
    /* setupShuffledIndices (main.cpp:515-537); the reference shuffles with a random_device-seeded
     * mt19937, i.e. irreproducibly — any uniform shuffle restates it */
    int* idx = (int*)malloc(sizeof(int) * (size_t)ns);
    for (int j = 0; j < ns; ++j) idx[j] = j;
    for (int q = 0; q < sets; ++q) {
        for (int j = ns - 1; j > 0; --j) { int r = (int)(sampler_u32(s) % (uint32_t)(j + 1)); int t = idx[j]; idx[j] = idx[r]; idx[r] = t; }
        for (int j = 0; j < ns; ++j) s->shuffled[q * ns + j] = idx[j];
    }
    free(idx);
    int n = (int)sqrtf((float)ns), k = 0;
    if (p->sampler == RT_SAMPLER_REGULAR) {
        for (int j = 0; j < sets; ++j) for (int pp = 0; pp < n; ++pp) for (int q = 0; q < n; ++q) { s->sx[k] = (q + 0.5f) / n; s->sy[k] = (pp + 0.5f) / n; ++k; }
    } else if (p->sampler == RT_SAMPLER_JITTER) {
        float gridSize = 1.0f / n;
        for (int j = 0; j < sets; ++j) for (int pp = 0; pp < n; ++pp) for (int q = 0; q < n; ++q) {
            s->sx[k] = q * gridSize + ((float)rng_rand(g) / RAND_MAX * gridSize);
            s->sy[k] = pp * gridSize + ((float)rng_rand(g) / RAND_MAX * gridSize);
            ++k;
        }
    } else {
        for (int j = 0; j < sets * ns; ++j) { s->sx[k] = (float)(sampler_u32(s) >> 8) * (1.0f / 16777216.0f); s->sy[k] = (float)(sampler_u32(s) >> 8) * (1.0f / 16777216.0f); ++k; }
    }
}
static void sampler_free(Sampler* s) { free(s->sx); free(s->sy); free(s->shuffled); }

/* Sampler::sampleUnitSquare (main.cpp:539-548) */
static void sample_unit_square(Sampler* s, float* x, float* y)
{
    if (s->count % (unsigned long)s->numSamples == 0) s->jump = (int)(sampler_u32(s) % (uint32_t)s->numSets) * s->numSamples;
    int i = s->jump + s->shuffled[s->jump + (int)(s->count++ % (unsigned long)s->numSamples)];
    *x = s->sx[i]; *y = s->sy[i];
}

/* sample point j from the counter hash — the GPU's ray-generation rule (same formulas as
 * Regular/Jitter::generateSamples, main.cpp:986-987, 1010-1011) */
static void hashed_sample(const rt_params* p, Rng* g, int j, float* x, float* y)
{
    int n = (int)sqrtf((float)p->num_samples);
    if (n < 1) n = 1;
    int jj = j % (n * n), pp = jj / n, q = jj % n;
    if (p->sampler == RT_SAMPLER_REGULAR) { *x = (q + 0.5f) / n; *y = (pp + 0.5f) / n; return; }
    if (p->sampler == RT_SAMPLER_RANDOM) {
        *x = (float)(rng_rand(g) >> 7) * (1.0f / 16777216.0f);
        *y = (float)(rng_rand(g) >> 7) * (1.0f / 16777216.0f);
        return;
    }
    float gridSize = 1.0f / n;
    *x = q * gridSize + ((float)rng_rand(g) / RAND_MAX * gridSize);
    *y = pp * gridSize + ((float)rng_rand(g) / RAND_MAX * gridSize);
}

/* ---- driver: raytrace() (main.cpp:1620-1665) with row-major r*width+c output ------------------------- */
typedef struct {
    int32_t rng_mode, nthreads, shade;
    const char* tmpdir;
} oracle_options; /* same layout as ref_options */

typedef struct {
    float* rgb; uint32_t* primid; float* t; float* normal; float* uv;
    uint64_t rays; double seconds;
} oracle_outputs; /* same layout as ref_outputs */

typedef struct {
    const Scene* s; const Cam* cam; const rt_params* p; const oracle_options* opt; oracle_outputs* out;
    int tid, nthreads; uint64_t rays;
} Job;

static void* worker(void* arg)
{
    Job* J = (Job*)arg;
    const rt_params* p = J->p;
    const int W = p->width, spp = p->num_samples;
    Ctx cx; memset(&cx, 0, sizeof cx);
    cx.rng.mode = J->opt->rng_mode; cx.rng.seed = p->seed;
    Sampler smp; memset(&smp, 0, sizeof smp);
    if (J->opt->rng_mode != 1) sampler_init(&smp, p, &cx.rng, (uint64_t)J->tid + 1);
    const float avg = 1.0f / spp;
    for (int r = p->y0 + J->tid; r < p->y1; r += J->nthreads)
        for (int c = p->x0; c < p->x1; ++c) {
            v3 pixelAverage = V(0, 0, 0);
            for (int j = 0; j < spp; ++j) {
                float sx, sy;
                cx.rng.pixel = (uint32_t)(r * W + c); cx.rng.sample = (uint32_t)j; cx.rng.counter = 0;
                if (J->opt->rng_mode == 1) hashed_sample(p, &cx.rng, j, &sx, &sy); else sample_unit_square(&smp, &sx, &sy);
                cx.rng.counter = 2;
                v3 o, d;
                cam_ray(J->cam, (c - (0.5f * p->width)) + sx, (r - (0.5f * p->height)) + sy, &o, &d);
                ShadeRec sr; rec_init(&sr);
                sr.t = FLT_MAX;
                if (scene_hit(J->s, &cx, o, d, &sr)) pixelAverage = vadd(pixelAverage, shade(J->s, &cx, &sr));
            }
            size_t i = ((size_t)r * W + c) * 3;
            J->out->rgb[i] = pixelAverage.x * avg; J->out->rgb[i + 1] = pixelAverage.y * avg; J->out->rgb[i + 2] = pixelAverage.z * avg;
        }
    if (J->opt->rng_mode != 1) sampler_free(&smp);
    J->rays = cx.rays;
    return NULL;
}

int oracle_render(const rt_scene_desc* d, const rt_camera* cam_in, const rt_params* p, const oracle_options* opt, oracle_outputs* out)
{
    Scene s;
    if (!scene_build(&s, d, p)) { scene_free(&s); return -1; }
    Cam cam; cam_init(&cam, cam_in);
    const int W = p->width;
    if (out->primid || out->t || out->normal || out->uv) {
        Ctx cx; memset(&cx, 0, sizeof cx);
        cx.rng.mode = opt->rng_mode == 0 ? 2 : opt->rng_mode; cx.rng.seed = p->seed;
        for (int r = p->y0; r < p->y1; ++r)
            for (int c = p->x0; c < p->x1; ++c) {
                float sx = 0.5f, sy = 0.5f;
                cx.rng.pixel = (uint32_t)(r * W + c); cx.rng.sample = 0; cx.rng.counter = 0;
                if (opt->rng_mode == 1) hashed_sample(p, &cx.rng, 0, &sx, &sy);
                v3 o, dd;
                cam_ray(&cam, (c - (0.5f * p->width)) + sx, (r - (0.5f * p->height)) + sy, &o, &dd);
                ShadeRec sr; rec_init(&sr);
                sr.t = FLT_MAX;
                int hit = scene_hit(&s, &cx, o, dd, &sr);
                size_t i = (size_t)r * W + c;
                if (out->primid) out->primid[i] = hit ? sr.prim : 0xFFFFFFFFu;
                if (out->t) out->t[i] = sr.t;
                if (out->normal) { out->normal[i * 3] = hit ? sr.normal.x : 0; out->normal[i * 3 + 1] = hit ? sr.normal.y : 0; out->normal[i * 3 + 2] = hit ? sr.normal.z : 0; }
                if (out->uv) { out->uv[i * 2] = hit ? sr.u : 0; out->uv[i * 2 + 1] = hit ? sr.v : 0; }
            }
    }
    if (opt->shade && out->rgb) {
        int nt = opt->nthreads > 0 ? opt->nthreads : 1;
        if (nt > 256) nt = 256;
        Job jobs[256]; pthread_t th[256];
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int i = 0; i < nt; ++i) {
            jobs[i].s = &s; jobs[i].cam = &cam; jobs[i].p = p; jobs[i].opt = opt; jobs[i].out = out;
            jobs[i].tid = i; jobs[i].nthreads = nt; jobs[i].rays = 0;
        }
        if (nt == 1) worker(&jobs[0]);
        else {
            for (int i = 0; i < nt; ++i) pthread_create(&th[i], NULL, worker, &jobs[i]);
            for (int i = 0; i < nt; ++i) pthread_join(th[i], NULL);
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        out->seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        out->rays = 0;
        for (int i = 0; i < nt; ++i) out->rays += jobs[i].rays;
    }
    scene_free(&s);
    return 0;
}

/* ---- single-function probes (mirrors of the ref_* probes in ref_harness.cpp) ------------------------------ */
int oracle_geq(float a, float b) { return geq(a, b); }
void oracle_sample_hemisphere(float a, float b, float* o) { v3 v = sample_hemisphere(a, b); o[0] = v.x; o[1] = v.y; o[2] = v.z; }
void oracle_rotate_to_space(const float* up, const float* v, float* o) { v3 r = rotate_to_space(vld(up), vld(v)); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
void oracle_camera_ray(const rt_camera* cam, float x, float y, float* o3, float* d3)
{
    Cam c; cam_init(&c, cam);
    v3 o, d; cam_ray(&c, x, y, &o, &d);
    o3[0] = o.x; o3[1] = o.y; o3[2] = o.z; d3[0] = d.x; d3[1] = d.y; d3[2] = d.z;
}
int oracle_box_intersects(const float* pts, int n, const float* o3, const float* d3)
{
    Box b; box_start(&b, vld(pts));
    for (int i = 1; i < n; ++i) box_add(&b, vld(pts + 3 * i));
    return box_intersects(&b, vld(o3), vld(d3));
}
float oracle_specular_fn(float ks, int exp, const float* n3, const float* wo3, const float* wi3) { return specular_fn(ks, exp, vld(n3), vld(wo3), vld(wi3)); }
void oracle_refraction(float material_ior, float sr_ior, const float* d3, const float* n3, float* out3, float* enter_ior)
{
    ShadeRec sr; rec_init(&sr);
    sr.rd = vld(d3); sr.normal = vld(n3); sr.ior = sr_ior;
    float e = material_ior;
    v3 r = refraction_dir(&sr, &e);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z; *enter_ior = e;
}
void oracle_quantize(const float* rgb, size_t n, uint8_t* out) { for (size_t i = 0; i < n; ++i) out[i] = (unsigned char)(rgb[i] * 255); }
int oracle_bvh_dump(const rt_scene_desc* d, uint32_t which, float* out, int cap_nodes)
{
    Scene s;
    if (!scene_build(&s, d, NULL) || which >= d->num_bvhs) { scene_free(&s); return -1; }
    const Bvh* b = &s.bvh[which];
    int n = (int)b->n;
    for (int i = 0; i < n && i < cap_nodes; ++i) {
        float* o = out + (size_t)i * 9;
        o[0] = b->nodes[i].leaf ? 1.0f : 0.0f;
        o[1] = b->nodes[i].leaf ? -1.0f : (float)b->nodes[i].left;
        o[2] = b->nodes[i].leaf ? -1.0f : (float)b->nodes[i].right;
        for (int k = 0; k < 3; ++k) { o[3 + k] = b->nodes[i].box.mn[k]; o[6 + k] = b->nodes[i].box.mx[k]; }
    }
    scene_free(&s);
    return n;
}
