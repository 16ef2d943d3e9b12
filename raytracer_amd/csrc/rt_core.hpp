// rt_core.hpp — per-ray / per-hit device functions of the MI355X render path.
//
// Everything here is written once and compiled twice: by hipcc into the gfx950 kernels
// (rt_kernels.hip) and by g++ into a test-only host emulation (tests/emul/) that lets the CPU
// test-suite exercise the exact device arithmetic.  The product never runs the host build.
//
// Arithmetic contract (what makes the GPU agree with the reference bit for bit in the
// deterministic modes): fp32, no FMA contraction (-ffp-contract=off; x86-64 g++ -O2 emits none
// for the reference), IEEE divide and sqrt, glm's operation order (dot = (x+y)+z, normalize =
// v*(1/sqrt(dot)), reflect = I - N*dot(N,I)*2; reference include/glm/detail/func_geometric.inl).
// Explicit fmaf() appears only in the conservative BVH culling test, which never decides a
// result on its own (every accepted hit comes from the exact tests below).
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "rt_amd.h"
#include "rt_rng.h"

#if defined(__HIPCC__)
#define RT_FN __host__ __device__ inline
#else
#define RT_FN inline
#endif

namespace rt {

struct alignas(16) q4 { float x, y, z, w; };
struct f3 { float x, y, z; };

RT_FN uint32_t f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
RT_FN float u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

RT_FN f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_FN f3 xyz(const q4& q) { return F3(q.x, q.y, q.z); }
RT_FN f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_FN f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_FN f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_FN f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
RT_FN f3 operator/(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
RT_FN f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
RT_FN float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_FN f3 cross(f3 x, f3 y) { return F3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
RT_FN f3 normalize(f3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
RT_FN float length(f3 v) { return sqrtf(dot(v, v)); }
RT_FN f3 reflect(f3 I, f3 N) { return I - N * dot(N, I) * 2.0f; }
RT_FN float clampf(float x, float lo, float hi) { float m = (x < lo) ? lo : x; return (hi < m) ? hi : m; }
RT_FN f3 clamp01(f3 L) { return F3(clampf(L.x, 0.0f, 1.0f), clampf(L.y, 0.0f, 1.0f), clampf(L.z, 0.0f, 1.0f)); }

// ---- flattened scene (built by rt_scene.cpp, resident in HBM) -------------------------------
enum : uint32_t { G_GROUP = 0, G_GATED = 1, G_PLANE = 2, G_SPHERE = 3, G_TRI = 4, G_MESH = 5, G_KIND_MASK = 0xFFu,
                  G_SAME_BOX = 0x100u };   // flag: box bit-identical to the enclosing gate's, whose test already passed

// Shape-level structure of the reference, in depth-first preorder with skip pointers (no stack):
// world->scene entries, BVHAccel internal nodes, MultiMesh and Mesh gates.  A gated node is
// entered only if the reference's own box test passes (BoundingVolumeBox::intersects).
// PLANE and SPHERE nodes carry their primitive inline (no second fetch during the walk):
// PLANE: mn = point, mx = normal; SPHERE: mn = centre, mx = {radius^2, radius, 0}.
struct alignas(16) GateNode {
    float mn[3]; uint32_t kind;
    float mx[3]; uint32_t skip;  // G_GATED: next node when this subtree is not entered; G_MESH: the Mesh's root again (`a`); other kinds: the next node
    uint32_t a;                  // PLANE/SPHERE: index; TRI: triangle slot; MESH: root of its triangle BVH
    uint32_t rank;               // reference visiting order of the (first) primitive below
    uint32_t prim;               // MESH: global triangle index of its triangle 0
    uint32_t item;               // world->scene entry this node belongs to
};

// hit.prim encoding
enum : uint32_t { H_PLANE = 0u << 30, H_SPHERE = 1u << 30, H_TRI = 2u << 30, H_MISS = 0xFFFFFFFFu, H_MASK = 0x3FFFFFFFu };

struct DMaterial { int32_t kind; float colour[3]; float kd, ka, ks; int32_t spec_exp; float ior; int32_t texture; float pad0, pad1; };
struct DLight { int32_t kind; float v[3]; float colour[3]; float radiance; };
struct DTexture { uint32_t offset; int32_t width, height; int32_t set; };

// Compact copy of a gate node for the LDS-resident walk (k_ao_walk_lds, round 3): 16 bytes instead of 48, so that the
// whole shape-level structure of a scene with thousands of shapes (config 5: 7 500 nodes = 120 KB) sits in a CU's 160 KB
// of LDS and a gate step costs no global gather unless the node may be entered.
//   x, y, z : the box as 16-bit grid indices, word a = mn[a] | mx[a] << 16, coordinate = fmaf(q, gq_scale[a], gq_base[a]),
//             rounded OUTWARDS (verified at build with this very formula): a conservative miss of the decoded box is a
//             conservative miss of the real one, so it proves the reference's test false exactly as gate_maybe does.
//             G_MESH | G_SAME_BOX (a MultiMesh's only Mesh: no test of its own): x = the Mesh's triangle-BVH root.
//   w       : skip (bits 0..23) | kind (bits 24..26) | G_SAME_BOX (bit 27)
// A box that may be entered is decided on the exact 32-byte box from global memory (gate_surely / box_intersects).
struct alignas(16) QGate { uint32_t x, y, z, w; };
#define RT_QGATE_MAX 7680u   // nodes the LDS copy holds (7680 * 16 B = 120 KB); scenes with more use the global walk
RT_FN float gate_dq(uint32_t q, float scale, float base) { return fmaf((float)q, scale, base); }

#define RT_LEAD_PLANES 8
struct DScene {
    const GateNode* gates; uint32_t num_gates;
    const GateNode* gates_fast; uint32_t num_gates_fast;   // GPU: the first nodes staged in LDS by the kernel (0 = none)
    const QGate* qgates; uint32_t num_qgates;              // compact copy of `gates` for the LDS-resident walk (0: none, too many nodes)
    f3 gq_base, gq_scale;                                  // its grid
    const q4* planes;    // 2 / plane : {point, bits(material)} {normal, 0}
    const q4* spheres;   // 2 / sphere: {centre, radius} {radius^2, bits(material), 0, 0}
    const q4* tris;      // 3 / slot  : {v0, bits(global tri)} {e1=v1-v0, bits(material)} {e2=v2-v0, bits(rank = reference visiting order)}
    const q4* tri_attr;  // 2 / global triangle: {uv0, uv1} {uv2, 0, 0}
    const q4* nodes;     // 4 / BVH4 node (64 B, one aligned fetch): four child boxes as 24 x 16-bit grid indices + 4 child refs
    f3 qbase, qscale;    // node grid: coordinate = qbase + q * qscale (one frame per scene)
    const DMaterial* materials;
    const DLight* lights; uint32_t num_lights;
    const DTexture* textures; const uint8_t* texels; uint32_t num_textures;
    const uint8_t* item_opaque; uint32_t num_items;  // 1: no transparent material below this scene entry
    uint32_t big_scene;      // 1: triangles + nodes exceed the 256-MiB Infinity Cache (k_bvh then runs 5 waves per SIMD, not 6)
    uint32_t shadow_defer;   // 1: every Mesh lies in an all-opaque entry: shadow rays queue their meshes for k_bvh<any hit>
    rt_ambient ambient;
    uint32_t prim_base_sphere, prim_base_tri;
    float cull_pad;      // conservative padding of the BVH culling test, relative (see bvh_traverse)
    float extent;        // largest |coordinate| of the scene
    uint32_t debug;      // experiment switches (0 in production)
    // The gate array's leading run of top-level Plane nodes (the room of the reference's scenes), by value:
    // kernel arguments sit in constant memory, so a closest-hit walk reads them with scalar loads in a
    // wave-uniform loop instead of interpreting gate nodes.  {point, normal, bits(rank), bits(plane index)}
    uint32_t num_lead_planes;
    uint32_t lead_axis;   // 4 bits per lead plane: 0..2 = the normal is +-(that unit axis) exactly, 3 = any other normal
    float lead_planes[RT_LEAD_PLANES][8];
};

// BVH child reference: bit31 = leaf, bits 28..30 = triangle count-1, bits 0..27 = first slot
enum : uint32_t { REF_LEAF = 0x80000000u, REF_NONE = 0x7FFFFFFFu };

struct Counters { uint32_t nodes, tris, gates, rays; };

struct Stack { uint32_t* p; uint32_t stride; };  // entry i lives at p[i*stride] (LDS: stride = threads per block)
// Traversal stack: a 4-wide node pushes up to 3 children per level, the builders bound the depth at 13 four-wide
// levels (25 two-wide ones), and up to RT_DEFER_ROOTS - 1 further mesh roots wait below: 3 * 13 + 7 = 46.
#define RT_STACK_DEPTH 48
// Entries an in-place traversal (bvh_drain) can hold at most over a two-wide tree with d2 inner levels: the roots that wait
// (RT_PENDING_MAX) + 3 per four-wide level (a node is popped, up to four children are pushed), + 1 to spare.  The in-place
// kernels size their LDS stacks by it (k_shadow: 24 / 32 / 48 entries per lane; 48 covers the builders' depth bound).
#define RT_STACK_NEED(d2) (RT_PENDING_MAX + 3u * (((d2) + 1u) / 2u) + 1u)
#define RT_DEFER_ROOTS 8   // mesh roots one queued BVH item can carry
#define RT_PENDING_MAX 2   // deferred mesh roots before the inline drain loop is forced

struct Hit { float t; uint32_t prim; float u, v; uint32_t rank; };

// ---- exact primitive tests (each cites the reference code it reproduces) -----------------------

// atlas::core::geq  (include/atlas/core/Float.hpp:17-21,49-52)
RT_FN bool geq(float a, float b)
{
    if (a > b) return true;
    float scale = (fabsf(a) + fabsf(b)) / 2.0f;
    return fabsf(a - b) <= FLT_EPSILON * scale;
}

// Plane::intersectRay (main.cpp:943-957) with tMin = FLT_MAX
RT_FN bool plane_t(f3 point, f3 normal, f3 o, f3 d, float& t)
{
    float denom = dot(normal, d);
    if (fabsf(denom) < 0.00001f) return false;
    float tt = dot(point - o, normal) / denom;
    if (tt >= 0 && tt < FLT_MAX) { t = tt; return true; }
    return false;
}

// plane_t for a normal that is exactly +-(a unit axis): the reference's dot products reduce exactly (finite rays:
// the other terms are +-0) to denom = +-d[a] and dot(p - o, n) = +-(p[a] - o[a]) with the same sign, and IEEE
// division is symmetric in sign, so t = (p[a] - o[a]) / d[a] has the bits of the general evaluation.
RT_FN bool plane_axis_t(float pa, float oa, float da, float& t)
{
    if (fabsf(da) < 0.00001f) return false;
    float tt = (pa - oa) / da;
    if (tt >= 0 && tt < FLT_MAX) { t = tt; return true; }
    return false;
}

// Conservative pre-test for the reference's box gate below.  The exact test can only pass when the
// ray, at some parameter t >= 0, is inside the box to within the rounding of its own arithmetic
// (<= 2.4e-7 * (|o| + extent) per coordinate), so a slab test over t >= 0 on the box grown by
// pad = cull_pad * (|o|_inf + extent) (16x that bound) that MISSES proves the exact test fails.
// A pre-test that passes decides nothing: box_intersects still runs.  1/d may be approximate.
struct RayGate { float ix, iy, iz, pad, lx, ly, lz, hx, hy, hz; };

RT_FN float approx_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// The slab distances are affine in the box coordinate c: (c -+ pad - o) / d = c * (1/d) + (-o -+ pad)/d, so the
// pre-test is 6 fma + min/max (cf. make_cull / cull_box below).  Folded into the per-ray constants: the space
// margin pad, and E = 2^-18 * (extent*|1/d| + |l| + |h|) per axis, a bound (16x the true one) on the rounding of
// c*(1/d) + l itself for |c| <= extent — approximate reciprocal, two products, one fma — which moves each axis'
// entry distance earlier and its exit distance later.
RT_FN RayGate make_gate(float cull_pad, float extent, f3 o, f3 d)
{
    RayGate r;
    r.pad = cull_pad * (fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) + extent);
    float dx = fabsf(d.x) > 1e-20f ? d.x : (d.x < 0 ? -1e-20f : 1e-20f);
    float dy = fabsf(d.y) > 1e-20f ? d.y : (d.y < 0 ? -1e-20f : 1e-20f);
    float dz = fabsf(d.z) > 1e-20f ? d.z : (d.z < 0 ? -1e-20f : 1e-20f);
    r.ix = approx_rcp(dx); r.iy = approx_rcp(dy); r.iz = approx_rcp(dz);
    r.lx = (-o.x - r.pad) * r.ix; r.hx = (-o.x + r.pad) * r.ix;
    r.ly = (-o.y - r.pad) * r.iy; r.hy = (-o.y + r.pad) * r.iy;
    r.lz = (-o.z - r.pad) * r.iz; r.hz = (-o.z + r.pad) * r.iz;
    const float k = 1.0f / 262144.0f;
    const float ex = k * (extent * fabsf(r.ix) + fabsf(r.lx) + fabsf(r.hx));
    const float ey = k * (extent * fabsf(r.iy) + fabsf(r.ly) + fabsf(r.hy));
    const float ez = k * (extent * fabsf(r.iz) + fabsf(r.lz) + fabsf(r.hz));
    // d > 0: the lower plane is the entry (earlier by E), the upper one the exit (later by E); d < 0: the reverse
    r.lx -= r.ix >= 0 ? ex : -ex; r.hx += r.ix >= 0 ? ex : -ex;
    r.ly -= r.iy >= 0 ? ey : -ey; r.hy += r.iy >= 0 ? ey : -ey;
    r.lz -= r.iz >= 0 ? ez : -ez; r.hz += r.iz >= 0 ? ez : -ez;
    return r;
}

RT_FN bool gate_maybe(const float* mn, const float* mx, const RayGate& r)
{
    const float ax = fmaf(mn[0], r.ix, r.lx), bx = fmaf(mx[0], r.ix, r.hx);
    const float ay = fmaf(mn[1], r.iy, r.ly), by = fmaf(mx[1], r.iy, r.hy);
    const float az = fmaf(mn[2], r.iz, r.lz), bz = fmaf(mx[2], r.iz, r.hz);
    const float n = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float f = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return n <= f && f >= 0.0f;
}

// Conservative ACCEPT for the same gate.  box_intersects() is true as soon as ONE face test passes:
// |d_a| >= 1e-5, t = (c - o_a)/d_a in [0, FLT_MAX) and the point o + d*t strictly inside the face's
// rectangle on the other two axes.  Take the exit planes (c = mx_a for d_a > 0, else mn_a) of the axes
// with |d_a| safely above 1e-5, the smallest approximate parameter t among them, and the point there:
// if t is safely positive and the point is inside the rectangle by the margin pad (>= 6x the
// rounding of either evaluation, see gate_maybe), that face's exact test passes too.  Anything
// else (edge exits, tiny |d_a|, origin on a plane) is left to the exact test.
RT_FN bool gate_surely(const float* mn, const float* mx, f3 o, f3 d, const RayGate& r)
{
    const float tx = fabsf(d.x) > 1.0001e-5f ? ((d.x > 0 ? mx[0] : mn[0]) - o.x) * r.ix : FLT_MAX;
    const float ty = fabsf(d.y) > 1.0001e-5f ? ((d.y > 0 ? mx[1] : mn[1]) - o.y) * r.iy : FLT_MAX;
    const float tz = fabsf(d.z) > 1.0001e-5f ? ((d.z > 0 ? mx[2] : mn[2]) - o.z) * r.iz : FLT_MAX;
    const float t = fminf(fminf(tx, ty), tz);
    if (!(t > 1e-20f && t < 1e30f)) return false;
    const float hx = o.x + d.x * t, hy = o.y + d.y * t, hz = o.z + d.z * t;
    const bool inx = hx > mn[0] + r.pad && hx < mx[0] - r.pad;
    const bool iny = hy > mn[1] + r.pad && hy < mx[1] - r.pad;
    const bool inz = hz > mn[2] + r.pad && hz < mx[2] - r.pad;
    const bool sx = t == tx, sy = !sx && t == ty, sz = !sx && !sy;   // exactly one axis is the face's own
    return (sx || inx) && (sy || iny) && (sz || inz);
}

// Both conservative answers from ONE set of slab values (round 3): 0 = the reference's test fails for sure (gate_maybe
// false), 1 = it passes for sure, 2 = undecided (gate_surely, then the exact test, decide).  The accept costs ~25
// instructions on top of the reject's 6 fma + min/max where gate_surely recomputes exit planes and a point (~45): the
// gate walks of config 5 (~100 box gates per ray) and of the reference's own scenes are made of this test.
// With i_b = 1/d_b (approximate), the values of gate_maybe are, per axis b, en_b = N_b - m_b and ex_b = X_b + m_b up to a
// rounding rho_b <= e_b/16: N_b / X_b the true entry / exit parameters of the box's slab, m_b = pad*|i_b| + e_b the folded
// margin, and e_b <= 2.0001*pad*|i_b| because make_gate's E uses the same 2^-18 as cull_pad (DScene::cull_pad; asserted at
// commit).  Let a = the axis of the smallest ex (f = ex_a) and K_b = 8*pad*|i_b| >= 2*m_b + pad*|i_b| + 4*rho_b.  If, for
// both axes b != a,  ex_b - f > K_b + K_a  and  f - en_b > K_b + K_a,  then  X_b - X_a > pad*|i_b|  and  X_a - N_b >
// pad*|i_b|: at the exact parameter X_a at which the ray leaves through face a, its other two coordinates lie inside that
// face's rectangle by more than pad (>= 20x the rounding of the reference's own t and point, see gate_maybe), so the
// reference's test of THAT face passes (strict inequalities included) provided its other conditions hold: |d_a| >= 1e-5
// (all |i_b| < 99 990 is required), t >= 0 (f > K_a gives X_a > m_a > 0) and t < FLT_MAX (f < 1e30).  Boxes that are flat
// along a, exits near an edge, tiny direction components: class 2.  tests/test_emul.py::test_gate_classification_is_sound
// checks both implications on random and adversarial rays.
RT_FN int gate_class(const float* mn, const float* mx, const RayGate& r)
{
    const float ax = fmaf(mn[0], r.ix, r.lx), bx = fmaf(mx[0], r.ix, r.hx);
    const float ay = fmaf(mn[1], r.iy, r.ly), by = fmaf(mx[1], r.iy, r.hy);
    const float az = fmaf(mn[2], r.iz, r.lz), bz = fmaf(mx[2], r.iz, r.hz);
    const float enx = fminf(ax, bx), exx = fmaxf(ax, bx);
    const float eny = fminf(ay, by), exy = fmaxf(ay, by);
    const float enz = fminf(az, bz), exz = fmaxf(az, bz);
    const float n = fmaxf(fmaxf(enx, eny), enz);
    const float f = fminf(fminf(exx, exy), exz);
    if (!(n <= f && f >= 0.0f)) return 0;
    const float k8 = 8.0f * r.pad;
    const float kx = k8 * fabsf(r.ix), ky = k8 * fabsf(r.iy), kz = k8 * fabsf(r.iz);
    const bool isx = f == exx, isy = !isx && f == exy;   // exactly one axis is the exit face's own
    const float ks = isx ? kx : (isy ? ky : kz);
    const float imax = fmaxf(fmaxf(fabsf(r.ix), fabsf(r.iy)), fabsf(r.iz));
    const bool ok = imax < 99990.0f && f > ks && f < 1e30f;
    const bool cx = exx - f > kx + ks && f - enx > kx + ks;
    const bool cy = exy - f > ky + ks && f - eny > ky + ks;
    const bool cz = exz - f > kz + ks && f - enz > kz + ks;
    const bool isz = !isx && !isy;
    return (ok && (isx || cx) && (isy || cy) && (isz || cz)) ? 1 : 2;
}

// One face of BoundingVolumeBox::intersects (main.cpp:245-296).  For an axis-aligned unit normal
// the reference's dot products reduce exactly to (c - o[a]) / d[a].
RT_FN bool box_face(int axis, float c, const float* mn, const float* mx, f3 o, f3 d)
{
    float oa = axis == 0 ? o.x : (axis == 1 ? o.y : o.z);
    // denom = dot(unit axis, d) and dot(p - o, unit axis): the zero products vanish for finite rays
    float denom = axis == 0 ? d.x : (axis == 1 ? d.y : d.z);
    if (fabsf(denom) < 0.00001f) return false;
    float t = (c - oa) / denom;
    if (!(t >= 0 && t < FLT_MAX)) return false;
    f3 h = o + d * t;
    if (axis == 0) return h.y > mn[1] && h.y < mx[1] && h.z > mn[2] && h.z < mx[2];
    if (axis == 1) return h.x > mn[0] && h.x < mx[0] && h.z > mn[2] && h.z < mx[2];
    return h.x > mn[0] && h.x < mx[0] && h.y > mn[1] && h.y < mx[1];
}

// The six faces in the reference's order, one at a time in a loop that is NOT unrolled: this is the rare path of every
// gate (taken only when neither conservative test decides), but inlined six times with its IEEE divisions interleaved
// it set the register allocation of all the gate-walking kernels — and those run faster with every wave that fits
// (k_ao: 39.7 / 47.8 / 65.8 ms at 4 / 3 / 2 waves per SIMD).  Same tests, same order, same result.
RT_FN bool box_intersects(const float* mn, const float* mx, f3 o, f3 d)
{
#if defined(__clang__)
#pragma clang loop unroll(disable)
#endif
    for (int f = 0; f < 6; ++f) {
        const int axis = f >> 1;
        const float c = (f & 1) ? (axis == 0 ? mx[0] : (axis == 1 ? mx[1] : mx[2])) : (axis == 0 ? mn[0] : (axis == 1 ? mn[1] : mn[2]));
        const float oa = axis == 0 ? o.x : (axis == 1 ? o.y : o.z);
        const float denom = axis == 0 ? d.x : (axis == 1 ? d.y : d.z);
        if (fabsf(denom) < 0.00001f) continue;
        const float t = (c - oa) / denom;
        if (!(t >= 0 && t < FLT_MAX)) continue;
        const f3 h = o + d * t;
        // the two coordinates other than the face's own, strictly inside the face's rectangle
        const float p0 = axis == 0 ? h.y : h.x, lo0 = axis == 0 ? mn[1] : mn[0], hi0 = axis == 0 ? mx[1] : mx[0];
        const float p1 = axis == 2 ? h.y : h.z, lo1 = axis == 2 ? mn[1] : mn[2], hi1 = axis == 2 ? mx[1] : mx[2];
        if (p0 > lo0 && p0 < hi0 && p1 > lo1 && p1 < hi1) return true;
    }
    return false;
}

// BoundingVolumeBox::intersects (main.cpp:245-296) as the walks evaluate it: the two conservative answers first, the
// exact six-face test only for what they leave undecided.  Always the reference's result.  A macro (GNU statement
// expression), not a function: as a function taking the box by pointer — inlined or not — it made the compiler keep the
// kernels' by-value DScene argument in scratch memory (560 bytes per lane; k_ao 34 -> 65 ms).
#define RT_GATE_ENTER(mn_, mx_, o_, d_, rg_) \
    ({ const int rt_gc_ = gate_class(mn_, mx_, rg_); \
       rt_gc_ == 1 || (rt_gc_ == 2 && (gate_surely(mn_, mx_, o_, d_, rg_) || box_intersects(mn_, mx_, o_, d_))); })

// Sphere::intersectRay (main.cpp:600-633): far root not divided by 2a
RT_FN bool sphere_t(f3 centre, float radius_sqr, f3 o, f3 d, float& t)
{
    f3 tmp = o - centre;
    float a = dot(d, d);
    float b = 2.0f * dot(d, tmp);
    float c = dot(tmp, tmp) - radius_sqr;
    float disc = (b * b) - (4.0f * a * c);
    if (geq(disc, 0.0f)) {
        const float kEpsilon = 0.01f;
        float e = sqrtf(disc);
        float denom = 2.0f * a;
        float tt = (-b - e) / denom;
        if (geq(tt, kEpsilon)) { t = tt; return true; }
        tt = (-b + e);
        if (geq(tt, kEpsilon)) { t = tt; return true; }
    }
    return false;
}

// Triangle::intersectRay, 3-argument overload (main.cpp:713-739): single-sided, any sign of t
RT_FN bool tri_t(f3 v0, f3 v0v1, f3 v0v2, f3 o, f3 d, float& t, float& u, float& v)
{
    f3 pvec = cross(d, v0v2);
    float det = dot(v0v1, pvec);
    if (det < 0.000001f) return false;
    float invDet = 1.0f / det;
    f3 tvec = o - v0;
    float uu = dot(tvec, pvec) * invDet;
    if (uu < 0 || uu > 1) return false;
    f3 qvec = cross(tvec, v0v1);
    float vv = dot(d, qvec) * invDet;
    if (vv < 0 || uu + vv > 1) return false;
    t = dot(v0v2, qvec) * invDet;
    u = uu; v = vv;
    return true;
}

// `if (intersect && t < sr.t)` of every Shape::hit, plus the reference's first-wins order on ties
RT_FN bool closer(float t, uint32_t rank, const Hit& h) { return t < h.t || (t == h.t && h.prim != H_MISS && rank < h.rank); }

// ---- triangle BVH of one Mesh ------------------------------------------------------------------
// The reference scans every triangle of a Mesh whose box gate passed and keeps the smallest t of
// ANY sign, i.e. the query is a line, not a ray.  So the BVH is traversed over the
// interval (-inf, best.t): children are culled with a slab test padded by
// pad = cull_pad * (|o|_inf + extent) so that float rounding in the reference's Moller-Trumbore
// arithmetic can never accept a triangle whose box was culled.  Culling never decides a hit.
struct RayCull { float sx, sy, sz, lx, ly, lz, hx, hy, hz; };

// Node boxes are 16-bit grid indices q (coordinate = qbase + q*qscale, rounded outwards by the
// builder), so the slab distances are affine in q: t = q * (qscale/d) + (qbase - o -+ pad)/d.
// Two margins are folded into the per-ray constants, so the per-node test is bare min/max/compare:
//  * pad (space): the lower planes move by -pad, the upper ones by +pad;
//  * E (parameter): a bound on the rounding of q*s + l itself, 2^-18 * (65535|s| + |l| + |h|) per axis
//    (the true bound is ~2^-22 of that sum: approximate reciprocal, two products, one fma), subtracted
//    from the axis' entry distance and added to its exit distance (which plane is which follows from
//    the sign of d).
RT_FN RayCull make_cull_grid(float cull_pad, float extent, f3 qbase, f3 qscale, f3 o, f3 d);
RT_FN RayCull make_cull(const DScene& S, f3 o, f3 d) { return make_cull_grid(S.cull_pad, S.extent, S.qbase, S.qscale, o, d); }
// the same for any 16-bit grid (the triangle-BVH nodes' grid above, the compact gate nodes' grid: gate_step_q)
RT_FN RayCull make_cull_grid(float cull_pad, float extent, f3 qbase, f3 qscale, f3 o, f3 d)
{
    RayCull c;
    struct { float cull_pad, extent; f3 qbase, qscale; } S = {cull_pad, extent, qbase, qscale};
    float pad = S.cull_pad * (fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) + S.extent);
    float dx = fabsf(d.x) > 1e-20f ? d.x : (d.x < 0 ? -1e-20f : 1e-20f);
    float dy = fabsf(d.y) > 1e-20f ? d.y : (d.y < 0 ? -1e-20f : 1e-20f);
    float dz = fabsf(d.z) > 1e-20f ? d.z : (d.z < 0 ? -1e-20f : 1e-20f);
    // approximate reciprocals are enough: their 1-ulp error is inside E
    float ix = approx_rcp(dx), iy = approx_rcp(dy), iz = approx_rcp(dz);
    float bx = S.qbase.x - o.x, by = S.qbase.y - o.y, bz = S.qbase.z - o.z;
    c.sx = S.qscale.x * ix; c.sy = S.qscale.y * iy; c.sz = S.qscale.z * iz;
    c.lx = (bx - pad) * ix; c.hx = (bx + pad) * ix;
    c.ly = (by - pad) * iy; c.hy = (by + pad) * iy;
    c.lz = (bz - pad) * iz; c.hz = (bz + pad) * iz;
    const float k = 1.0f / 262144.0f;
    float ex = k * (65535.0f * fabsf(c.sx) + fabsf(c.lx) + fabsf(c.hx));
    float ey = k * (65535.0f * fabsf(c.sy) + fabsf(c.ly) + fabsf(c.hy));
    float ez = k * (65535.0f * fabsf(c.sz) + fabsf(c.lz) + fabsf(c.hz));
    // d > 0: the lower plane is the entry (earlier by E), the upper one the exit (later by E); d < 0: the reverse
    c.lx -= ix >= 0 ? ex : -ex; c.hx += ix >= 0 ? ex : -ex;
    c.ly -= iy >= 0 ? ey : -ey; c.hy += iy >= 0 ? ey : -ey;
    c.lz -= iz >= 0 ? ez : -ez; c.hz += iz >= 0 ? ez : -ez;
    return c;
}

RT_FN float q_lo(float w) { return (float)(f2u(w) & 0xFFFFu); }
RT_FN float q_hi(float w) { return (float)(f2u(w) >> 16); }

RT_FN bool cull_box(const RayCull& c, float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                    float tbest, float& tnear)
{
    float ax = fmaf(mnx, c.sx, c.lx), bx = fmaf(mxx, c.sx, c.hx);
    float ay = fmaf(mny, c.sy, c.ly), by = fmaf(mxy, c.sy, c.hy);
    float az = fmaf(mnz, c.sz, c.lz), bz = fmaf(mxz, c.sz, c.hz);
    float n = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float f = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    tnear = n;
    return n <= f && n <= tbest;
}

// both child boxes of a 32-byte TWO-wide node {w0..w3}{w4,w5,ref0,ref1}: w = lo | hi<<16 over the sequence
// lo0.x lo0.y lo0.z hi0.x hi0.y hi0.z lo1.x ... hi1.z.  This is the builders' intermediate format (binned SAH on
// the host, LBVH on the device); the traversal reads the four-wide nodes that collapse_node() makes of it.
RT_FN void node_test(const RayCull& rc, q4 n0, q4 n1, float tbest, bool& h0, bool& h1, float& t0, float& t1, uint32_t& r0, uint32_t& r1)
{
    r0 = f2u(n1.z); r1 = f2u(n1.w);
    h0 = cull_box(rc, q_lo(n0.x), q_hi(n0.x), q_lo(n0.y), q_hi(n0.y), q_lo(n0.z), q_hi(n0.z), tbest, t0);
    h1 = cull_box(rc, q_lo(n0.w), q_hi(n0.w), q_lo(n1.x), q_hi(n1.x), q_lo(n1.y), q_hi(n1.y), tbest, t1);
}

// ---- four-wide nodes (64 bytes: the unit the traversal kernels fetch) ------------------------------------
// Node = {c0.x c0.y c0.z c1.x} {c1.y c1.z c2.x c2.y} {c2.z c3.x c3.y c3.z} {ref0 ref1 ref2 ref3}; word c.a = mn | mx<<16
// (grid indices of child c's box on axis a).  An unused slot holds the inverted box mn = 65535, mx = 0, which no ray
// near the scene passes, and `empty_ref`: a leaf of one degenerate triangle (all-zero record: det = 0, never hit) that
// the builders append for this purpose — the margins of the box test grow with |o|, so a ray from very far away may
// pass even an inverted box; it then tests a triangle that cannot be hit, and the traversal needs no "is there a
// child" test.
// collapse_node(): two-wide node i -> four-wide node i.  Child k of the two-wide node fills slots 2k, 2k+1: a leaf
// fills one slot; an inner child is replaced by ITS two children (their boxes are tighter than the child's own).
// Nodes on odd levels of the two-wide tree are no longer referenced.  Pure re-packing of integers: the host
// builder and the device builder (rt_lbvh.hip) run the same function.
RT_FN void collapse_node(const q4* nodes2, uint32_t i, uint32_t empty_ref, q4* out)
{
    uint32_t w[12], ref[4];
    for (int k = 0; k < 12; ++k) w[k] = 0x0000FFFFu;   // mn = 65535, mx = 0
    for (int k = 0; k < 4; ++k) ref[k] = empty_ref;
    auto unpack = [](q4 a, q4 b, uint32_t* q) {
        const uint32_t v[6] = {f2u(a.x), f2u(a.y), f2u(a.z), f2u(a.w), f2u(b.x), f2u(b.y)};
        for (int k = 0; k < 6; ++k) { q[2 * k] = v[k] & 0xFFFFu; q[2 * k + 1] = v[k] >> 16; }
    };
    auto put = [&](int slot, const uint32_t* q6, uint32_t r) {   // q6 = {mn.xyz, mx.xyz}
        for (int a = 0; a < 3; ++a) w[slot * 3 + a] = q6[a] | (q6[3 + a] << 16);
        ref[slot] = r;
    };
    const q4 a0 = nodes2[(size_t)i * 2], a1 = nodes2[(size_t)i * 2 + 1];
    uint32_t q[12];
    unpack(a0, a1, q);
    const uint32_t r2[2] = {f2u(a1.z), f2u(a1.w)};
    for (int k = 0; k < 2; ++k) {
        if (r2[k] == REF_NONE) continue;   // the absent second child of a one-leaf root
        if (r2[k] & REF_LEAF) { put(2 * k, q + 6 * k, r2[k]); continue; }
        const q4 b0 = nodes2[(size_t)r2[k] * 2], b1 = nodes2[(size_t)r2[k] * 2 + 1];
        uint32_t g[12];
        unpack(b0, b1, g);
        const uint32_t gr[2] = {f2u(b1.z), f2u(b1.w)};
        for (int j = 0; j < 2; ++j) if (gr[j] != REF_NONE) put(2 * k + j, g + 6 * j, gr[j]);
    }
    out[0] = {u2f(w[0]), u2f(w[1]), u2f(w[2]), u2f(w[3])};
    out[1] = {u2f(w[4]), u2f(w[5]), u2f(w[6]), u2f(w[7])};
    out[2] = {u2f(w[8]), u2f(w[9]), u2f(w[10]), u2f(w[11])};
    out[3] = {u2f(ref[0]), u2f(ref[1]), u2f(ref[2]), u2f(ref[3])};
}

// Per-ray constants of the four-wide node test.  Same conservative slab test as cull_box, rearranged so that a
// box costs 6 byte-permutes + 3 packed fmas + max3 + min3:
//  * which plane of an axis is the entry and which the exit follows from the sign of d, so the ray carries, per
//    axis, byte selectors that pick the entry / exit plane's 16 bits out of the word and the matching additive
//    constants: no min/max per axis;
//  * v_perm_b32 drops the 16 bits q into the mantissa of 2^23 (0x4B000000 | q = 2^23 + q exactly), so no integer
//    conversion is needed: t = (2^23 + q) * s + (c - 2^23 * s).  The shifted constant is rounded to ~|s|/2 (half a
//    grid step in t), so the entry constant is moved earlier and the exit constant later by delta >= that bound
//    (0.51 |s| + 2^-20 |c|) on top of cull_box's margins — one more grid step of looseness, never a lost box;
//  * entry and exit distance of an axis are the two halves of one packed fma.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float rt_v2f __attribute__((ext_vector_type(2)));   // a 64-bit register pair: the operand of v_pk_fma_f32
#else
struct rt_v2f { float x, y; };
#endif
// per axis: c = {entry constant, exit constant}; sp = {s = qscale/d, bits(byte selector of the ENTRY plane)}.  Round 3: the
// scale is held once (the packed fma reads the low half of `sp` for both of its halves, op_sel_hi) and the exit plane's
// selector is the entry selector ^ 0x0202 (RT_PERM_LO ^ RT_PERM_HI), computed per use: 12 registers of per-ray state
// instead of 18, for three more xors per node step — k_bvh's time falls with every wave that fits.
struct RayCull4 { rt_v2f cx, cy, cz; rt_v2f spx, spy, spz; };
#define RT_PERM_LO 0x07040100u   // bytes {w.b0, w.b1, 0x00, 0x4B} of the pair {0x4B000000, w}
#define RT_PERM_HI 0x07040302u   // bytes {w.b2, w.b3, 0x00, 0x4B}
#define RT_PERM_FLIP 0x00000202u

RT_FN RayCull4 make_cull4_from(const RayCull& c);
RT_FN RayCull4 make_cull4(const DScene& S, f3 o, f3 d) { return make_cull4_from(make_cull(S, o, d)); }
RT_FN RayCull4 make_cull4_from(const RayCull& c)
{
    RayCull4 r;
    const float big = 8388608.0f, kd = 0.51f, kc = 1.0f / 1048576.0f;
    // s >= 0 <=> d > 0 (qscale > 0): the lower plane (constant l, already moved earlier) is the entry
    {
        const bool pos = !(c.sx < 0.0f);
        const float ne = pos ? c.lx : c.hx, fa = pos ? c.hx : c.lx;
        r.spx.x = c.sx; r.spx.y = u2f(pos ? RT_PERM_LO : RT_PERM_HI);
        r.cx.x = (ne - (kd * fabsf(c.sx) + kc * fabsf(ne))) - big * c.sx;
        r.cx.y = (fa + (kd * fabsf(c.sx) + kc * fabsf(fa))) - big * c.sx;
    }
    {
        const bool pos = !(c.sy < 0.0f);
        const float ne = pos ? c.ly : c.hy, fa = pos ? c.hy : c.ly;
        r.spy.x = c.sy; r.spy.y = u2f(pos ? RT_PERM_LO : RT_PERM_HI);
        r.cy.x = (ne - (kd * fabsf(c.sy) + kc * fabsf(ne))) - big * c.sy;
        r.cy.y = (fa + (kd * fabsf(c.sy) + kc * fabsf(fa))) - big * c.sy;
    }
    {
        const bool pos = !(c.sz < 0.0f);
        const float ne = pos ? c.lz : c.hz, fa = pos ? c.hz : c.lz;
        r.spz.x = c.sz; r.spz.y = u2f(pos ? RT_PERM_LO : RT_PERM_HI);
        r.cz.x = (ne - (kd * fabsf(c.sz) + kc * fabsf(ne))) - big * c.sz;
        r.cz.y = (fa + (kd * fabsf(c.sz) + kc * fabsf(fa))) - big * c.sz;
    }
    return r;
}

// 0x4B000000 | (the half of w that `sel` names) as a float: 2^23 + q
RT_FN float perm_q(uint32_t w, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return u2f(__builtin_amdgcn_perm(0x4B000000u, w, sel));
#else
    return u2f(0x4B000000u | ((sel & 0xFFu) == 0u ? (w & 0xFFFFu) : (w >> 16)));
#endif
}

// entry / exit distance of one axis: {near, far} = {2^23 + q_near, 2^23 + q_far} * s + {c.x, c.y}
// sel_far = the exit plane's selector when the caller holds it in a register (k_bvh, which has registers to spare at its
// occupancy), 0 = derive it from the entry selector here (one xor per axis and node)
RT_FN void slab_axis(uint32_t w, rt_v2f sp, rt_v2f c, float& tn, float& tf, uint32_t sel_far_held = 0u)
{
    const uint32_t sel_near = f2u(sp.y), sel_far = sel_far_held ? sel_far_held : sel_near ^ RT_PERM_FLIP;
#if defined(__HIP_DEVICE_COMPILE__)
    const rt_v2f q = {perm_q(w, sel_near), perm_q(w, sel_far)};
    rt_v2f t;
    // v_pk_fma_f32 with op_sel_hi = [1,0,1]: the high half of the result uses the LOW half of `sp` (the scale) as well
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(q), "v"(sp), "v"(c));
    tn = t.x; tf = t.y;
#else
    tn = fmaf(perm_q(w, sel_near), sp.x, c.x);
    tf = fmaf(perm_q(w, sel_far), sp.x, c.y);
#endif
}

struct SelFar { uint32_t x, y, z; };   // the three exit-plane selectors, or zeros (see slab_axis)
RT_FN bool cull_box4(const RayCull4& rc, uint32_t wx, uint32_t wy, uint32_t wz, float tbest, float& tnear, SelFar sf = SelFar{0u, 0u, 0u})
{
    float ax, bx, ay, by, az, bz;
    slab_axis(wx, rc.spx, rc.cx, ax, bx, sf.x);
    slab_axis(wy, rc.spy, rc.cy, ay, by, sf.y);
    slab_axis(wz, rc.spz, rc.cz, az, bz, sf.z);
    const float n = fmaxf(fmaxf(ax, ay), az), f = fminf(fminf(bx, by), bz);
    tnear = n;
    return n <= f && n <= tbest;
}

// The conservative pre-test of a box GATE (gate_maybe) on a box given as three packed 16-bit words of the compact gate
// nodes' grid: the same padded slab test over t >= 0, with the per-ray constants of that grid (make_cull_grid with
// DScene::gq_base / gq_scale; its margins contain gate_maybe's).  False proves the reference's 6-plane test false.
RT_FN bool gate_maybe4(const RayCull4& rc, uint32_t wx, uint32_t wy, uint32_t wz)
{
    float ax, bx, ay, by, az, bz;
    slab_axis(wx, rc.spx, rc.cx, ax, bx);
    slab_axis(wy, rc.spy, rc.cy, ay, by);
    slab_axis(wz, rc.spz, rc.cz, az, bz);
    const float n = fmaxf(fmaxf(ax, ay), az), f = fminf(fminf(bx, by), bz);
    return n <= f && f >= 0.0f;
}

// the four child boxes of a 64-byte node (every slot holds a valid reference: see collapse_node)
RT_FN void node_test4(const RayCull4& rc, q4 n0, q4 n1, q4 n2, q4 n3, float tbest, bool* h, float* t, uint32_t* r, SelFar sf = SelFar{0u, 0u, 0u})
{
    r[0] = f2u(n3.x); r[1] = f2u(n3.y); r[2] = f2u(n3.z); r[3] = f2u(n3.w);
    h[0] = cull_box4(rc, f2u(n0.x), f2u(n0.y), f2u(n0.z), tbest, t[0], sf);
    h[1] = cull_box4(rc, f2u(n0.w), f2u(n1.x), f2u(n1.y), tbest, t[1], sf);
    h[2] = cull_box4(rc, f2u(n1.z), f2u(n1.w), f2u(n2.x), tbest, t[2], sf);
    h[3] = cull_box4(rc, f2u(n2.y), f2u(n2.z), f2u(n2.w), tbest, t[3], sf);
}

// compare-exchange of (distance, reference) pairs: afterwards ta <= tb
RT_FN void order2(float& ta, uint32_t& ra, float& tb, uint32_t& rb)
{
    const bool sw = tb < ta;
    const float t = sw ? tb : ta, u = sw ? ta : tb;
    const uint32_t r = sw ? rb : ra, q = sw ? ra : rb;
    ta = t; tb = u; ra = r; rb = q;
}

// One node: tests the four children and sorts them by entry distance (missed ones get distance +inf and sort
// last).  Returns the number of hit children, which are r[0 .. n): r[0] is the nearest.
RT_FN uint32_t node_children(const DScene& S, const RayCull4& rc, uint32_t node, float tbest, float* t, uint32_t* r)
{
    const q4* np = S.nodes + (size_t)node * 4;
    const q4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    bool h[4];
    node_test4(rc, n0, n1, n2, n3, tbest, h, t, r);
    uint32_t nh = 0;
    for (int c = 0; c < 4; ++c) {
        nh += h[c] ? 1u : 0u;
        t[c] = h[c] ? t[c] : INFINITY;   // a hit child's entry distance is <= tbest < inf
    }
    order2(t[0], r[0], t[1], r[1]); order2(t[2], r[2], t[3], r[3]);
    order2(t[0], r[0], t[2], r[2]); order2(t[1], r[1], t[3], r[3]);
    order2(t[1], r[1], t[2], r[2]);
    return nh;
}

// The same test for the compacted kernel (rt_bvh.hip), sorted the other way round: hit children FARTHEST first,
// missed ones (distance -inf) last, so that a lane can store all four {reference, distance} pairs at consecutive
// stack positions without predication and advance its stack pointer by the number of hits: the nearest hit child
// ends on top, what lies beyond the new top is never read.
RT_FN uint32_t node_children_desc(const DScene& S, const RayCull4& rc, uint32_t node, float tbest, float* t, uint32_t* r, SelFar sf = SelFar{0u, 0u, 0u})
{
    const q4* np = S.nodes + (size_t)node * 4;
    const q4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    bool h[4];
    node_test4(rc, n0, n1, n2, n3, tbest, h, t, r, sf);
    uint32_t nh = 0;
    for (int c = 0; c < 4; ++c) {
        nh += h[c] ? 1u : 0u;
        t[c] = h[c] ? t[c] : -INFINITY;
    }
    order2(t[1], r[1], t[0], r[0]); order2(t[3], r[3], t[2], r[2]);
    order2(t[2], r[2], t[0], r[0]); order2(t[3], r[3], t[1], r[1]);
    order2(t[2], r[2], t[1], r[1]);
    return nh;
}

template <bool STATS, bool ANYHIT, int DEPTH = RT_STACK_DEPTH>
RT_FN bool bvh_drain(const DScene& S, f3 o, f3 d, const RayCull4& rc, Hit& best, Stack st, uint32_t& sp, Counters& cnt)
{
    // Pops references until the stack is empty.  The stack holds the roots of every Mesh whose reference gate
    // passed plus deferred children (inner nodes and leaves alike).  The plain loop used wherever a ray is
    // traversed in place (shadow rays, queue overflow, the host emulation); the compacted kernel k_bvh
    // (rt_bvh.hip) walks the same nodes with its vote scheme and finds the same closest hit (ties by rank).
    bool found = false;
    if (sp == 0) return false;
    if (S.debug & 1u) { sp = 0; return false; }
    while (sp > 0) {
        const uint32_t cur = st.p[(--sp) * st.stride];
        if (cur == REF_NONE) continue;
        if (!(cur & REF_LEAF)) {
            float t[4]; uint32_t r[4];
            if (STATS) cnt.nodes++;
            const uint32_t nh = node_children(S, rc, cur, best.t, t, r);
            // farthest first, so that the nearest child is popped next; depth is bounded by the builders
            for (int c = 3; c >= 0; --c)
                if ((uint32_t)c < nh && sp < (uint32_t)DEPTH) st.p[(sp++) * st.stride] = r[c];
            continue;
        }
        // cur is a leaf: up to 8 triangles stored contiguously
        const uint32_t first = cur & 0x0FFFFFFFu, count = ((cur >> 28) & 7u) + 1u;
        for (uint32_t i = 0; i < count; ++i) {
            const q4* T = S.tris + (size_t)(first + i) * 3;
            q4 a = T[0], b = T[1], e = T[2];
            if (STATS) cnt.tris++;
            float t, u, v;
            if (tri_t(xyz(a), xyz(b), xyz(e), o, d, t, u, v)) {
                uint32_t rank = f2u(e.w);
                if (closer(t, rank, best)) {
                    best.t = t; best.prim = H_TRI | (first + i); best.u = u; best.v = v; best.rank = rank;
                    found = true;
                    if (ANYHIT) { sp = 0; return true; }
                }
            }
        }
    }
    return found;
}

// Gate node i: from the kernel's LDS copy when it is staged there (explicit LDS address space, so
// the two sources are not folded into one generic-pointer load), else from global memory.
RT_FN GateNode load_gate(const DScene& S, uint32_t i)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (i < S.num_gates_fast) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const __attribute__((address_space(3))) v4f* p = (const __attribute__((address_space(3))) v4f*)(S.gates_fast + i);
        union { v4f v[3]; GateNode g; } u;
        u.v[0] = p[0]; u.v[1] = p[1]; u.v[2] = p[2];
        return u.g;
    }
#endif
    return S.gates[i];
}

// The gate array's leading run of top-level Plane nodes (the room of the reference's scenes), read from the kernel
// arguments in a wave-uniform loop (DScene::lead_planes).  Returns the index of the first node after them.
// An axis plane that cannot become the closest hit is dropped before the IEEE division (6 per ray otherwise, for one
// winner): with the approximate reciprocal of d[a] that the ray's gate constants hold (relative error <= 2^-22), the
// approximate parameter decides when it is negative or beyond the best hit by a margin of 16x that error; everything
// else — the candidates, parameters near 0, a best hit that is not positive — takes the exact evaluation.
RT_FN bool plane_axis_cannot_win(float pa, float oa, float inv_da, float tbest)
{
    const float ta = (pa - oa) * inv_da;
    return (ta < -1e-30f) || (tbest > 1e-30f && ta > tbest * 1.000004f + 1e-30f);
}

// DIST (the occlusion walks): only the distance of the closest plane is kept (see gate_step).
template <bool DIST = false>
RT_FN uint32_t lead_planes_hit(const DScene& S, f3 o, f3 d, const RayGate& rg, Hit& best)
{
    for (uint32_t j = 0; j < S.num_lead_planes; ++j) {
        const float* pl = S.lead_planes[j];
        float t;
        bool hit;
        switch ((S.lead_axis >> (4u * j)) & 15u) {   // wave-uniform
        case 0: hit = !plane_axis_cannot_win(pl[0], o.x, rg.ix, best.t) && plane_axis_t(pl[0], o.x, d.x, t); break;
        case 1: hit = !plane_axis_cannot_win(pl[1], o.y, rg.iy, best.t) && plane_axis_t(pl[1], o.y, d.y, t); break;
        case 2: hit = !plane_axis_cannot_win(pl[2], o.z, rg.iz, best.t) && plane_axis_t(pl[2], o.z, d.z, t); break;
        default: hit = plane_t(F3(pl[0], pl[1], pl[2]), F3(pl[3], pl[4], pl[5]), o, d, t); break;
        }
        if (DIST) { if (hit && t < best.t) best.t = t; }
        else if (hit && closer(t, f2u(pl[6]), best)) {
            best.t = t; best.prim = H_PLANE | f2u(pl[7]); best.rank = f2u(pl[6]);
        }
    }
    return S.num_lead_planes;
}

// The gate walk of an occlusion (AO) ray, resumable: closest hit among the primitives that are tested in place
// (planes, spheres, free triangles) into `best`; the triangle-BVH root of every Mesh whose reference gates pass is
// pushed on `st`.  Walks from node i until the array ends or RT_DEFER_ROOTS roots wait; returns the index to
// resume at (>= S.num_gates: done).  Nothing is traversed here: the caller queues the roots for k_bvh and calls
// again, so this walk needs 8 stack entries and no traversal code (the kernel that runs it stays small).
// One node of that walk: tests gate node i and returns the index of the node to visit next.
// DIST (the occlusion walks): only the distance of the closest primitive leaves the walk, so the reference's first-wins
// rule on equal t cannot matter and the record's primitive / rank / uv are not kept (fewer registers through the walk).
// (Measured and dropped, round 3: leaving a box gate that neither conservative answer decides open until the end of
// k_ao_walk's round, so that gate_surely and the exact test run once for all such lanes instead of in every step in which
// one lane needs them: 38.9 -> 39.2 ms on config 5.)
// FAST = false: the kernel has no LDS copy of the first gate nodes (k_ao_walk) — a template flag, not DScene::num_gates_fast
// = 0 written by the kernel: a kernel that writes its by-value DScene makes the compiler keep a private copy of it, and
// whether that copy is split into registers or left in scratch memory (600 bytes per lane) has flipped with unrelated edits.
template <bool STATS, bool DIST = false, bool FAST = true>
RT_FN uint32_t gate_step(const DScene& S, f3 o, f3 d, const RayGate& rg, uint32_t i, Hit& best, Stack st, uint32_t& sp, Counters& cnt)
{
    {
        // the node's first 32 bytes (box / primitive, kind, skip); its last 16 (index, rank, ...) only where they are needed
        const GateNode* gp = S.gates + i;
        q4 g0, g1;
#if defined(__HIP_DEVICE_COMPILE__)
        if (FAST && i < S.num_gates_fast) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const __attribute__((address_space(3))) v4f* p = (const __attribute__((address_space(3))) v4f*)(S.gates_fast + i);
            const v4f a = p[0], b = p[1];
            g0 = {a.x, a.y, a.z, a.w}; g1 = {b.x, b.y, b.z, b.w};
        } else
#endif
        { g0 = ((const q4*)gp)[0]; g1 = ((const q4*)gp)[1]; }
        auto tail = [&]() -> q4 {   // {a, rank, prim, item}
#if defined(__HIP_DEVICE_COMPILE__)
            if (FAST && i < S.num_gates_fast) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f c = ((const __attribute__((address_space(3))) v4f*)(S.gates_fast + i))[2];
                return q4{c.x, c.y, c.z, c.w};
            }
#endif
            return ((const q4*)gp)[2];
        };
        const float mn[3] = {g0.x, g0.y, g0.z}, mx[3] = {g1.x, g1.y, g1.z};
        const uint32_t kindw = f2u(g0.w), skip = f2u(g1.w);
        auto take = [&](float t, uint32_t prim, uint32_t rank, float u, float v, bool uv) {
            if (DIST) { if (t < best.t) best.t = t; return; }
            if (closer(t, rank, best)) { best.t = t; best.prim = prim; best.rank = rank; if (uv) { best.u = u; best.v = v; } }
        };
        const uint32_t kind = kindw & G_KIND_MASK;
        // The two box-gated kinds share ONE copy of the box test: lanes on inner gates and lanes on Mesh gates (about half
        // each in a walk over a BVHAccel of Meshes: config 5) then take it together instead of one kind after the other.
        static_assert(G_GATED == 1u && G_MESH == 5u && G_TRI == 4u, "the two box-gated kinds are the ones with (kind & 3) == 1");
        if ((kindw & 3u) == 1u) {
            if (STATS) cnt.gates++;
            const bool pass = (kindw & G_SAME_BOX) || RT_GATE_ENTER(mn, mx, o, d, rg);
            if (!(kindw & 4u)) return pass ? i + 1 : skip;   // G_GATED
            if (pass) st.p[(sp++) * st.stride] = skip;   // a Mesh node's `skip` word is its root: no second fetch (the node's last 16 bytes) in the step that queues it
            return i + 1;
        }
        switch (kind) {
        case G_PLANE: {
            float t;
            if (plane_t(F3(mn[0], mn[1], mn[2]), F3(mx[0], mx[1], mx[2]), o, d, t)) { const q4 c = tail(); take(t, H_PLANE | f2u(c.x), f2u(c.y), 0, 0, false); }
            break;
        }
        case G_SPHERE: {
            float t;
            if (sphere_t(F3(mn[0], mn[1], mn[2]), mx[0], o, d, t)) { const q4 c = tail(); take(t, H_SPHERE | f2u(c.x), f2u(c.y), 0, 0, false); }
            break;
        }
        case G_TRI: {
            const q4 c = tail();
            const q4* T = S.tris + (size_t)f2u(c.x) * 3;
            q4 a = T[0], b = T[1], e = T[2];
            if (STATS) cnt.tris++;
            float t, u, v;
            if (tri_t(xyz(a), xyz(b), xyz(e), o, d, t, u, v)) take(t, H_TRI | f2u(c.x), f2u(c.y), u, v, true);
            break;
        }
        default:   // G_GROUP
            break;
        }
        i++;
    }
    return i;
}

// gate_step for the primitives tested in place only (the caller has dealt with groups and box gates)
template <bool STATS>
RT_FN uint32_t gate_step_prim(const DScene& S, f3 o, f3 d, uint32_t i, Hit& best, Counters& cnt)
{
    const GateNode g = S.gates[i];
    const uint32_t kind = g.kind & G_KIND_MASK;
    float t, u = 0.0f, v = 0.0f;
    bool hit = false;
    uint32_t prim = H_MISS;
    if (kind == G_PLANE) { hit = plane_t(F3(g.mn[0], g.mn[1], g.mn[2]), F3(g.mx[0], g.mx[1], g.mx[2]), o, d, t); prim = H_PLANE | g.a; }
    else if (kind == G_SPHERE) { hit = sphere_t(F3(g.mn[0], g.mn[1], g.mn[2]), g.mx[0], o, d, t); prim = H_SPHERE | g.a; }
    else if (kind == G_TRI) {
        const q4* T = S.tris + (size_t)g.a * 3;
        const q4 a = T[0], b = T[1], e = T[2];
        if (STATS) cnt.tris++;
        hit = tri_t(xyz(a), xyz(b), xyz(e), o, d, t, u, v); prim = H_TRI | g.a;
    }
    if (hit && closer(t, g.rank, best)) {
        best.t = t; best.prim = prim; best.rank = g.rank;
        if (kind == G_TRI) { best.u = u; best.v = v; }   // planes and spheres leave the record's uv as it was
    }
    return i + 1u;
}

// The same step with the node taken from the compact copy `qg` (QGate: LDS on the GPU).  Box gates are classified from
// the 16 bytes alone when the decoded (outward-rounded) box is missed; a box that may be entered is decided on the exact
// box from global memory as in gate_step; planes, spheres and free triangles take gate_step itself.  Same results.
template <bool STATS>
RT_FN uint32_t gate_step_q(const DScene& S, const QGate* qg, f3 o, f3 d, const RayGate& rg, const RayCull4& gc, uint32_t i, Hit& best, Stack st, uint32_t& sp, Counters& cnt)
{
    const QGate q = qg[i];
    const uint32_t kind = (q.w >> 24) & 7u;
    if (kind != G_GATED && kind != G_MESH) return kind == G_GROUP ? i + 1u : gate_step_prim<STATS>(S, o, d, i, best, cnt);
    if (STATS) cnt.gates++;
    bool pass = (q.w & (1u << 27)) != 0u;   // G_SAME_BOX: the enclosing gate's test has just passed
    uint32_t root = q.x;
    if (!pass && gate_maybe4(gc, q.x, q.y, q.z)) {
        const GateNode* g = S.gates + i;   // the exact box (and, for a Mesh, its root): 32 / 48 bytes from global memory
        const q4 b0 = ((const q4*)g)[0], b1 = ((const q4*)g)[1];
        const float emn[3] = {b0.x, b0.y, b0.z}, emx[3] = {b1.x, b1.y, b1.z};
        pass = gate_surely(emn, emx, o, d, rg) || box_intersects(emn, emx, o, d);
        if (pass && kind == G_MESH) root = g->a;
    }
    if (kind == G_GATED) return pass ? i + 1u : (q.w & 0xFFFFFFu);
    if (pass) st.p[(sp++) * st.stride] = root;
    return i + 1u;
}
// (Measured and dropped, round 3: the step split in two — up to 1 / 2 / 3 / 5 hops over nodes that the compact copy
// decides alone before the lanes meet at the exact test — 58.6 / 53.2 / 57.8 / 69.5 ms against 47.2 for this form on
// config 5: the extra ballots and the second read of the node cost more than the fuller lanes of the exact test gain.)

template <bool STATS, bool DIST = false>
RT_FN uint32_t walk_gates(const DScene& S, f3 o, f3 d, const RayGate& rg, uint32_t i, Hit& best, Stack st, uint32_t& sp, Counters& cnt)
{
    const uint32_t n = S.num_gates;
    while (i < n && sp < RT_DEFER_ROOTS) i = gate_step<STATS, DIST>(S, o, d, rg, i, best, st, sp, cnt);
    return i;
}

// The same walk made by a whole WAVE in step (GPU only; round 3).  On the reference's own scenes (a BVHAccel over a dozen
// shapes of four kinds: configs 1, 2 and 4) the lanes of a wave soon sit on different gate nodes of different kinds, so the
// per-lane walk above executes the box-gate, sphere, triangle and mesh branches one after the other with a few lanes each
// (52 % active lanes, 1 775 vector instructions per AO ray against 610 on config 3).  Here ONE node index j runs through the
// array in preorder and the lanes whose own next node is j (every lane's node index only grows) test it together: the
// node is wave-uniform, so it comes through scalar loads from the constant address space into scalar registers (no vector
// registers, no LDS), its kind is a scalar branch, and a wave pays each node of the UNION of its lanes' walks once.  Pays
// when that union is short (a gate array that fits the kernels' LDS staging limit, RT_LDS_GATES) — with thousands of nodes
// (config 5) the union is far longer than any one walk.  Every lane visits exactly the nodes of walk_gates, in the same
// order, with the same tests: same results.  Must be called by all 64 lanes; lanes with nothing to walk pass i >= num_gates.
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_CONST_AS __attribute__((address_space(4)))
template <bool STATS>
__device__ inline uint32_t walk_gates_wave(const DScene& S, f3 o, f3 d, const RayGate& rg, uint32_t i, float& best_t, Stack st, uint32_t& sp, Counters& cnt)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const RT_CONST_AS v4f* G = (const RT_CONST_AS v4f*)(uintptr_t)S.gates;
    const RT_CONST_AS v4f* TR = (const RT_CONST_AS v4f*)(uintptr_t)S.tris;
    const uint32_t n = S.num_gates;
    for (uint32_t j = S.num_lead_planes; j < n; ++j) {   // wave-uniform
        const bool mine = i == j && sp < RT_DEFER_ROOTS;
        if (__builtin_amdgcn_ballot_w64(mine) == 0ull) continue;
        const v4f g0 = G[3u * j], g1 = G[3u * j + 1u];
        const float mn[3] = {g0.x, g0.y, g0.z}, mx[3] = {g1.x, g1.y, g1.z};
        const uint32_t kindw = f2u(g0.w), skip = f2u(g1.w);
        switch (kindw & G_KIND_MASK) {   // a scalar branch
        case G_GROUP:
            if (mine) i = j + 1u;
            break;
        case G_GATED:
            if (mine) {
                if (STATS) cnt.gates++;
                i = ((kindw & G_SAME_BOX) || RT_GATE_ENTER(mn, mx, o, d, rg)) ? j + 1u : skip;
            }
            break;
        case G_PLANE:
            if (mine) {
                float t;
                if (plane_t(F3(mn[0], mn[1], mn[2]), F3(mx[0], mx[1], mx[2]), o, d, t) && t < best_t) best_t = t;
                i = j + 1u;
            }
            break;
        case G_SPHERE:
            if (mine) {
                float t;
                if (sphere_t(F3(mn[0], mn[1], mn[2]), mx[0], o, d, t) && t < best_t) best_t = t;
                i = j + 1u;
            }
            break;
        case G_TRI: {
            const uint32_t slot = f2u(G[3u * j + 2u].x);
            const v4f a = TR[(size_t)slot * 3], b = TR[(size_t)slot * 3 + 1], e = TR[(size_t)slot * 3 + 2];
            if (mine) {
                if (STATS) cnt.tris++;
                float t, u, v;
                if (tri_t(F3(a.x, a.y, a.z), F3(b.x, b.y, b.z), F3(e.x, e.y, e.z), o, d, t, u, v) && t < best_t) best_t = t;
                i = j + 1u;
            }
            break;
        }
        default: {   // G_MESH
            const uint32_t root = skip;   // a Mesh node's `skip` word is its root (no third scalar load)
            if (mine) {
                if (STATS) cnt.gates++;
                if ((kindw & G_SAME_BOX) || RT_GATE_ENTER(mn, mx, o, d, rg))
                    st.p[(sp++) * st.stride] = root;
                i = j + 1u;
            }
            break;
        }
        }
    }
    return i;
}
#endif

// ---- one closest-hit query over world->scene ("one ray") ----------------------------------------
// MODE 0: closest hit (`for obj in scene: hit |= obj->hit(ray, rec)`, main.cpp:1645-1648, 1390-1393,
//         1457-1460, 1548-1551) with rec.t starting at tmax.
// MODE 1: Material::shadowed (main.cpp:40-76): closest hit per world->scene entry with
//         rec.t = tmax; returns 1, 0 or 0.8 in `shadow`.
// DEFER (MODE 1 on scenes with DScene::shadow_defer, i.e. every Mesh sits in an all-opaque world->scene entry, where
//         ANY triangle hit closer than tmax decides the query): Mesh gates that pass are not traversed here; their BVH
//         roots are left on the stack and the count is returned (RT_DEFER_ROOTS + 1: more than one queue item
//         carries, the walk was abandoned), so the caller can queue the ray for the any-hit variant of k_bvh.
//         `shadow` is then the result as far as the primitives tested in place decide it.
template <int MODE, bool STATS, bool DEFER = false, int DEPTH = RT_STACK_DEPTH>
RT_FN uint32_t trace(const DScene& S, f3 o, f3 d, float tmax, Hit& best, float& shadow, Stack st, Counters& cnt)
{
    best.t = tmax; best.prim = H_MISS; best.u = 0; best.v = 0; best.rank = 0;
    const RayGate rg = make_gate(S.cull_pad, S.extent, o, d);
    bool shadowed = false, transparentOnly = true;
    if (S.debug & 2u) { shadow = 1.0f; return 0; }
    uint32_t cur_item = 0;
    uint32_t i = 0, sp = 0;
    const uint32_t n = S.num_gates;
    if (MODE == 0) i = lead_planes_hit(S, o, d, rg, best);
    else {
        // The shadow query keeps per-entry state, and each of the leading planes is a world->scene entry of its own: its closest
        // hit is the plane's hit if t < tmax (closer() against {tmax, none}).  Same wave-uniform loop as lead_planes_hit, with the
        // same cheap exclusion of axis planes whose approximate parameter is negative or beyond tmax by the margin
        // (plane_axis_cannot_win) before the IEEE division — for a light inside the room that is all six.  The deferring kernel
        // reads the planes from the kernel arguments; k_shadow, which traverses in place, reads the same numbers from the gate
        // array through scalar loads (indexed reads of the by-value scene made the compiler keep a copy of it in scratch memory
        // there, 520 bytes per lane).
        bool stop = false;
#if defined(__HIP_DEVICE_COMPILE__)
        typedef float rt_v4f __attribute__((ext_vector_type(4)));
        const RT_CONST_AS rt_v4f* G = (const RT_CONST_AS rt_v4f*)(uintptr_t)S.gates;
#endif
        for (uint32_t j = 0; j < S.num_lead_planes; ++j) {
            float px, py, pz, nx, ny, nz;
            uint32_t plane;
#if defined(__HIP_DEVICE_COMPILE__)
            if (!DEFER) {
                const rt_v4f g0 = G[3u * j], g1 = G[3u * j + 1u];
                px = g0.x; py = g0.y; pz = g0.z; nx = g1.x; ny = g1.y; nz = g1.z; plane = f2u(G[3u * j + 2u].x);
            } else
#endif
            { const float* pl = S.lead_planes[j]; px = pl[0]; py = pl[1]; pz = pl[2]; nx = pl[3]; ny = pl[4]; nz = pl[5]; plane = f2u(pl[7]); }
            float t = 0.0f;
            bool hit = false;
            if (!stop) switch ((S.lead_axis >> (4u * j)) & 15u) {   // wave-uniform
            case 0: hit = !plane_axis_cannot_win(px, o.x, rg.ix, tmax) && plane_axis_t(px, o.x, d.x, t); break;
            case 1: hit = !plane_axis_cannot_win(py, o.y, rg.iy, tmax) && plane_axis_t(py, o.y, d.y, t); break;
            case 2: hit = !plane_axis_cannot_win(pz, o.z, rg.iz, tmax) && plane_axis_t(pz, o.z, d.z, t); break;
            default: hit = plane_t(F3(px, py, pz), F3(nx, ny, nz), o, d, t); break;
            }
            if (hit && t < tmax) {
                shadowed = true;
                transparentOnly = transparentOnly && (S.materials[f2u(S.planes[(size_t)plane * 2].w)].kind == RT_MAT_REFRACTION);
                if (!transparentOnly) stop = true;   // an opaque entry in front of the light: decided
            }
        }
        i = stop ? n : S.num_lead_planes;   // (the first node behind the planes belongs to a later entry: the loop below starts it afresh)
    }
    while (i < n) {
        const GateNode g = load_gate(S, i);
        if (MODE == 1 && g.item != cur_item) {
            if (best.prim != H_MISS) {
                shadowed = true;
                uint32_t mat;
                uint32_t k = best.prim & H_MASK;
                if ((best.prim & ~H_MASK) == H_PLANE) mat = f2u(S.planes[k * 2].w);
                else if ((best.prim & ~H_MASK) == H_SPHERE) mat = f2u(S.spheres[k * 2 + 1].y);
                else mat = f2u(S.tris[(size_t)k * 3 + 1].w);
                transparentOnly = transparentOnly && (S.materials[mat].kind == RT_MAT_REFRACTION);
                if (!transparentOnly) break;
            }
            best.t = tmax; best.prim = H_MISS;
            cur_item = g.item;
        }
        const uint32_t kind = g.kind & G_KIND_MASK;
        if (kind == G_GATED || kind >= G_MESH) {
            // ONE copy of the box test for inner gates and Mesh gates (as in gate_step): the lanes of a wave that sit on the two
            // kinds take it together
            if (STATS) cnt.gates++;
            const bool pass = (g.kind & G_SAME_BOX) || RT_GATE_ENTER(g.mn, g.mx, o, d, rg);
            if (kind == G_GATED) {
                i = pass ? i + 1 : g.skip;
            } else {   // G_MESH: defer its triangle BVH to the shared drain loop
                bool next = true;
                if (pass) {
                    if (DEFER) {   // the caller queues the roots for k_bvh; one more than an item carries: the caller's overflow path
                        if (sp < RT_DEFER_ROOTS) st.p[sp * st.stride] = g.a;
                        if (++sp > RT_DEFER_ROOTS) { i = n; next = false; }
                    } else {
                        st.p[(sp++) * st.stride] = g.a;
                        if (sp >= RT_PENDING_MAX) {
                            const RayCull4 rc = make_cull4(S, o, d);
                            if (MODE == 1 && S.item_opaque[g.item]) {
                                if (bvh_drain<STATS, true, DEPTH>(S, o, d, rc, best, st, sp, cnt)) { shadowed = true; transparentOnly = false; i = n; next = false; }
                            } else {
                                bvh_drain<STATS, false, DEPTH>(S, o, d, rc, best, st, sp, cnt);
                            }
                        }
                    }
                }
                if (next) i++;
            }
        } else {
        switch (kind) {
        case G_PLANE: {
            float t;
            if (plane_t(F3(g.mn[0], g.mn[1], g.mn[2]), F3(g.mx[0], g.mx[1], g.mx[2]), o, d, t) && closer(t, g.rank, best)) {
                best.t = t; best.prim = H_PLANE | g.a; best.rank = g.rank;
            }
            break;
        }
        case G_SPHERE: {
            float t;
            if (sphere_t(F3(g.mn[0], g.mn[1], g.mn[2]), g.mx[0], o, d, t) && closer(t, g.rank, best)) {
                best.t = t; best.prim = H_SPHERE | g.a; best.rank = g.rank;
            }
            break;
        }
        case G_TRI: {
            const q4* T = S.tris + (size_t)g.a * 3;
            q4 a = T[0], b = T[1], e = T[2];
            if (STATS) cnt.tris++;
            float t, u, v;
            if (tri_t(xyz(a), xyz(b), xyz(e), o, d, t, u, v) && closer(t, g.rank, best)) {
                best.t = t; best.prim = H_TRI | g.a; best.u = u; best.v = v; best.rank = g.rank;
            }
            break;
        }
        default:   // G_GROUP
            break;
        }
        i++;
        }
        // end of the walk, or (shadow query) end of a world->scene entry: drain the deferred meshes
        if (!DEFER && sp > 0 && (i >= n || (MODE == 1 && load_gate(S, i).item != cur_item))) {
            const RayCull4 rc = make_cull4(S, o, d);
            if (MODE == 1 && S.item_opaque[cur_item]) {
                if (bvh_drain<STATS, true, DEPTH>(S, o, d, rc, best, st, sp, cnt)) { shadowed = true; transparentOnly = false; i = n; }
            } else {
                bvh_drain<STATS, false, DEPTH>(S, o, d, rc, best, st, sp, cnt);
            }
        }
    }
    if (MODE == 1) {
        if (i >= n && best.prim != H_MISS && !(shadowed && !transparentOnly)) {
            // finalise the last world->scene entry
            shadowed = true;
            uint32_t mat;
            uint32_t k = best.prim & H_MASK;
            if ((best.prim & ~H_MASK) == H_PLANE) mat = f2u(S.planes[k * 2].w);
            else if ((best.prim & ~H_MASK) == H_SPHERE) mat = f2u(S.spheres[k * 2 + 1].y);
            else mat = f2u(S.tris[(size_t)k * 3 + 1].w);
            transparentOnly = transparentOnly && (S.materials[mat].kind == RT_MAT_REFRACTION);
        }
        shadow = shadowed ? (transparentOnly ? 0.8f : 0.0f) : 1.0f;
    }
    return sp;
}

// ---- camera + pixel samples -----------------------------------------------------------------------
struct DCamera { f3 eye, U, V, W; float dist; };

// Camera::Camera (main.cpp:1407-1413)
RT_FN DCamera make_camera(const rt_camera& c)
{
    DCamera r;
    r.eye = F3(c.eye[0], c.eye[1], c.eye[2]);
    r.dist = c.frustum_dist;
    r.W = normalize(r.eye - F3(c.look_at[0], c.look_at[1], c.look_at[2]));
    r.V = normalize(cross(r.W, F3(c.up[0], c.up[1], c.up[2])));
    r.U = normalize(cross(r.W, r.V));
    return r;
}

// Camera::calculateRay (main.cpp:1420-1424)
RT_FN f3 camera_dir(const DCamera& c, float x, float y) { return normalize((c.V * x + c.U * y) - c.W * c.dist); }

// Sample point j of a pixel: the formulas of Regular/Jitter::generateSamples (main.cpp:986-987,
// 1010-1011) fed by the counter hash instead of rand(); strata are walked in order (the reference
// walks them in a shuffled order, which only permutes the terms of the pixel sum).
RT_FN void sample_point(int sampler, int num_samples, uint32_t seed, uint32_t pixel, int j, float& sx, float& sy)
{
    int n = (int)sqrtf((float)num_samples);
    if (n < 1) n = 1;
    int jj = j % (n * n), pp = jj / n, q = jj % n;
    if (sampler == RT_SAMPLER_REGULAR) { sx = (q + 0.5f) / n; sy = (pp + 0.5f) / n; return; }
    int r0 = (int)rt_rand31(seed, pixel, (uint32_t)j, 0), r1 = (int)rt_rand31(seed, pixel, (uint32_t)j, 1);
    if (sampler == RT_SAMPLER_RANDOM) {
        sx = (float)(r0 >> 7) * (1.0f / 16777216.0f);
        sy = (float)(r1 >> 7) * (1.0f / 16777216.0f);
        return;
    }
    float gridSize = 1.0f / n;
    sx = q * gridSize + ((float)r0 / (float)RT_RAND_MAX * gridSize);
    sy = pp * gridSize + ((float)r1 / (float)RT_RAND_MAX * gridSize);
}

// ---- surface attributes of a hit ------------------------------------------------------------------
struct Surface { f3 normal; int32_t material; float u, v; uint32_t prim_id; };

RT_FN Surface surface_of(const DScene& S, const Hit& h, f3 o, f3 d)
{
    Surface s; s.u = 0; s.v = 0;
    uint32_t k = h.prim & H_MASK, type = h.prim & ~H_MASK;
    if (type == H_PLANE) { // Plane::hit (main.cpp:930-938)
        q4 p0 = S.planes[k * 2], p1 = S.planes[k * 2 + 1];
        s.normal = xyz(p1); s.material = (int32_t)f2u(p0.w); s.prim_id = k;
    } else if (type == H_SPHERE) { // Sphere::hit (main.cpp:586-594): normal = (tmp + t*d)/r
        q4 s0 = S.spheres[k * 2], s1 = S.spheres[k * 2 + 1];
        s.normal = ((o - xyz(s0)) + d * h.t) / s0.w;
        s.material = (int32_t)f2u(s1.y); s.prim_id = S.prim_base_sphere + k;
    } else { // Triangle::hit (main.cpp:648-660)
        const q4* T = S.tris + (size_t)k * 3;
        q4 a = T[0], b = T[1], e = T[2];
        uint32_t gid = f2u(a.w);
        s.normal = normalize(cross(xyz(b), xyz(e)));
        s.material = (int32_t)f2u(b.w);
        q4 t0 = S.tri_attr[(size_t)gid * 2], t1 = S.tri_attr[(size_t)gid * 2 + 1];
        float bz = 1.0f - h.u - h.v;
        s.u = (t0.z * h.u + t1.x * h.v) + t0.x * bz;
        s.v = (t0.w * h.u + t1.y * h.v) + t0.y * bz;
        s.prim_id = S.prim_base_tri + gid;
    }
    return s;
}

// ---- lights -----------------------------------------------------------------------------------------
RT_FN f3 light_pos(const DLight& l) { return F3(l.v[0], l.v[1], l.v[2]); }
// getDirection / getSourcePoint (main.cpp:1320-1345); P = sr.ray(sr.t)
RT_FN f3 light_direction(const DLight& l, f3 P) { return l.kind == RT_LIGHT_POINT ? light_pos(l) - P : light_pos(l); }
RT_FN f3 light_source(const DLight& l, f3 P) { return l.kind == RT_LIGHT_POINT ? light_pos(l) : P + light_pos(l) * 1000000.0f; }
RT_FN f3 light_L(const DLight& l) { return F3(l.colour[0], l.colour[1], l.colour[2]) * l.radiance; } // Light::L (main.cpp:551-554)

// SpecularReflection::fn (main.cpp:1090-1105).  The reference evaluates ks*pow(vDotr, exp) in
// double (std::pow(double,double)); exp is an int, so the power is formed by repeated squaring
// in double and narrowed once.
RT_FN float specular_fn(float ks, int exp, f3 normal, f3 wo, f3 wi)
{
    f3 r = reflect(wi, normal);
    float vDotr = dot(r, -wo);
    if (!(vDotr > 0.0f)) return 0.0f;
    double base = (double)vDotr, acc = 1.0;
    unsigned e = exp < 0 ? (unsigned)(-exp) : (unsigned)exp;
    while (e) { if (e & 1u) acc *= base; base *= base; e >>= 1; }
    if (exp < 0) acc = 1.0 / acc;
    return (float)((double)ks * acc);
}

// ImageTexture::getColour (main.cpp:173-188); out-of-image reads are clamped to the image
RT_FN f3 texture_colour(const DScene& S, int tex, float u, float v)
{
    if (tex < 0 || (uint32_t)tex >= S.num_textures) return F3(0, 0, 0);
    DTexture T = S.textures[tex];
    if (!T.set) return F3(0, 0, 0);
    int x = (int)floorf(u * (float)T.width);
    int y = (int)floorf(v * (float)T.height);
    long long texel = (long long)x + (long long)T.height * y;
    long long last = (long long)T.width * T.height - 1;
    if (texel < 0) texel = 0;
    if (texel > last) texel = last;
    const uint8_t* px = S.texels + T.offset + texel * 4;
    return F3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
}

// A double constant that reaches its use through scalar registers set inside the loop (two s_mov_b32, `volatile` so that
// they are not hoisted): v_fma_f64 takes a scalar pair as one operand, so the constant costs no vector register.
// Host build: the same bits.
#if defined(__HIP_DEVICE_COMPILE__)
template <uint64_t BITS>
__device__ inline double rt_k64()
{
    uint32_t lo, hi;
    asm volatile("s_mov_b32 %0, %1" : "=s"(lo) : "i"((uint32_t)(BITS & 0xFFFFFFFFull)));
    asm volatile("s_mov_b32 %0, %1" : "=s"(hi) : "i"((uint32_t)(BITS >> 32)));
    union { uint32_t u[2]; double d; } c;
    c.u[0] = lo; c.u[1] = hi;
    return c.d;
}
#define RT_K64(bits) rt_k64<bits>()
// fma(a, b, K) with the constant as the instruction's scalar ADDEND: left to itself the compiler picks the two-address
// v_fmac_f64 here and first copies the constant into a vector pair (two v_mov_b32 per Horner step of sincos_f64)
template <uint64_t BITS>
__device__ inline double rt_fma_k64(double a, double b)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(rt_k64<BITS>()));
    return r;
}
#define RT_FMA_K64(a, b, bits) rt_fma_k64<bits>(a, b)
#else
inline double rt_k64_host(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
#define RT_K64(bits) rt_k64_host(bits)
#define RT_FMA_K64(a, b, bits) fma(a, b, rt_k64_host(bits))
#endif

// ---- ambient occlusion ray (Ambient::L, main.cpp:1361-1393; sampleHemisphere :4-19;
//      rotationMatToSpace :21-34 and the mat4*vec4 product) ---------------------------------------
// (double)r / (double)RT_RAND_MAX without an fp64 division: with y = RN(1/M), q = r*y,
// rem = fma(-q, M, r) (exact) the step fma(rem, y, q) is the correctly rounded quotient (Markstein);
// checked against the division for all 2^31 numerators (tests/test_emul.py samples it).
RT_FN double div_rand_max(int r)
{
    const double M = RT_K64(0x41DFFFFFFFC00000ULL /* 2147483647.0 */), y = RT_K64(0x3E00000000200000ULL /* 0x1.00000002p-31 */);
    double a = (double)r, q = a * y;
    double rem = fma(-q, M, a);
    return fma(rem, y, q);
}

// sin and cos of x in [0, 2*pi] (here: float phi widened to double) to ~1e-16, for rounding to
// float.  Cody-Waite reduction by pi/2 (k <= 4, so k*PIO2_HI is exact) + the fdlibm kernel
// polynomials.  Replaces two separate library calls (and their large-argument paths).
// The constants go through RT_K64 (above): as plain literals the compiler hoisted all 17 of them out of the AO
// kernels' stratum loop into 34 vector registers that stayed live through the whole gate walk.
RT_FN void sincos_f64(double x, double& sn, double& cs)
{
    const double k = rint(x * RT_K64(0x3FE45F306DC9C883ULL /* 6.36619772367581382433e-01 */));
    double r = fma(-k, RT_K64(0x3FF921FB54400000ULL /* 1.57079632673412561417e+00 */), x);
    r = fma(-k, RT_K64(0x3DD0B4611A626331ULL /* 6.07710050650619224932e-11 */), r);
    const double z = r * r;
    double ps = fma(z, RT_K64(0x3DE5D93A5ACFD57CULL /* 1.58969099521155010221e-10 */), RT_K64(0xBE5AE5E68A2B9CEBULL /* -2.50507602534068634195e-08 */));
    ps = RT_FMA_K64(z, ps, 0x3EC71DE357B1FE7DULL /* 2.75573137070700676789e-06 */);
    ps = RT_FMA_K64(z, ps, 0xBF2A01A019C161D5ULL /* -1.98412698298579493134e-04 */);
    ps = RT_FMA_K64(z, ps, 0x3F8111111110F8A6ULL /* 8.33333333332248946124e-03 */);
    ps = RT_FMA_K64(z, ps, 0xBFC5555555555549ULL /* -1.66666666666666324348e-01 */);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, RT_K64(0xBDA8FAE9BE8838D4ULL /* -1.13596475577881948265e-11 */), RT_K64(0x3E21EE9EBDB4B1C4ULL /* 2.08757232129817482790e-09 */));
    pc = RT_FMA_K64(z, pc, 0xBE927E4F809C52ADULL /* -2.75573143513906633035e-07 */);
    pc = RT_FMA_K64(z, pc, 0x3EFA01A019CB1590ULL /* 2.48015872894767294178e-05 */);
    pc = RT_FMA_K64(z, pc, 0xBF56C16C16C15177ULL /* -1.38888888888741095749e-03 */);
    pc = RT_FMA_K64(z, pc, 0x3FA555555555554CULL /* 4.16666666666666019037e-02 */);
    const double c = fma(z * z, pc, fma(z, -0.5, 1.0));   // -0.5 and 1.0 are inline operands of v_fma_f64
    const int n = (int)k & 3;
    sn = (n & 1) ? c : s; cs = (n & 1) ? s : c;
    if (n & 2) sn = -sn;
    if ((n + 1) & 2) cs = -cs;
}

// The part of the AO ray that is the same for all n*n rays of a hit (computed once, by k_prep):
// rotationMatToSpace's basis and the offset origin.
RT_FN void ao_frame(f3 P, f3 N, f3& o, f3& u, f3& w)
{
    w = normalize(N);
    u = normalize(cross(w, F3(0.053f, 1.0f, 0.0332f)));
    o = P + N * 0.01f;
}

// sqrtf(x) for x that is 0, NaN, at least 2^-96, or negative and normal — never a tiny positive number or a negative denormal
// (v_sqrt_f32 flushes denormal inputs: -denormal would give -0 where sqrtf gives NaN): the correctly rounded root as the
// compiler's own expansion computes it (v_sqrt_f32, then the one-ulp neighbours tried against x by fma), without that
// expansion's rescaling of inputs below 2^-96, which is 5 of its 16 instructions.  Both square roots of ao_dir qualify:
// 1 - sampleX^2 is 0 or a multiple of 2^-25, and the direction before normalisation has length ~1.
RT_FN float sqrt_not_tiny(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float s = __builtin_amdgcn_sqrtf(x);
    const uint32_t sb = __builtin_bit_cast(uint32_t, s);
    const float lo = __builtin_bit_cast(float, sb - 1u), hi = __builtin_bit_cast(float, sb + 1u);
    const float elo = __builtin_fmaf(-lo, s, x), ehi = __builtin_fmaf(-hi, s, x);
    float r = elo <= 0.0f ? lo : s;
    r = ehi > 0.0f ? hi : r;
    return __builtin_amdgcn_classf(x, 0x260 /* +-0, +inf */) ? x : r;
#else
    return sqrtf(x);
#endif
}

// Direction of AO ray k (stratum x = k / n, y = k % n) in the frame (u, w); v = cross(w, u), the same for all rays of the hit
RT_FN f3 ao_dir(const rt_ambient& A, f3 u, f3 w, f3 v, uint32_t rng_prefix, uint32_t ao_call, uint32_t k)
{
    unsigned n = (unsigned)A.occlusion_samples;
    uint32_t x = k / n, y = k % n;
    uint32_t ctr = 2u + ao_call * 2u * n * n + k * 2u;
    float grid = 1.0f / (float)n;
    int r0 = (int)rt_rand31_finish(rng_prefix, ctr), r1 = (int)rt_rand31_finish(rng_prefix, ctr + 1);
    float sampleX = (float)x * grid + (float)((div_rand_max(r0)) * (double)grid);
    float sampleY = (float)y * grid + (float)((div_rand_max(r1)) * (double)grid);
    float r = sqrt_not_tiny(1.0f - sampleX * sampleX);
    float phi = 2 * 3.14159265358979323846264338327950288f * sampleY;
    // glibc's cosf/sinf are (almost always) correctly rounded; get the same bits from fp64
    double sn, cs;
    sincos_f64((double)phi, sn, cs);
    f3 h = F3((float)cs * r, sampleX, (float)sn * r);
    f3 rr = (u * h.x + w * h.y) + (v * h.z + F3(0.0f, 0.0f, 0.0f));
    return rr * (1.0f / sqrt_not_tiny(dot(rr, rr)));   // normalize()
}
RT_FN f3 ao_dir(const rt_ambient& A, f3 u, f3 w, uint32_t rng_prefix, uint32_t ao_call, uint32_t k)
{
    return ao_dir(A, u, w, cross(w, u), rng_prefix, ao_call, k);
}

// Transparency::refraction (main.cpp:1566-1592)
RT_FN f3 refraction_dir(f3 d, f3 normal, float sr_ior, float& enter_ior)
{
    float cosi = clampf(dot(d, normal), -1.0f, 1.0f);
    f3 n = normal;
    if (cosi < 0) cosi = -cosi; else n = -normal;
    float eta = sr_ior / enter_ior;
    float k = 1 - eta * eta * (1 - cosi * cosi);
    if (k < 0) { enter_ior = sr_ior; return reflect(d, n); }
    return normalize(d * eta + n * (eta * cosi - sqrtf(k)));
}

} // namespace rt
