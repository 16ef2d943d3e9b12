// raytracer_amd/host/Raytracer.hpp — C++17 host side of the MI355X render path.
//
// Mirrors the class surface of the reference's Raytracer.hpp (same names, constructor arguments,
// setters and defaults: reference Raytracer/Raytracer.hpp:75-641) so that a scene script written
// against the reference — World, Camera, Sphere/Triangle/Plane/MultiMesh, BVHAccel::addShape /
// generateBVH, the five materials, PointLight/Directional/Ambient, Regular/Jitter/Random,
// ThreadPool::startJob/waitDone, raytrace(beginY,endY,beginX,endX,camera,world), saveToFile — compiles
// against this header and renders on the GPU through the C ABI of include/rt_amd.h.
//
// Nothing here computes pixels: the classes only record the scene; raytrace() flattens the World
// once (re-flattened when the scene is edited) and calls rt_render for the requested rectangle.
// glm / atlas / stb are not needed: the few value types the script touches are defined below.
#pragma once

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <queue>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "rt_amd.h"

// ---- value types the scene script uses (atlas::math / atlas::core / atlas::utils / tinyobj names) ----
namespace atlas::math {
struct Vector2 { float x{0}, y{0}; Vector2() = default; Vector2(float a, float b) : x(a), y(b) {} };
struct Vector {
    union { float x; float r; };
    union { float y; float g; };
    union { float z; float b; };
    Vector() : x(0), y(0), z(0) {}
    explicit Vector(float s) : x(s), y(s), z(s) {}
    Vector(float a, float b_, float c) : x(a), y(b_), z(c) {}
};
inline Vector operator+(Vector a, Vector b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vector operator-(Vector a, Vector b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vector operator*(float s, Vector a) { return {a.x * s, a.y * s, a.z * s}; }
inline Vector operator*(Vector a, float s) { return {a.x * s, a.y * s, a.z * s}; }
using Point = Vector; using Normal = Vector; using Point2 = Vector2;
template <typename T> struct Ray { T o{}, d{}; T operator()(float t) const { return o + (t * d); } };
} // namespace atlas::math

namespace atlas::core {
template <typename T> class Timer {   // include/atlas/core/Timer.hpp:8-32
public:
    void start() { mBegin = std::chrono::steady_clock::now(); }
    T elapsed() const { return std::chrono::duration<T>(std::chrono::steady_clock::now() - mBegin).count(); }
private:
    std::chrono::steady_clock::time_point mBegin{std::chrono::steady_clock::now()};
};
} // namespace atlas::core

namespace tinyobj { struct material_t { std::string name; std::string diffuse_texname; }; }

namespace atlas::utils {   // include/atlas/utils/LoadObjFile.hpp:13-39
struct Vertex { math::Point position{}; math::Normal normal{}; math::Point2 texCoord{}; std::size_t index{}; std::size_t faceId{}; };
struct Shape {
    bool hasNormals{true}; bool hasTextureCoords{true};
    std::vector<Vertex> vertices{}; std::vector<std::size_t> indices{}; std::vector<int> materialIds; std::vector<unsigned int> smoothingGroupIds;
};
struct ObjMesh { std::vector<Shape> shapes; std::vector<tinyobj::material_t> materials; };
// Minimal OBJ/MTL reader honouring the contract the reference reads (positions, texcoords, one vertex
// per index, per-shape material ids, map_Kd): `v`, `vt`, `f a/b ...` (triangles), `o`/`g`, `usemtl`, `mtllib`.
std::optional<ObjMesh> loadObjMesh(std::string const& filename, std::string const& materialPath = {});
} // namespace atlas::utils

using Colour = atlas::math::Vector;
using Point = atlas::math::Point;
using Vector = atlas::math::Vector;
using Vector2 = atlas::math::Vector2;

void saveToFile(std::string const& filename, std::size_t width, std::size_t height, std::vector<Colour> const& image);

inline std::string modelRoot{"models/"};          // reference Raytracer.hpp:51 (run-time assignable; inline: one instance across translation units)
static const int maxBounceDepth = 3;               // :52
static const int lightAttenuationFactor = 500000;  // :53

class Material; class Light; class Shape; class Sampler; class Camera; class Textured;

struct World {   // Raytracer.hpp:75-87
    std::size_t width{0}, height{0};
    Colour background;
    std::shared_ptr<Sampler> sampler;
    std::vector<std::shared_ptr<Shape>> scene;
    std::vector<Colour> image;
    std::vector<std::shared_ptr<Light>> lights;
    std::shared_ptr<Light> ambient;
    std::shared_ptr<Camera> camera;
    int threadsAccessed = 0;
    // extensions (no counterpart in the reference): render controls of the GPU path
    int maxDepth = maxBounceDepth;
    unsigned seed = 1;
    void markDirty() { mRevision++; }   // call after editing the scene between renders
    unsigned long mRevision = 0;
};

class ImageTexture {   // Raytracer.hpp:104-116; decoded RGBA8 texels
public:
    explicit ImageTexture(std::string const& imageFilePath);         // binary PPM (P6) or 24/32-bit uncompressed TGA
    ImageTexture(int width, int height, std::vector<unsigned char> rgba) : mSet(true), mWidth(width), mHeight(height), mImage(std::move(rgba)) {}
    bool mSet{false}; int mWidth{0}, mHeight{0};
    std::vector<unsigned char> mImage;
};

class World;
// Raytracer.hpp:119-143; same interface, predicate-based completion.  raytrace() calls made
// from inside a pool job are not rendered one by one: the job records its rectangle and waitDone()
// renders everything recorded in as few GPU launches as possible (one, when the rectangles tile a
// rectangle, as the 1600 tile jobs of the reference's main() do).  world->image is complete when
// waitDone() returns, which is where the reference's main() reads it.
class ThreadPool {
public:
    explicit ThreadPool(unsigned int const& numThreads);
    void startJob(std::function<void()> job);
    void waitDone();
    ~ThreadPool();
    struct DeferredTile { int y0, y1, x0, x1; rt_camera camera; std::shared_ptr<World> world; };
    void defer(DeferredTile t);          // used by raytrace()
    static ThreadPool* current();        // the pool whose job the calling thread is running, or nullptr
    unsigned launches() const { return mLaunches; }   // GPU launches made by waitDone() so far
private:
    void loopFunction();
    void flushDeferred();
    std::mutex mDeferredMutex;
    std::vector<DeferredTile> mDeferred;
    unsigned mLaunches{0};
    bool mTerminatePool{false};
    int mPending{0};
    std::vector<std::thread> mThreads;
    std::queue<std::function<void()>> mQueue;
    std::mutex mQueueMutex;
    std::condition_variable mCondition, mComplCondition;
};

class Camera {   // Raytracer.hpp:147-159
public:
    Camera(Point position, Point lookAt, Vector up, float frustrumDist) : mEye(position), mLookAt(lookAt), mUp(up), mFrustrumDist(frustrumDist) {}
    Point getEye() const { return mEye; }
    rt_camera toC() const { return rt_camera{{mEye.x, mEye.y, mEye.z}, {mLookAt.x, mLookAt.y, mLookAt.z}, {mUp.x, mUp.y, mUp.z}, mFrustrumDist}; }
protected:
    Point mEye, mLookAt; Vector mUp; float mFrustrumDist;
};

class Sampler {   // Raytracer.hpp:161-183
public:
    Sampler(int numSamples, int numSets, int kind) : mNumSamples(numSamples), mNumSets(numSets), mKind(kind) {}
    virtual ~Sampler() = default;
    int getNumSamples() const { return mNumSamples; }
    int mNumSamples, mNumSets, mKind;
};
class Regular : public Sampler { public: Regular(int n, int s) : Sampler(n, s, RT_SAMPLER_REGULAR) {} };
class Jitter : public Sampler { public: Jitter(int n, int s) : Sampler(n, s, RT_SAMPLER_JITTER) {} };
class Random : public Sampler { public: Random(int n, int s) : Sampler(n, s, RT_SAMPLER_RANDOM) {} };

class Material {   // Raytracer.hpp:224-234 and 516-600
public:
    virtual ~Material() = default;
    bool mTransparent = false;
    rt_material c{RT_MAT_SPECULAR, {1, 1, 1}, 0.5f, 0.5f, 0.3f, 16, 1.7f, -1};
    std::shared_ptr<ImageTexture> texture;
};
class Matte : public Material {
public:
    Matte() { c.kind = RT_MAT_MATTE; }
    Matte(float kd, float ka, Colour color) : Matte() { setDiffuseReflection(kd); setAmbientReflection(ka); setDiffuseColour(color); }
    void setDiffuseReflection(float k) { c.kd = k; }
    void setAmbientReflection(float k) { c.ka = k; }
    void setDiffuseColour(Colour col) { c.colour[0] = col.x; c.colour[1] = col.y; c.colour[2] = col.z; }
};
class Specular : public Material {
public:
    Specular() { c.kind = RT_MAT_SPECULAR; }
    void setSpecularCoefficient(float k) { c.ks = k; }
    void setSpecularExp(int k) { c.spec_exp = k; }
    void setDiffuseReflection(float k) { c.kd = k; }
    void setAmbientReflection(float k) { c.ka = k; }
    void setDiffuseColour(Colour col) { c.colour[0] = col.x; c.colour[1] = col.y; c.colour[2] = col.z; }
};
class Mirror : public Material {
public:
    Mirror() { c.kind = RT_MAT_MIRROR; }
    void setAmbientColour(Colour col) { c.colour[0] = col.x; c.colour[1] = col.y; c.colour[2] = col.z; }
    void setAmbientReflection(float ka) { c.ka = ka; }
};
class Refraction : public Material {
public:
    Refraction() { c.kind = RT_MAT_REFRACTION; mTransparent = true; }
    void setIndexOfRefraction(float index) { c.ior = index; }
};
class Textured : public Material {
public:
    Textured(tinyobj::material_t const& material, std::string const& modelSubDirName = "")
    {
        c.kind = RT_MAT_TEXTURED;
        texture = std::make_shared<ImageTexture>(modelRoot + modelSubDirName + "/" + material.diffuse_texname);
    }
    explicit Textured(std::shared_ptr<ImageTexture> tex) { c.kind = RT_MAT_TEXTURED; texture = std::move(tex); }
    void setAmbientReflection(float ka) { c.ka = ka; }
    void setSpecularReflection(float ks) { c.ks = ks; }
    void setSpecularExp(int exp) { c.spec_exp = exp; }
    void setSpecularColour(Colour) {}
};

class Light {   // Raytracer.hpp:236-251, 602-641
public:
    virtual ~Light() = default;
    void scaleRadiance(float b) { mRadiance = b; }
    void setColour(Colour const& col) { mColour = col; }
    Colour mColour; float mRadiance{0};
    int mKind{-1}; Vector mV;     // -1: ambient
    int mOcclusionSamples{0}; float mOcclusionRayDist{1000};
};
class PointLight : public Light { public: PointLight() { mKind = RT_LIGHT_POINT; } explicit PointLight(Point const& p) { mKind = RT_LIGHT_POINT; mV = p; } };
class Directional : public Light {
public:
    Directional() { mKind = RT_LIGHT_DIRECTIONAL; }
    explicit Directional(Vector const& d) { mKind = RT_LIGHT_DIRECTIONAL; mV = d; }
    void setDirection(Vector const& d) { mV = d; }
};
class Ambient : public Light {
public:
    Ambient() { mOcclusionSamples = 4; mOcclusionRayDist = 1000; }   // Raytracer.hpp:639-640
    // extensions: the reference's two AO constants are private with no setter
    void setOcclusionSamples(unsigned n) { mOcclusionSamples = (int)n; }
    void setOcclusionRayDist(float d) { mOcclusionRayDist = d; }
};

class Shape {   // Raytracer.hpp:185-210
public:
    virtual ~Shape() = default;
    void setColour(Colour const& col) { mColour = col; }
    Colour getColour() const { return mColour; }
    void setMaterial(std::shared_ptr<Material> const& material) { mMaterial = material; }
    std::shared_ptr<Material> getMaterial() const { return mMaterial; }
    Colour mColour; std::shared_ptr<Material> mMaterial;
};
// The six support points of an axis-aligned box (reference main.cpp:190-226, 303-313): a point replaces the
// support point of a side only when it lies strictly beyond it, and getBoundingBoxPoints() lists them in the order
// x-, x+, y-, y+, z-, z+.  Scripts pass these lists to BVHAccel::addShape; the GPU path rebuilds every box from the
// geometry at rt_scene_commit, so they are bookkeeping for scripts that look at them.
class BoundingVolumeBox {
public:
    explicit BoundingVolumeBox(Vector const& startPoint) { for (auto& p : mP) p = startPoint; }
    void addVolumePoint(Point const& q)
    {
        if (q.x < mP[0].x) mP[0] = q;
        if (q.x > mP[1].x) mP[1] = q;
        if (q.y < mP[2].y) mP[2] = q;
        if (q.y > mP[3].y) mP[3] = q;
        if (q.z < mP[4].z) mP[4] = q;
        if (q.z > mP[5].z) mP[5] = q;
    }
    Point getCentroid() const { return {(mP[0].x + mP[1].x) / 2.0f, (mP[2].y + mP[3].y) / 2.0f, (mP[4].z + mP[5].z) / 2.0f}; }
    std::vector<Point> getBoundingBoxPoints() const { return {mP[0], mP[1], mP[2], mP[3], mP[4], mP[5]}; }
private:
    Point mP[6];
};

class Sphere : public Shape {
public:
    Sphere(Point center, float radius) : mCentre(center), mRadius(radius) {}
    std::vector<Point> getBoundingBoxPoints()   // main.cpp:571-576
    {
        BoundingVolumeBox b(Point{mCentre.x + mRadius, mCentre.y + mRadius, mCentre.z + mRadius});
        b.addVolumePoint(Point{mCentre.x - mRadius, mCentre.y - mRadius, mCentre.z - mRadius});
        return b.getBoundingBoxPoints();
    }
    Point mCentre; float mRadius;
};
class Triangle : public Shape {
public:
    Triangle(Point p0, Point p1, Point p2, Vector2 p0UV = {0, 0}, Vector2 p1UV = {0, 0}, Vector2 p2UV = {0, 0})
        : mV0(p0), mV1(p1), mV2(p2), mUV0(p0UV), mUV1(p1UV), mUV2(p2UV) {}
    std::vector<Point> getBoundingBoxPoints()   // main.cpp:705-711
    {
        BoundingVolumeBox b(mV0);
        b.addVolumePoint(mV1); b.addVolumePoint(mV2);
        return b.getBoundingBoxPoints();
    }
    Point mV0, mV1, mV2; Vector2 mUV0, mUV1, mUV2;
};
class Plane : public Shape {
public:
    Plane(Point point, Vector normal) : mPoint(point), mNormal(normal) {}
    Point mPoint; Vector mNormal;
};
class Mesh : public Shape {   // one OBJ shape, vertices translated by `offset` in fp32 (main.cpp:859-864)
public:
    Mesh(atlas::utils::Shape const& shape, unsigned int matIndex, std::vector<std::shared_ptr<Textured>> const& loadedMaterials,
         std::string modelSubDirName, Vector offset = {0, 0, 0});
    std::vector<Point> getBoundingBoxPoints() const;   // main.cpp:875-886: v0, then every triangle's v0, v1, v2
    std::vector<rt_triangle> mTriangles;   // material filled at flatten time
};
class MultiMesh : public Shape {
public:
    MultiMesh(atlas::utils::ObjMesh const& mesh, std::string modelSubDirName = "", Vector offset = {0, 0, 0});
    std::vector<Point> getBoundingBoxPoints() const;   // main.cpp:757-765, 798-801: over the support points of its meshes
    std::vector<Mesh> meshes;
};
class BVHAccel : public Shape {
public:
    void addShape(std::vector<Point>, std::shared_ptr<Shape> shape) { mLeaves.push_back(std::move(shape)); }
    void generateBVH() {}   // the hierarchy is rebuilt by rt_scene_commit with the reference's algorithm (from the shapes, not from the point lists)
    std::vector<std::shared_ptr<Shape>> mLeaves;
};

// raytrace(): renders rows [beginBlockY,endBlockY) x columns [beginBlockX,endBlockX) of world->image on the GPU
// (reference main.cpp:1620-1665; image index is r*width + c).
void raytrace(int beginBlockY, int endBlockY, int beginBlockX, int endBlockX, Camera const& camera, std::shared_ptr<World> const& world);
// One call for the whole frame (what a GPU caller wants instead of 1600 tile jobs); optional stats.
void render(Camera const& camera, std::shared_ptr<World> const& world, rt_stats* stats = nullptr);
