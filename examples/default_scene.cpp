// examples/default_scene.cpp — config 1: the scene script of the reference's main()
// (scene data: reference Raytracer/main.cpp:1678-1801) written against
// raytracer_amd/host/Raytracer.hpp, i.e. rendered on the GPU.  Same flow as the reference: build the
// World, a BVHAccel over 12 shapes, a ThreadPool, 40x40 tile jobs calling raytrace(), waitDone(),
// saveToFile("raytrace.bmp").
//
//   default_scene <modelRoot> [out.bmp] [width] [height] [whole | bbox]
// <modelRoot>/teapot/teapot.obj (+ .mtl + texture) is the mesh; the reference's teapot.obj is not in
// its repository, tests/test_host_cpp.py writes the procedural stand-in used everywhere in this repo.
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "Raytracer.hpp"

int main(int argc, char** argv)
{
    using atlas::core::Timer;
    if (argc < 2) { std::cerr << "usage: default_scene <modelRoot> [out.bmp] [width] [height]\n"; return 2; }
    modelRoot = std::string(argv[1]) + "/";
    const std::string out = argc > 2 ? argv[2] : "raytrace.bmp";
    Timer<float> timer;
    timer.start();

    auto world = std::make_shared<World>();
    world->width = argc > 3 ? std::atoi(argv[3]) : 1000;
    world->height = argc > 4 ? std::atoi(argv[4]) : 1000;
    world->background = {0, 0, 0};
    world->sampler = std::make_shared<Jitter>(4, 83);

    auto bvhAccel = std::make_shared<BVHAccel>();
    auto teapotObjMesh = atlas::utils::loadObjMesh(modelRoot + "/teapot/teapot.obj");
    if (!teapotObjMesh) { std::cerr << "cannot read " << modelRoot << "/teapot/teapot.obj\n"; return 1; }

    auto teapotTexMaterial = std::make_shared<Textured>(teapotObjMesh.value().materials[0], "teapot");
    teapotTexMaterial->setAmbientReflection(0.7f);
    auto makeSpecular = [](Colour c) { auto m = std::make_shared<Specular>(); m->setDiffuseColour(c); return m; };
    auto specular = makeSpecular({0.9f, 0.1f, 0.4f}); specular->setDiffuseReflection(2.0f);
    auto planeSpecular1 = makeSpecular({1.0f, 0.1f, 0.1f}); planeSpecular1->setAmbientReflection(1.0f);
    auto planeSpecular2 = makeSpecular({0.1f, 0.1f, 1.0f}); planeSpecular2->setAmbientReflection(1.0f);
    auto planeSpecular3 = makeSpecular({0.85f, 0.85f, 0.85f}); planeSpecular3->setAmbientReflection(1.0f);
    auto planeSpecular4 = makeSpecular({0.1f, 1.0f, 0.1f}); planeSpecular4->setAmbientReflection(1.0f);
    auto mirror = std::make_shared<Mirror>();
    mirror->setAmbientColour({0.753f, 0.753f, 0.753f});
    mirror->setAmbientReflection(0.0f);
    auto refract = std::make_shared<Refraction>();

    auto triangle1 = std::make_shared<Triangle>(Point{-200, 0, -300}, Point{200, 0, -300}, Point{0, 400, -300}, Vector2{0, 0}, Vector2{1, 0}, Vector2{0.5f, 1});
    triangle1->setMaterial(mirror);
    auto triangle2 = std::make_shared<Triangle>(Point{200, 0, 0}, Point{200, 0, 400}, Point{200, 400, 200}, Vector2{0, 0}, Vector2{1, 0}, Vector2{0.5f, 1});
    triangle2->setMaterial(teapotTexMaterial);
    auto sphere = [](Point c, float r, std::shared_ptr<Material> m) { auto s = std::make_shared<Sphere>(c, r); s->setMaterial(m); return s; };
    auto sphere1 = sphere({-450, 0, -350}, 100, specular);
    auto sphere2 = sphere({-350, 0, 300}, 100, mirror);
    auto sphere3 = sphere({-250, 0, 80}, 100, mirror);
    auto sphere4 = sphere({-250, 180, 150}, 100, refract);
    auto sphere5 = sphere({-600, -50, -50}, 30, planeSpecular4);
    auto sphere6 = sphere({-620, -30, -90}, 50, planeSpecular1);
    auto sphere7 = sphere({-585, -20, -55}, 40, planeSpecular2);
    auto plane = [](Point p, Vector n, std::shared_ptr<Material> m) { auto s = std::make_shared<Plane>(p, n); s->setMaterial(m); return s; };
    auto plane1 = plane({0, -100, 0}, {0, 1, 0}, planeSpecular3);
    auto plane3 = plane({0, 550, 0}, {0, -1, 0}, planeSpecular3);
    auto plane4 = plane({-1050, 0, 0}, {1, 0, 0}, planeSpecular3);
    auto plane2 = plane({350, 0, 0}, {-1, 0, 0}, planeSpecular4);
    auto plane5 = plane({0, 0, -450}, {0, 0, 1}, planeSpecular1);
    auto plane6 = plane({0, 0, 450}, {0, 0, -1}, planeSpecular2);
    auto multiMesh1 = std::make_shared<MultiMesh>(teapotObjMesh.value(), "teapot", Vector{40, 0, -200});
    auto multiMesh2 = std::make_shared<MultiMesh>(teapotObjMesh.value(), "teapot", Vector{-430, -100, 400});
    auto multiMesh3 = std::make_shared<MultiMesh>(teapotObjMesh.value(), "teapot", Vector{-150, 500, 200});

    if (argc > 5 && std::string(argv[5]) == "bbox") {   // what the script hands to addShape, bit for bit (no GPU involved)
        auto dump = [](const char* name, std::vector<Point> const& pts) {
            std::printf("%s", name);
            for (auto const& p : pts) std::printf(" %a %a %a", (double)p.x, (double)p.y, (double)p.z);
            std::printf("\n");
        };
        for (auto& s : {sphere1, sphere2, sphere3, sphere4, sphere5, sphere6, sphere7}) dump("sphere", s->getBoundingBoxPoints());
        dump("triangle", triangle1->getBoundingBoxPoints()); dump("triangle", triangle2->getBoundingBoxPoints());
        for (auto& m : {multiMesh1, multiMesh2, multiMesh3}) dump("multimesh", m->getBoundingBoxPoints());
        return 0;
    }
    bvhAccel->addShape(triangle1->getBoundingBoxPoints(), triangle1);
    bvhAccel->addShape(triangle2->getBoundingBoxPoints(), triangle2);
    bvhAccel->addShape(multiMesh1->getBoundingBoxPoints(), multiMesh1);
    bvhAccel->addShape(multiMesh2->getBoundingBoxPoints(), multiMesh2);
    bvhAccel->addShape(multiMesh3->getBoundingBoxPoints(), multiMesh3);
    for (auto& s : {sphere1, sphere2, sphere3, sphere4, sphere5, sphere6, sphere7}) bvhAccel->addShape(s->getBoundingBoxPoints(), s);
    bvhAccel->generateBVH();
    for (auto& p : {plane1, plane2, plane3, plane4, plane5, plane6}) world->scene.push_back(p);
    world->scene.push_back(bvhAccel);

    world->ambient = std::make_shared<Ambient>();
    world->lights.push_back(std::make_shared<PointLight>(Point{-1000, 300, -400}));
    world->lights[0]->setColour({1, 1, 1});
    world->lights[0]->scaleRadiance(1.0f);
    world->ambient->setColour({1, 1, 1});
    world->ambient->scaleRadiance(0.5f);

    Camera camera{Point{-1000, 300, 250}, Point{0, 0, -100}, Vector{0, 1, 0}, 700.0f * world->width / 1000.0f};
    world->camera = std::make_shared<Camera>(camera);
    ThreadPool threadPool{std::thread::hardware_concurrency()};
    world->image = std::vector<Colour>(world->height * world->width, Colour{0, 0, 0});

    const bool whole = argc > 5 && std::string(argv[5]) == "whole";   // one render() call instead of the tile jobs
    const int gridN = whole ? 0 : 40;
    if (whole) render(camera, world);
    for (int y = 0; y < gridN; y++)
        for (int x = 0; x < gridN; x++) {
            const int cols = static_cast<int>(world->width / gridN), rows = static_cast<int>(world->height / gridN);
            const int beginY = y * rows, endY = (y == gridN - 1) ? (int)world->height : beginY + rows;
            const int beginX = x * cols, endX = (x == gridN - 1) ? (int)world->width : beginX + cols;
            threadPool.startJob([=, &camera, &world] { raytrace(beginY, endY, beginX, endX, camera, world); });
        }
    threadPool.waitDone();
    std::cout << "tile jobs: " << gridN * gridN << ", GPU launches: " << threadPool.launches() << std::endl;
    saveToFile(out, world->width, world->height, world->image);
    std::cout << "Running time is " << timer.elapsed() << std::endl;
    return 0;
}
