// The reference's four shape tests (tests/shape_test.cc:6-108) against this
// repository's host API: same transforms, same rays, same expected
// n_intersections.  Shape::getIntersections here runs the gfx950 kernel through
// the C ABI, so this binary needs a GPU.  The distance checks after each block
// are this repository's additions (the reference asserts counts only).
#include <chrono>

#include "mini_gtest.h"
#include "../../include/pt/api.h"

TEST(ShapeTest, SphereIntersectTest) {
    Affine3f t1 = IDENTITY, t2 = IDENTITY, t3 = IDENTITY;
    t2.translation() = vec3(0, 10, 0);
    t3.translation() = vec3(5, 5, 5);
    Material m;
    vec3 origin(0, 0, 0);
    vec3 direction(1, 1, 1);
    direction.normalize();
    ray ray1(origin, direction);
    Sphere sphere1(m, t1), sphere2(m, t2), sphere3(m, t3);
    RayPath p1 = sphere1.getIntersections(ray1);
    RayPath p2 = sphere2.getIntersections(ray1);
    RayPath p3 = sphere3.getIntersections(ray1);
    EXPECT_EQ(p1.n_intersections, 1);
    EXPECT_EQ(p2.n_intersections, 0);
    EXPECT_EQ(p3.n_intersections, 2);
    EXPECT_NEAR(p3.first.distance, 5 * std::sqrt(3.0) - 0.5, 1e-5);
    EXPECT_NEAR(p3.second.distance, 5 * std::sqrt(3.0) + 0.5, 1e-5);
}

TEST(ShapeTest, PlaneIntersectTest) {
    Affine3f t1 = IDENTITY;
    Material m;
    vec3 origin(0, 0, -5);
    vec3 direction(1, 1, 1);
    direction.normalize();
    ray ray1(origin, direction);
    Plane plane1(m, t1);
    Plane plane2(m, t1, 0, 0, 2, 2);
    RayPath p1 = plane1.getIntersections(ray1);
    RayPath p2 = plane2.getIntersections(ray1);
    EXPECT_EQ(p1.n_intersections, 1);
    EXPECT_EQ(p2.n_intersections, 0);
    EXPECT_NEAR(p1.first.distance, 5 * std::sqrt(3.0), 1e-5);
}

TEST(ShapeTest, CubeIntersectTest) {
    Affine3f t1 = IDENTITY, t2 = IDENTITY, t3 = IDENTITY;
    t2.translation() = vec3(0, 10, 0);
    t3.translation() = vec3(5, 5, 5);
    Material m;
    vec3 origin(0, 0, 0);
    vec3 direction(1, 1, 1);
    direction.normalize();
    ray ray1(origin, direction);
    Cube Cube1(m, t1), Cube2(m, t2), Cube3(m, t3);
    RayPath p1 = Cube1.getIntersections(ray1);
    RayPath p2 = Cube2.getIntersections(ray1);
    RayPath p3 = Cube3.getIntersections(ray1);
    EXPECT_EQ(p1.n_intersections, 1);
    EXPECT_EQ(p2.n_intersections, 0);
    EXPECT_EQ(p3.n_intersections, 2);
    EXPECT_NEAR(p3.first.distance, 4.5 * std::sqrt(3.0), 1e-5);
    EXPECT_NEAR(p3.second.distance, 5.5 * std::sqrt(3.0), 1e-5);
}

TEST(ShapeTest, CylinderIntersectTest) {
    Affine3f t1 = IDENTITY, t2 = IDENTITY, t3 = IDENTITY;
    t2.translation() = vec3(0, 10, 0);
    t3.translation() = vec3(0, 0, -5);
    Material m;
    ray ray1(vec3(0, 0, 0), vec3(0, 0, -1));
    ray ray2(vec3(0, 0, 0), vec3(0, 1, 0));
    Cylinder Cylinder1(m, t1), Cylinder2(m, t2), Cylinder3(m, t3);
    RayPath p1 = Cylinder1.getIntersections(ray1);
    RayPath p2 = Cylinder2.getIntersections(ray1);
    RayPath p3 = Cylinder3.getIntersections(ray1);
    EXPECT_EQ(p1.n_intersections, 1);
    EXPECT_EQ(p2.n_intersections, 0);
    EXPECT_EQ(p3.n_intersections, 2);
    EXPECT_EQ(p3.first.distance, 4.5f);
    EXPECT_EQ(p3.second.distance, 5.5f);
    p1 = Cylinder1.getIntersections(ray2);
    p2 = Cylinder2.getIntersections(ray2);
    p3 = Cylinder3.getIntersections(ray2);
    EXPECT_EQ(p1.n_intersections, 1);
    EXPECT_EQ(p2.n_intersections, 2);
    EXPECT_EQ(p3.n_intersections, 0);
}

// The remaining API of the path: nearest hit, tracePath and render agree with each other.
TEST(PathTest, NearestTraceRender) {
    Material light; light.emissive = vec3(2.0f, 3.0f, 4.0f);
    Affine3f t = IDENTITY;
    t.translation() = vec3(0, 0, -3);
    Shape* scene[1] = {new Plane(light, t)};
    ray r(vec3(0, 0, 0), vec3(0, 0, -1));
    IntersectionPoint ip = getNearestIntersection(r, scene, 1);
    EXPECT_EQ(ip.intersects, true);
    EXPECT_EQ(ip.distance, 3.0f);
    EXPECT_EQ(ip.normal.z(), 1.0f);
    EXPECT_EQ(ip.inside, false);
    // an emissive plane with albedo 0: the path sees the light once and dies
    pt_rng s;
    vec3 c = tracePath(scene, 1, r, &s);
    EXPECT_EQ(c.x(), 2.0f); EXPECT_EQ(c.y(), 3.0f); EXPECT_EQ(c.z(), 4.0f);
    // render: every pixel looks at the plane, so the image is the emission
    config_t config = {vec3(0, 0, 0), 8, 4, 6.0355339f, 5.0f};
    vec3* fb = nullptr;
    checkHipErrors(hipMallocManaged((void**)&fb, 32 * sizeof(vec3)));
    render_options opt; opt.n_samples = 3; opt.math_mode = PT_MATH_PARITY;
    render(fb, config, scene, 1, opt);
    for (int i = 0; i < 32; i++) { EXPECT_EQ(fb[i].x(), 2.0f); EXPECT_EQ(fb[i].z(), 4.0f); }
    checkHipErrors(hipFree(fb));
    delete scene[0];
}

// One-ray calls keep the uploaded scene and the staging buffers from call to call: 10 000 of them take well under a
// second, a mutated shape or material is seen at once, and tracePath leaves in `r` what the reference leaves there.
TEST(ApiTest, OneRayCallsAreCachedAndSeeMutations) {
    Material grey; grey.albedo = vec3(0.5f, 0.5f, 0.5f);
    Material light; light.emissive = vec3(1.0f, 1.0f, 1.0f);
    Affine3f back = IDENTITY; back.translation() = vec3(0, 0, -5);
    Affine3f ball = IDENTITY; ball.translation() = vec3(0, 0, -3);
    Shape* scene[2] = {new Plane(light, back), new Sphere(grey, ball)};
    ray r(vec3(0, 0, 0), vec3(0, 0, -1));
    IntersectionPoint first = getNearestIntersection(r, scene, 2);      // uploads
    EXPECT_EQ(first.distance, 2.5f);
    const auto t0 = std::chrono::steady_clock::now();
    int agree = 0;
    for (int k = 0; k < 10000; k++) {
        IntersectionPoint ip = getNearestIntersection(r, scene, 2);
        agree += ip.intersects && ip.distance == 2.5f;
    }
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("10000 getNearestIntersection calls: %.3f s\n", seconds);
    EXPECT_EQ(agree, 10000);
    EXPECT_EQ(seconds < 1.0, true);
    // a shape replaced in the array, and a material changed in place: both are seen by the next call
    delete scene[1];
    Affine3f nearer = IDENTITY; nearer.translation() = vec3(0, 0, -2);
    scene[1] = new Sphere(grey, nearer);
    EXPECT_EQ(getNearestIntersection(r, scene, 2).distance, 1.5f);
    scene[1]->material.albedo = vec3(0.25f, 0.5f, 0.75f);
    EXPECT_EQ(getNearestIntersection(r, scene, 2).material.albedo.x(), 0.25f);
    // one-shape queries go through the same cache
    RayPath rp;
    for (int k = 0; k < 2000; k++) rp = scene[1]->getIntersections(r);
    EXPECT_EQ(rp.n_intersections, 2);
    EXPECT_EQ(rp.first.distance, 1.5f);
    pt::release_cached();
    EXPECT_EQ(getNearestIntersection(r, scene, 2).distance, 1.5f);     // and come back after a release
    delete scene[0]; delete scene[1];
}

TEST(ApiTest, TracePathLeavesItsLastRayInR) {
    Material light; light.emissive = vec3(2.0f, 3.0f, 4.0f);              // albedo 0: the roulette ends the path on the light
    Material white; white.albedo = vec3(1.0f, 1.0f, 1.0f);               // albedo 1: the roulette never ends a path
    Affine3f t = IDENTITY; t.translation() = vec3(0, 0, -3);
    pt_rng s;
    {   // the path ends on its first segment: r stays the ray that was passed (bsdf.h:161 breaks before :165)
        Shape* scene[1] = {new Plane(light, t)};
        ray r(vec3(0, 0, 0), vec3(0, 0, -1));
        vec3 c = tracePath(scene, 1, r, &s);
        EXPECT_EQ(c.y(), 3.0f);
        EXPECT_EQ(r.origin().z(), 0.0f); EXPECT_EQ(r.direction().z(), -1.0f);
        delete scene[0];
    }
    {   // the first segment bounces, the second misses: r is the bounced ray, which starts on the plane
        Shape* scene[1] = {new Plane(white, t)};
        ray r(vec3(0, 0, 0), vec3(0, 0, -1));
        vec3 c = tracePath(scene, 1, r, &s);
        EXPECT_EQ(c.x(), 0.0f);
        EXPECT_EQ(r.origin().z(), -3.0f);
        EXPECT_EQ(r.direction().z() > 0.0f, true);                       // into the half space in front of the plane
        EXPECT_NEAR(r.direction().norm(), 1.0, 1e-6);
        // no bounces allowed: the only segment runs out of depth after assigning the new ray (bsdf.h:165), as above
        ray r0(vec3(0, 0, 0), vec3(0, 0, -1));
        tracePath(scene, 1, r0, &s, 0);
        EXPECT_EQ(r0.origin().z(), -3.0f);
        delete scene[0];
    }
}

TEST(ApiTest, UnusedAliasesOfVecMathExist) {                              // vec_math.h:15-17
    vec4 v(1, 2, 3, 4);
    matrix4 m = matrix4::Identity();
    Q4 q = Q4::Identity();
    EXPECT_EQ((m * v).w(), 4.0f);
    EXPECT_EQ(v.dot(v), 30.0f);
    EXPECT_EQ(q.w(), 1.0f);
}

int main() { return RUN_ALL_TESTS(); }
