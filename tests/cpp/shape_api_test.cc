// The reference's four shape tests (tests/shape_test.cc:6-108) against this
// repository's host API: same transforms, same rays, same expected
// n_intersections.  Shape::getIntersections here runs the gfx950 kernel through
// the C ABI, so this binary needs a GPU.  The distance checks after each block
// are this repository's additions (the reference asserts counts only).
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

int main() { return RUN_ALL_TESTS(); }
