// The host API of include/pt/api.h on the GPU: the reference's 14 known answers through Shape::getIntersections (which
// runs the gfx950 kernel through the C ABI, so this binary needs a GPU), then nearest hit, tracePath and render against
// each other, and the cost and side effects of the one-ray calls.
#include <chrono>
#include <cmath>

#include "mini_gtest.h"
#include "../../include/pt/api.h"

// The reference's known answers (tests/shape_test.cc:6-108) as a table: shape, where it stands, the ray, how many
// intersections Shape::getIntersections must report — and, where the geometry makes them obvious, the distances (this
// repository's addition: the reference asserts counts only; NAN = not checked).
namespace {
enum Kind { SPHERE, PLANE, PLANE_OFFSET, CUBE, CYLINDER };
struct KnownAnswer {
    Kind kind;
    float at[3];
    float from[3];
    float towards[3];     // normalised before use
    int n;
    double first, second;
};
const double R3 = 1.7320508075688772;
const KnownAnswer KNOWN[] = {
    {SPHERE, {0, 0, 0}, {0, 0, 0}, {1, 1, 1}, 1, NAN, NAN},
    {SPHERE, {0, 10, 0}, {0, 0, 0}, {1, 1, 1}, 0, NAN, NAN},
    {SPHERE, {5, 5, 5}, {0, 0, 0}, {1, 1, 1}, 2, 5 * R3 - 0.5, 5 * R3 + 0.5},
    {PLANE, {0, 0, 0}, {0, 0, -5}, {1, 1, 1}, 1, 5 * R3, NAN},
    {PLANE_OFFSET, {0, 0, 0}, {0, 0, -5}, {1, 1, 1}, 0, NAN, NAN},          // bounds (0,0)-(2,2): the hit at (5,5) is outside
    {CUBE, {0, 0, 0}, {0, 0, 0}, {1, 1, 1}, 1, NAN, NAN},
    {CUBE, {0, 10, 0}, {0, 0, 0}, {1, 1, 1}, 0, NAN, NAN},
    {CUBE, {5, 5, 5}, {0, 0, 0}, {1, 1, 1}, 2, 4.5 * R3, 5.5 * R3},
    {CYLINDER, {0, 0, 0}, {0, 0, 0}, {0, 0, -1}, 1, NAN, NAN},
    {CYLINDER, {0, 10, 0}, {0, 0, 0}, {0, 0, -1}, 0, NAN, NAN},
    {CYLINDER, {0, 0, -5}, {0, 0, 0}, {0, 0, -1}, 2, 4.5, 5.5},
    {CYLINDER, {0, 0, 0}, {0, 0, 0}, {0, 1, 0}, 1, NAN, NAN},
    {CYLINDER, {0, 10, 0}, {0, 0, 0}, {0, 1, 0}, 2, 9.5, 10.5},
    {CYLINDER, {0, 0, -5}, {0, 0, 0}, {0, 1, 0}, 0, NAN, NAN},
};
}  // namespace

TEST(ShapeTest, KnownAnswersOfTheReference) {
    Material m;
    int rows = 0;
    for (const KnownAnswer& k : KNOWN) {
        Affine3f t = IDENTITY;
        t.translation() = vec3(k.at[0], k.at[1], k.at[2]);
        Shape* shape = nullptr;
        switch (k.kind) {
            case SPHERE: shape = new Sphere(m, t); break;
            case PLANE: shape = new Plane(m, t); break;
            case PLANE_OFFSET: shape = new Plane(m, t, 0, 0, 2, 2); break;
            case CUBE: shape = new Cube(m, t); break;
            case CYLINDER: shape = new Cylinder(m, t); break;
        }
        vec3 d(k.towards[0], k.towards[1], k.towards[2]);
        d.normalize();
        const RayPath p = shape->getIntersections(ray(vec3(k.from[0], k.from[1], k.from[2]), d));
        EXPECT_EQ(p.n_intersections, k.n);
        if (!std::isnan(k.first)) EXPECT_NEAR(p.first.distance, k.first, 1e-5);
        if (!std::isnan(k.second)) EXPECT_NEAR(p.second.distance, k.second, 1e-5);
        delete shape;
        rows++;
    }
    EXPECT_EQ(rows, 14);
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

TEST(ApiTest, StagingFollowsTheDeviceOfTheScene) {
    // The one-ray calls use device 0; a batch call with a scene on another device must stage on THAT device.  With
    // one GPU the second device is device 0 again and the test still walks the interleaved order.
    int n_dev = 0;
    checkHipErrors(hipGetDeviceCount(&n_dev));
    const int other = n_dev >= 2 ? 1 : 0;
    if (n_dev < 2) std::printf("           (one GPU visible: both scenes on device 0)\n");
    Material white; white.albedo = vec3(1.0f, 1.0f, 1.0f);
    Affine3f t = IDENTITY; t.translation() = vec3(0, 0, -3);
    Shape* scene[1] = {new Plane(white, t)};
    ray r(vec3(0, 0, 0), vec3(0, 0, -1));
    pt::DeviceScene far_scene(scene, 1, other);
    for (int round = 0; round < 3; round++) {
        IntersectionPoint a = getNearestIntersection(r, scene, 1);              // device 0, cached scene
        IntersectionPoint b;
        getNearestIntersection_batch(&r, 1, far_scene, scene, &b);               // the other device
        EXPECT_EQ(a.intersects, true); EXPECT_EQ(b.intersects, true);
        EXPECT_EQ(a.distance, 3.0f); EXPECT_EQ(b.distance, 3.0f);
        pt_rng s;
        vec3 c;
        tracePath_batch(far_scene, &r, &s, 1, &c);
        EXPECT_EQ(c.x(), 0.0f);
    }
    checkHipErrors(hipSetDevice(0));
    pt::release_cached();
    delete scene[0];
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
