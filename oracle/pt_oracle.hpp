// CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A from-scratch CPU restatement of the reference path tracer's hot path
//   render -> tracePath -> getNearestIntersection -> Shape::getRayPath
// (reference: main.cu:30-69, bsdf.h:29-168, shapes.h:11-361, vec_math.h:5-30,
// ray.h:4-14, materials.h:7-39).  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may build, link or call this.  The product
// (libpt_hip.so) never includes or calls anything in this directory.
//
// PARITY STATUS: "parity unpinned" beyond the reference's own 14 known-answer
// n_intersections rows (tests/shape_test.cc:27-29,47-48,72-74,97-99,105-107).
// The reference cannot be compiled here: its Eigen submodule is empty, cuRAND is
// CUDA-only and shapes.h:61 binds a temporary to `ray&`.  The vector/matrix
// semantics below restate Eigen 3.4 fixed-size behaviour from its published
// algorithm (sum order x+(y+z), zero-safe normalized(), cofactor 3x3 inverse,
// AngleAxis::toRotationMatrix) and are anchored only by the reference's call
// sites.
//
// Stated deviations from the reference text (each is forced, none is optional):
//  (i)   float math uses the device overloads the reference ships with
//        (sqrtf/fabsf/expf/sinf/cosf, pow(float,int) = repeated multiply);
//  (ii)  RNG draw order inside one expression is fixed left-to-right
//        (main.cu:62, bsdf.h:139,150 leave it unspecified);
//  (iii) the two uninitialised reads are defined: Plane out of bounds or
//        parallel ray => n_intersections = 0 (shapes.h:117-132);
//  (iv)  cuRAND XORWOW (main.cu:58-60) is replaced by a counter-based generator
//        keyed (seed, pixel, sample, slot) with the same (0,1] output range;
//  (v)   `ray&` parameters are `const ray&`.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

namespace ptor {

// ----------------------------------------------------------------------------
// Event counters for the algorithmic-flop model (one count per event; the flop
// weight of each event lives in flops()).  Enabled with -DPT_ORACLE_COUNT.
// ----------------------------------------------------------------------------
struct Stats {
    uint64_t samples = 0;         // tracePath calls (= camera rays)
    uint64_t segments = 0;        // path segments traced (getNearestIntersection calls)
    uint64_t hits = 0;            // segments that hit something
    uint64_t tests[4] = {0, 0, 0, 0};  // getIntersections calls per type (plane, cube, cylinder, sphere)
    uint64_t sphere_roots = 0;    // sphere tests with discriminant >= 0
    uint64_t cyl_roots = 0;       // cylinder tests with discriminant > 0
    uint64_t emissive = 0;
    uint64_t beer = 0;
    uint64_t fresnel = 0;
    uint64_t fresnel_dense = 0;   // n1 > n2 branch taken
    uint64_t mode[4] = {0, 0, 0, 0};  // index 1..3
    void add(const Stats& o) {
        samples += o.samples; segments += o.segments; hits += o.hits;
        for (int i = 0; i < 4; i++) { tests[i] += o.tests[i]; mode[i] += o.mode[i]; }
        sphere_roots += o.sphere_roots; cyl_roots += o.cyl_roots;
        emissive += o.emissive; beer += o.beer; fresnel += o.fresnel; fresnel_dense += o.fresnel_dense;
    }
    // Flop weights: add/sub/mul/min/max/div/sqrt/rcp/sin/cos/exp = 1, compares,
    // selects and integer work = 0; minimal work to produce the reference result.
    double flops() const {
        const double XFORM = 33;
        double f = 28.0 * samples;
        f += (XFORM + 7) * tests[0] + (XFORM + 25) * tests[1] + (XFORM + 49) * tests[2] + (XFORM + 21) * tests[3];
        f += 6.0 * sphere_roots + 18.0 * cyl_roots;
        f += (53.0 + 20.0) * hits + 6.0 * emissive + 9.0 * beer + 22.0 * fresnel + 7.0 * fresnel_dense;
        f += 66.0 * mode[1] + 75.0 * mode[2] + 23.0 * mode[3];
        return f;
    }
};

#ifdef PT_ORACLE_COUNT
extern thread_local Stats* g_stats;
#define PT_COUNT(expr) do { if (g_stats) { g_stats->expr; } } while (0)
#else
#define PT_COUNT(expr) do { } while (0)
#endif

// ----------------------------------------------------------------------------
// vec_math.h:5-30 — constants, vec3, lerp, reflect, refract.
// ----------------------------------------------------------------------------
constexpr double PI = 3.14159265358979323846;      // vec_math.h:5
constexpr double SMALL_NUMBER = 0.00001;           // vec_math.h:7 (a double)
inline double DEG_TO_RAD(double x) { return (x * PI) / 180; }  // vec_math.h:6

// Stand-in for Eigen::Vector3<float> (vec_math.h:14).  Reductions sum as
// x + (y + z); normalized() divides each component by sqrt(squaredNorm) and
// returns the vector unchanged when the squared norm is not > 0.
struct vec3 {
    float v[3];
    vec3() : v{0.0f, 0.0f, 0.0f} {}
    vec3(float x, float y, float z) : v{x, y, z} {}
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float z() const { return v[2]; }
    float operator[](int i) const { return v[i]; }
    float& operator[](int i) { return v[i]; }
    float dot(const vec3& o) const { return v[0] * o.v[0] + (v[1] * o.v[1] + v[2] * o.v[2]); }
    float squaredNorm() const { return v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]); }
    float norm() const { return sqrtf(squaredNorm()); }
    vec3 normalized() const {
        float n = squaredNorm();
        if (n > 0.0f) { float s = sqrtf(n); return vec3(v[0] / s, v[1] / s, v[2] / s); }
        return *this;
    }
    vec3 cwiseProduct(const vec3& o) const { return vec3(v[0] * o.v[0], v[1] * o.v[1], v[2] * o.v[2]); }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]); }
inline vec3 operator-(const vec3& a) { return vec3(-a.v[0], -a.v[1], -a.v[2]); }
inline vec3 operator*(const vec3& a, float s) { return vec3(a.v[0] * s, a.v[1] * s, a.v[2] * s); }
inline vec3 operator*(float s, const vec3& a) { return vec3(s * a.v[0], s * a.v[1], s * a.v[2]); }
inline vec3 operator/(const vec3& a, float s) { return vec3(a.v[0] / s, a.v[1] / s, a.v[2] / s); }

template <typename T> inline T lerp(T v0, T v1, float t) { return (1 - t) * v0 + t * v1; }  // vec_math.h:10-12

inline vec3 reflect(vec3 original, vec3 normal) {  // vec_math.h:19-21
    return original - 2 * (original.dot(normal) * normal);
}

inline vec3 refract(vec3 I, vec3 N, float eta) {   // vec_math.h:23-30
    // k is a genuine double chain: the float products are promoted by the 1.0 literals.
    double k = 1.0 - (double)(eta * eta) * (1.0 - (double)((N.dot(I)) * (N.dot(I))));
    float kf = (float)k;
    if (kf < 0.0) {
        return vec3(0, 0, 0);
    } else {
        return eta * I - (eta * (N.dot(I)) + sqrtf(kf)) * N;
    }
}

// Stand-in for Eigen::Affine3f: rows of [L | t], last row implied (0,0,0,1).
struct Affine3f {
    float m[3][4];
    static Affine3f Identity() {
        Affine3f a;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) a.m[i][j] = (i == j) ? 1.0f : 0.0f;
        return a;
    }
    void setTranslation(float x, float y, float z) { m[0][3] = x; m[1][3] = y; m[2][3] = z; }
    vec3 translation() const { return vec3(m[0][3], m[1][3], m[2][3]); }
    // linear() * v, each row reduced as x + (y + z).
    vec3 linearMul(const vec3& p) const {
        return vec3(m[0][0] * p.v[0] + (m[0][1] * p.v[1] + m[0][2] * p.v[2]),
                    m[1][0] * p.v[0] + (m[1][1] * p.v[1] + m[1][2] * p.v[2]),
                    m[2][0] * p.v[0] + (m[2][1] * p.v[1] + m[2][2] * p.v[2]));
    }
    // Transform * point = linear()*p + translation().
    vec3 operator*(const vec3& p) const { return linearMul(p) + translation(); }
    // scale(s): post-multiplies the linear block, translation untouched (main.cu:119,124,129,134).
    void scale(float s) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] *= s; }
    void scale(const vec3& s) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] *= s.v[j]; }
    // AngleAxisf(angle, axis).toRotationMatrix() assigned to linear() (main.cu:78-107).
    void setRotation(float angle, const vec3& axis) {
        float s = sinf(angle), c = cosf(angle);
        vec3 sin_axis = s * axis;
        vec3 cos1_axis = (1.0f - c) * axis;
        float tmp;
        tmp = cos1_axis.x() * axis.y(); m[0][1] = tmp - sin_axis.z(); m[1][0] = tmp + sin_axis.z();
        tmp = cos1_axis.x() * axis.z(); m[0][2] = tmp + sin_axis.y(); m[2][0] = tmp - sin_axis.y();
        tmp = cos1_axis.y() * axis.z(); m[1][2] = tmp - sin_axis.x(); m[2][1] = tmp + sin_axis.x();
        m[0][0] = cos1_axis.x() * axis.x() + c;
        m[1][1] = cos1_axis.y() * axis.y() + c;
        m[2][2] = cos1_axis.z() * axis.z() + c;
    }
    // inverse() in Affine mode: 3x3 cofactor inverse of the linear block, then
    // translation = (-Linv) * t (shapes.h:37).
    Affine3f inverse() const {
        auto cof = [&](int i, int j) {
            int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
        };
        float c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
        float det = c00 * m[0][0] + (c10 * m[1][0] + c20 * m[2][0]);
        float invdet = 1.0f / det;
        Affine3f r;
        for (int row = 0; row < 3; row++)
            for (int col = 0; col < 3; col++)
                r.m[row][col] = cof(col, row) * invdet;
        for (int row = 0; row < 3; row++)
            r.m[row][3] = (-r.m[row][0]) * m[0][3] + ((-r.m[row][1]) * m[1][3] + (-r.m[row][2]) * m[2][3]);
        return r;
    }
};

// ----------------------------------------------------------------------------
// ray.h:4-14
// ----------------------------------------------------------------------------
class ray {
    vec3 o, d;
  public:
    ray() {}
    ray(vec3 origin, vec3 direction) : o(origin), d(direction) {}
    vec3 origin() const { return o; }
    vec3 direction() const { return d; }
    vec3 at(float t) const { return o + d * t; }
};

// ----------------------------------------------------------------------------
// materials.h:7-39
// ----------------------------------------------------------------------------
class Material {
  public:
    vec3 albedo = vec3(0.0f, 0.0f, 0.0f);
    vec3 emissive = vec3(0.0f, 0.0f, 0.0f);
    float f0 = 0.0f;
    float specularRoughness = 0.0f;
    vec3 specularColor = vec3(0.0f, 0.0f, 0.0f);
    float IOR = 1.0f;
    float transparency = 0.0f;
    float refractionRoughness = 0.0f;
    vec3 refractionColor = vec3(0.0f, 0.0f, 0.0f);
    int bsdfType = 0;
    int id = -1;   // oracle-only tag so tests can report which material won
};

// ----------------------------------------------------------------------------
// shapes.h:11-25
// ----------------------------------------------------------------------------
struct IntersectionPoint {
    bool intersects = false;
    float distance = 0.0f;      // reference leaves this uninitialised; 0 is the defined stand-in
    Material material;
    vec3 position;
    vec3 normal;
    bool inside = false;
    int prim = -1;              // oracle-only: index of the winning primitive
};

struct RayPath {
    int n_intersections = 0;    // reference leaves this uninitialised in Plane (deviation iii)
    IntersectionPoint first;
    IntersectionPoint second;
};

enum ShapeType { PT_PLANE = 0, PT_CUBE = 1, PT_CYLINDER = 2, PT_SPHERE = 3 };

// ----------------------------------------------------------------------------
// shapes.h:28-76 — Shape base
// ----------------------------------------------------------------------------
class Shape {
  protected:
    Affine3f transform;
    Affine3f transform_inv;
    vec3 wOrigin;
    explicit Shape(Affine3f t) {                        // shapes.h:35-39
        transform = t;
        transform_inv = t.inverse();
        wOrigin = transform * vec3(0, 0, 0);
    }
  public:
    Material material;
    virtual ~Shape() {}
    virtual int type() const = 0;
    ray transformRay(const ray& r) const {              // shapes.h:42-44
        return ray(transformPointLocal(r.origin()), transformVectorLocal(r.direction()));
    }
    vec3 transformPointLocal(vec3 v) const { return transform_inv * v; }            // :46
    vec3 transformVectorLocal(vec3 v) const { return transform_inv.linearMul(v); }  // :48
    vec3 transformPointWorld(vec3 v) const { return transform * v; }                // :50
    vec3 transformVectorWorld(vec3 v) const { return transform.linearMul(v); }      // :52
    vec3 transformNormalWorld(vec3 n) const { return transform.linearMul(n); }      // :56

    virtual RayPath getRayPath(const ray& r) const = 0;                             // :58

    RayPath getIntersections(const ray& r) const {      // shapes.h:60-75
        PT_COUNT(tests[type()]++);
        RayPath res = getRayPath(transformRay(r));
        if (res.n_intersections == 1 && (double)res.first.distance <= SMALL_NUMBER) {
            res.n_intersections = 0;
        } else if (res.n_intersections == 2) {
            if ((double)res.first.distance <= SMALL_NUMBER && (double)res.second.distance <= SMALL_NUMBER) {
                res.n_intersections = 0;
            } else if ((double)res.first.distance <= SMALL_NUMBER) {
                res.n_intersections = 1;
                res.first = res.second;
            } else if ((double)res.second.distance <= SMALL_NUMBER) {
                res.n_intersections = 1;
            }
        }
        return res;
    }
};

// shapes.h:79-100
inline IntersectionPoint getNearestIntersection(const ray& r, Shape* const* scene, int n_objects) {
    PT_COUNT(segments++);
    IntersectionPoint min_p;
    min_p.intersects = false;
    for (int k = 0; k < n_objects; k++) {
        const Shape* obj = scene[k];
        RayPath p = obj->getIntersections(r);
        IntersectionPoint point;
        if (p.n_intersections == 1) {
            point = p.first;
        } else if (p.n_intersections == 2) {
            point = (p.second.distance < p.first.distance) ? p.second : p.first;
        } else {
            point.intersects = false;
        }
        if (point.intersects && (!min_p.intersects || point.distance < min_p.distance)) {
            min_p = point;
            min_p.prim = k;
            min_p.inside = min_p.normal.dot(r.direction()) > 0;
            if (min_p.inside) { min_p.normal = -1.0f * min_p.normal; }
        }
    }
    return min_p;
}

// ----------------------------------------------------------------------------
// shapes.h:103-134 — Plane (local z = 0, normal +z, rectangular bounds)
// ----------------------------------------------------------------------------
class Plane : public Shape {
    float xMin, yMin, xMax, yMax;
  public:
    Plane(Material material, Affine3f transform,
          float xMin = std::numeric_limits<float>::lowest(), float yMin = std::numeric_limits<float>::lowest(),
          float xMax = std::numeric_limits<float>::max(), float yMax = std::numeric_limits<float>::max())
        : Shape(transform), xMin(xMin), yMin(yMin), xMax(xMax), yMax(yMax) { this->material = material; }
    int type() const override { return PT_PLANE; }
    RayPath getRayPath(const ray& r) const override {   // shapes.h:116-133
        RayPath res;
        res.n_intersections = 0;
        if (r.direction().dot(vec3(0, 0, 1)) != 0) {
            float d = ((-1.0f * r.origin()).dot(vec3(0, 0, 1))) / (r.direction().dot(vec3(0, 0, 1)));
            vec3 pos = r.at(d);
            if (xMin < pos.x() && yMin < pos.y() && xMax > pos.x() && yMax > pos.y()) {
                res.n_intersections = 1;
                res.first.distance = d;
                res.first.intersects = true;
                res.first.normal = (transformPointWorld(vec3(0, 0, 1).dot(r.direction()) > 0 ? vec3(0, 0, -1) : vec3(0, 0, 1)) - wOrigin).normalized();
                res.first.position = transformPointWorld(pos);
                res.first.material = material;
            }
        }
        return res;
    }
};

// ----------------------------------------------------------------------------
// shapes.h:137-221 — Cube (local AABB +-0.5)
// ----------------------------------------------------------------------------
class Cube : public Shape {
    float xMin, xMax, yMin, yMax, zMin, zMax;
  public:
    Cube(Material material, Affine3f transform) : Shape(transform) {
        this->material = material;
        vec3 cornerLeft(-0.5f, -0.5f, -0.5f), cornerRight(0.5f, 0.5f, 0.5f);
        xMin = fminf(cornerLeft.x(), cornerRight.x()); yMin = fminf(cornerLeft.y(), cornerRight.y());
        zMin = fminf(cornerLeft.z(), cornerRight.z()); xMax = fmaxf(cornerLeft.x(), cornerRight.x());
        yMax = fmaxf(cornerLeft.y(), cornerRight.y()); zMax = fmaxf(cornerLeft.z(), cornerRight.z());
    }
    int type() const override { return PT_CUBE; }
    vec3 getNormal(vec3 position) const {               // shapes.h:158-169
        float t = fmaxf(fabsf(position.x()), fmaxf(fabsf(position.y()), fabsf(position.z())));
        vec3 res;
        if (fabsf(position.x()) == t) {
            res = vec3(position.x(), 0, 0);
        } else if (fabsf(position.y()) == t) {
            res = vec3(0, position.y(), 0);
        } else {
            res = vec3(0, 0, position.z());
        }
        return (transformPointWorld(res) - wOrigin).normalized();
    }
    RayPath getRayPath(const ray& r) const override {   // shapes.h:171-220
        float x = 1.0f / r.direction().x();
        float y = 1.0f / r.direction().y();
        float z = 1.0f / r.direction().z();
        float t1 = (xMin - r.origin().x()) * x;
        float t2 = (xMax - r.origin().x()) * x;
        float t3 = (yMin - r.origin().y()) * y;
        float t4 = (yMax - r.origin().y()) * y;
        float t5 = (zMin - r.origin().z()) * z;
        float t6 = (zMax - r.origin().z()) * z;
        float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
        float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
        RayPath res;
        res.n_intersections = 0;
        if (tmax < 0) return res;
        if (tmin > tmax) return res;
        vec3 pos = r.at(tmin);
        res.n_intersections = 2;
        res.first.distance = tmin;
        res.first.intersects = true;
        res.first.material = material;
        res.first.position = transformPointWorld(pos);
        res.first.normal = getNormal(pos);
        pos = r.at(tmax);
        res.second.distance = tmax;
        res.second.intersects = true;
        res.second.material = material;
        res.second.position = transformPointWorld(pos);
        res.second.normal = getNormal(pos);
        return res;
    }
};

// ----------------------------------------------------------------------------
// shapes.h:225-319 — Cylinder (r = 0.5, z in [-0.5, 0.5])
// ----------------------------------------------------------------------------
class Cylinder : public Shape {
    float zmin = -0.5f, zmax = 0.5f;
  public:
    Cylinder(Material material, Affine3f transform) : Shape(transform) { this->material = material; }
    int type() const override { return PT_CYLINDER; }
    IntersectionPoint getIntersectionDisc(const ray& r, vec3 disc_normal) const {   // shapes.h:231-255
        IntersectionPoint res;
        res.intersects = false;
        vec3 p = disc_normal * 0.5f;
        if (r.direction().dot(disc_normal) == 0) return res;
        float t = (p - r.origin()).dot(disc_normal) / (r.direction().dot(disc_normal));
        vec3 pos = r.at(t);
        if ((pos - p).norm() > 0.5f) return res;
        res.intersects = true;
        res.position = transformPointWorld(pos);
        res.material = material;
        res.normal = (transformPointWorld(disc_normal) - wOrigin).normalized();
        res.distance = t;
        return res;
    }
    RayPath getRayPath(const ray& r) const override {   // shapes.h:257-318
        float dx = r.direction().x(), dy = r.direction().y();
        float ox = r.origin().x(), oy = r.origin().y();
        float a = dx * dx + dy * dy;
        float b = 2 * (dx * ox + dy * oy);
        float c = (float)((double)(ox * ox + oy * oy) - 0.25);
        float discriminant = b * b - 4 * a * c;
        RayPath res;
        res.n_intersections = 0;
        IntersectionPoint disc1 = getIntersectionDisc(r, vec3(0, 0, 1));
        IntersectionPoint disc2 = getIntersectionDisc(r, vec3(0, 0, -1));
        IntersectionPoint cylinder1, cylinder2;
        if (discriminant > 0) {
            PT_COUNT(cyl_roots++);
            float d = (float)((double)(-b - sqrtf(discriminant)) / (2.0 * (double)a));
            vec3 pos = r.at(d);
            cylinder1.intersects = (zmin < pos.z() && pos.z() < zmax);
            cylinder1.position = transformPointWorld(pos);
            cylinder1.material = material;
            cylinder1.distance = d;
            cylinder1.normal = (transformPointWorld(vec3(pos[0], pos[1], 0)) - wOrigin).normalized();
            d = (float)((double)(-b + sqrtf(discriminant)) / (2.0 * (double)a));
            pos = r.at(d);
            cylinder2.intersects = (zmin < pos.z() && pos.z() < zmax);
            cylinder2.position = transformPointWorld(pos);
            cylinder2.material = material;
            cylinder2.distance = d;
            cylinder2.normal = (transformPointWorld(vec3(pos[0], pos[1], 0)) - wOrigin).normalized();
        }
        if (disc1.intersects && disc2.intersects) {
            res.n_intersections = 2; res.first = disc1; res.second = disc2;
        } else if (disc1.intersects && discriminant > 0) {
            res.n_intersections = 2; res.first = disc1; res.second = cylinder1.intersects ? cylinder1 : cylinder2;
        } else if (disc2.intersects && discriminant > 0) {
            res.n_intersections = 2; res.first = disc2; res.second = cylinder1.intersects ? cylinder1 : cylinder2;
        } else if (discriminant > 0 && cylinder1.intersects && cylinder2.intersects) {
            res.n_intersections = 2; res.first = cylinder1; res.second = cylinder2;
        } else {
            res.n_intersections = 0;
        }
        return res;
    }
};

// ----------------------------------------------------------------------------
// shapes.h:323-361 — Sphere (r = 0.5)
// ----------------------------------------------------------------------------
class Sphere : public Shape {
  public:
    Sphere(Material material, Affine3f transform) : Shape(transform) { this->material = material; }
    int type() const override { return PT_SPHERE; }
    RayPath getRayPath(const ray& r) const override {   // shapes.h:327-360
        float a = r.direction().dot(r.direction());
        float b = 2.0f * (r.direction().dot(r.origin()));
        float c = (float)((double)(r.origin()).dot(r.origin()) - 0.25);
        float discriminant = b * b - 4 * a * c;
        RayPath res;
        if (discriminant < 0) {
            res.n_intersections = 0;
        } else {
            res.n_intersections = 2;
        }
        if (res.n_intersections == 2) {
            PT_COUNT(sphere_roots++);
            float d = (float)((double)(-b - sqrtf(discriminant)) / (2.0 * (double)a));
            vec3 pos = r.at(d);
            res.first.intersects = true;
            res.first.position = transformPointWorld(pos);
            res.first.material = material;
            res.first.distance = d;
            res.first.normal = (transformPointWorld(pos) - wOrigin).normalized();
            d = (float)((double)(-b + sqrtf(discriminant)) / (2.0 * (double)a));
            pos = r.at(d);
            res.second.intersects = true;
            res.second.position = transformPointWorld(pos);
            res.second.material = material;
            res.second.distance = d;
            res.second.normal = (transformPointWorld(pos) - wOrigin).normalized();
        }
        return res;
    }
};

// ----------------------------------------------------------------------------
// Counter-based RNG (replaces curandState, main.cu:58-60; deviation iv).
//   H(x)      : 32-bit multiply-xorshift integer hash
//   K_px      = H(pixel_index ^ H(seed + 0x9E3779B9))
//   K_smp     = H(K_px + 0x9E3779B9 * (sample + 1))
//   bits(k)   = H(K_smp ^ (0x85EBCA6B * (k + 1)))        k = draw slot
// Slot map: 0,1 = pixel jitter x,y; segment s: 2+6s+{0 mode, 1,2 diffuse u1,u2,
// 3,4 refraction-roughness u1,u2, 5 Russian roulette}.
// A draw that decides a branch (mode, roulette) uses all 32 bits of its slot:
//   uniform   = (float)bits * 2^-32 + 2^-33   in float arithmetic  -> (0, 1]
// The two coordinates of a point (jitter x,y; u1,u2 of a direction) share the
// bits of the FIRST slot of their pair, 16 bits each, cell centres:
//   u(first)     = (bits(first) >> 16)     * 2^-16 + 2^-17
//   u(first + 1) = (bits(first) & 0xffff)  * 2^-16 + 2^-17         -> (0, 1)
// (65 536 levels per coordinate; bits(first + 1) is never used).
// ----------------------------------------------------------------------------
inline uint32_t rng_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
    return x;
}
struct pt_rng {
    uint32_t key;      // K_smp
    uint32_t slot;     // next draw slot
};
inline pt_rng rng_init(uint32_t seed, uint32_t pixel_index, uint32_t sample) {
    uint32_t kpx = rng_hash(pixel_index ^ rng_hash(seed + 0x9E3779B9u));
    pt_rng s;
    s.key = rng_hash(kpx + 0x9E3779B9u * (sample + 1u));
    s.slot = 0;
    return s;
}
inline uint32_t rng_bits(uint32_t key, uint32_t slot) { return rng_hash(key ^ (0x85EBCA6Bu * (slot + 1u))); }
inline float rng_to_uniform(uint32_t bits) { return (float)bits * 2.3283064365386963e-10f + 1.1641532182693481e-10f; }
inline float rng_half_to_uniform(uint32_t h16) { return (float)h16 * 1.52587890625e-05f + 7.62939453125e-06f; }
// The uniform of draw slot k under the slot map above.
inline float rng_slot_uniform(uint32_t key, uint32_t k) {
    uint32_t j = k < 2u ? k + 1u : (k - 2u) % 6u;          // jitter behaves like the pair (1, 2) of a segment
    if (j == 1u || j == 3u) return rng_half_to_uniform(rng_bits(key, k) >> 16);
    if (j == 2u || j == 4u) return rng_half_to_uniform(rng_bits(key, k - 1u) & 0xffffu);
    return rng_to_uniform(rng_bits(key, k));
}
inline float rng_uniform(pt_rng* s) { return rng_slot_uniform(s->key, s->slot++); }

// ----------------------------------------------------------------------------
// bsdf.h:29-69
// ----------------------------------------------------------------------------
inline vec3 sampleHemisphere(float u1, float u2) {      // bsdf.h:29-36 (a uniform unit-sphere point)
    float z = u1 * 2.0f - 1.0f;
    float a = (float)((double)(u2 * 2) * PI);
    float r = sqrtf(1.0f - z * z);
    float x = r * cosf(a);
    float y = r * sinf(a);
    return vec3(x, y, z);
}

inline float FresnelSchlick(float n1, float n2, float cosTheta) {   // bsdf.h:48-51
    float q = (n1 - n2) / (n1 + n2);
    float r0 = q * q;
    float m = 1 - cosTheta;
    float m2 = m * m;
    float m5 = (m2 * m2) * m;
    return r0 + (1 - r0) * m5;
}

inline float getFresnelRatio(vec3 N, vec3 I, float f0, float f90, float n1, float n2) {   // bsdf.h:53-69
    PT_COUNT(fresnel++);
    float cX = -N.dot(I);
    if (n1 > n2) {
        PT_COUNT(fresnel_dense++);
        float n = n1 / n2;
        float s2 = (float)((double)(n * n) * (1.0 - (double)(cX * cX)));
        if ((double)s2 > 1.0) return f90;
        cX = (float)sqrt(1.0 - (double)s2);
    }
    float fresnel_res = FresnelSchlick(n1, n2, cX);
    return lerp(f0, f90, fresnel_res);
}

// ----------------------------------------------------------------------------
// bsdf.h:73-168 — tracePath.  n_bounces is a parameter (reference literal: 5).
// ----------------------------------------------------------------------------
inline vec3 tracePath(Shape* const* scene, int n_objects, ray& r, pt_rng* s, int n_bounces = 5) {
    PT_COUNT(samples++);
    vec3 contribution(0.0f, 0.0f, 0.0f);
    vec3 coefficient = vec3(1.0f, 1.0f, 1.0f);
    for (int i = 0; i <= n_bounces; i++) {
        IntersectionPoint min_p = getNearestIntersection(r, scene, n_objects);
        if (!min_p.intersects) break;
        PT_COUNT(hits++);
        Material material = min_p.material;
        if (material.emissive.norm() > 0) {
            PT_COUNT(emissive++);
            contribution = contribution + material.emissive.cwiseProduct(coefficient);
        }
        if (i > 0 && min_p.inside) {
            PT_COUNT(beer++);
            vec3 attenuation = (-1.0f * min_p.material.refractionColor) * min_p.distance;
            coefficient = coefficient.cwiseProduct(vec3(expf(attenuation[0]), expf(attenuation[1]), expf(attenuation[2])));
        }
        float f0 = material.f0;
        float transparency = material.transparency;
        float p_specular = f0;
        float p_refract = transparency;
        if (f0 > 0.0f) {
            p_specular = getFresnelRatio(r.direction(), min_p.normal, f0, 1.0f,
                                         min_p.inside ? material.IOR : 1.0f, !min_p.inside ? material.IOR : 1.0f);
            p_refract = transparency * (1.0f - p_specular) / (1.0f - f0);
        }
        int mode = 0;
        float u = rng_uniform(s);
        float p_ray;
        if (p_specular > 0.0f && u < p_specular) {
            mode = 1; p_ray = p_specular;
        } else if (p_refract > 0.0f && u < p_specular + p_refract) {
            mode = 2; p_ray = p_refract;
        } else {
            mode = 3; p_ray = 1.0f - (p_specular + p_refract);
        }
        PT_COUNT(mode[mode]++);
        p_ray = fmaxf(p_ray, 0.001f);
        vec3 newPosition = min_p.position;
        float du1 = rng_uniform(s);
        float du2 = rng_uniform(s);
        vec3 diffuseRayDir = (min_p.normal + sampleHemisphere(du1, du2)).normalized();
        vec3 specularRayDir = reflect(r.direction(), min_p.normal).normalized();
        specularRayDir = (lerp(specularRayDir, diffuseRayDir, material.specularRoughness * material.specularRoughness)).normalized();
        vec3 refractionRayDir = refract(r.direction(), min_p.normal,
                                        min_p.inside ? min_p.material.IOR : 1.0f / min_p.material.IOR).normalized();
        float ru1 = rng_uniform(s);
        float ru2 = rng_uniform(s);
        refractionRayDir = (lerp(refractionRayDir, (min_p.normal + sampleHemisphere(ru1, ru2)).normalized(),
                                 material.refractionRoughness * material.refractionRoughness).normalized());
        vec3 newDirection = (mode == 1) ? specularRayDir : (mode == 2) ? refractionRayDir : diffuseRayDir;
        coefficient = mode == 2 ? coefficient : coefficient.cwiseProduct(mode == 1 ? material.specularColor : material.albedo);
        coefficient = coefficient / p_ray;
        float q = fmaxf(coefficient.x(), fmaxf(coefficient.y(), coefficient.z()));
        if (rng_uniform(s) > q) break;
        coefficient = (coefficient * 1.0f) / q;
        r = ray(newPosition, newDirection);
    }
    return contribution;
}

// ----------------------------------------------------------------------------
// bsdf_alternative.h — the class-based BSDF interface (Lambertian /
// SmoothDielectric selected by Material::bsdfType) and its tracePath.  The
// reference does not include this file from main.cu; it is restated here as the
// second shading model.  RNG slots of segment s: 2+6s+{0 dielectric choice,
// 1,2 Lambertian u1,u2, 5 Russian roulette}.  Defined values for what the text
// leaves undefined: the `default:` case of the switch (bsdf_alternative.h:195-199
// fills shadowing locals, the outer ones stay uninitialised) is Lambertian; a
// SmoothDielectric with f0 == 0 and transparency == 0 (`*transmission` never
// written, :131-132) counts as transmission = false.
// ----------------------------------------------------------------------------
struct ONB { vec3 x, y, z; };

inline ONB constructONB(vec3 N) {                       // bsdf_alternative.h:9-18
    float sign = (N.z() < 0 ? -1.0f : 1.0f);
    float a = -1 / (sign + N.z());
    float b = N.x() * N.y() * a;
    ONB res;
    res.x = vec3(1 + sign * (N.x() * N.x()) * a, sign * b, -sign * N.x());
    res.y = vec3(b, sign + (N.y() * N.y()) * a, -N.y());
    res.z = N;
    return res;
}
inline vec3 inverseONB(const ONB& onb, vec3 v) {        // bsdf_alternative.h:24-26
    return v.x() * onb.x + v.y() * onb.y + v.z() * onb.z;
}

struct Lambertian {                                     // bsdf_alternative.h:74-114
    static vec3 evalBSDF(const ray& r, vec3 sampleDirection, const IntersectionPoint& p, const Material& m) {
        vec3 N = p.normal;
        vec3 V = -1.0f * r.direction();
        vec3 L = sampleDirection;
        float NDotL = N.dot(L);
        float NDotV = N.dot(V);
        if (NDotL <= 0.0f || NDotV <= 0.0f) return vec3(0.0f, 0.0f, 0.0f);
        vec3 out = (float)(1.0f / PI) * m.albedo;
        return out * fmaxf(fminf(NDotL, 1.0f), 0.0f);
    }
    static vec3 sampleBSDF(const ray&, const IntersectionPoint& p, const Material&, float u1, float u2) {
        ONB onb = constructONB(p.normal);
        vec3 res = sampleHemisphere(u1, u2);
        return inverseONB(onb, res);
    }
    static float EvalPDF(const ray&, vec3 sampleDirection, const IntersectionPoint& p, const Material&) {
        return (float)((double)fabsf(sampleDirection.dot(p.normal)) * (1.0f / PI));
    }
};

struct SmoothDielectric {                               // bsdf_alternative.h:116-149
    static vec3 evalBSDF(const Material& m) { return m.specularColor; }
    // `u` is drawn by the caller only when it is needed (:133), `need_u` says so.
    static void probabilities(const ray& r, const IntersectionPoint& p, const Material& m, float* p_reflect, float* p_refract) {
        float pr = getFresnelRatio(p.normal, r.direction(), m.f0, 1.0f, p.inside ? m.IOR : 1.0f, !p.inside ? 1.0f : m.IOR);
        float pt = 1 - pr;
        pr *= (m.f0 > 0);
        pt *= (m.transparency > 0);
        *p_reflect = pr; *p_refract = pt;
    }
    static vec3 sampleBSDF(const ray& r, const IntersectionPoint& p, float p_reflect, float p_refract, float u, bool* transmission) {
        if (p_reflect == 0.0f && p_refract == 0.0f) { *transmission = false; return vec3(0.0f, 0.0f, 0.0f); }
        if (u < p_reflect / (p_reflect + p_refract)) {
            *transmission = false;
            return reflect(r.direction(), p.normal).normalized();
        }
        *transmission = true;
        return refract(r.direction(), p.normal, (p.inside ? p.material.IOR : 1.0f / p.material.IOR)).normalized();
    }
    static float EvalPDF(bool transmission, float p_transmission) { return transmission ? p_transmission : 1 - p_transmission; }
};

// bsdf_alternative.h:153-219
inline vec3 tracePathAlternative(Shape* const* scene, int n_objects, ray& r, pt_rng* s, int n_bounces = 5) {
    vec3 contribution(0.0f, 0.0f, 0.0f);
    vec3 coefficient = vec3(1.0f, 1.0f, 1.0f);
    const uint32_t key = s->key;
    for (int i = 0; i <= n_bounces; i++) {
        const uint32_t slot = 2u + 6u * (uint32_t)i;
        IntersectionPoint min_p = getNearestIntersection(r, scene, n_objects);
        if (!min_p.intersects) break;
        Material material = min_p.material;
        if (material.emissive.norm() > 0) contribution = contribution + material.emissive.cwiseProduct(coefficient);
        vec3 newPosition = min_p.position;
        vec3 newDirection, f;
        float pdf;
        if (material.bsdfType == 1) {
            float p_reflect, p_refract;
            SmoothDielectric::probabilities(r, min_p, material, &p_reflect, &p_refract);
            bool transmission;
            float u = (p_reflect == 0.0f && p_refract == 0.0f) ? 0.0f : rng_slot_uniform(key, slot);
            newDirection = SmoothDielectric::sampleBSDF(r, min_p, p_reflect, p_refract, u, &transmission).normalized();
            f = SmoothDielectric::evalBSDF(material);
            pdf = SmoothDielectric::EvalPDF(transmission, p_refract);
        } else {
            float u1 = rng_slot_uniform(key, slot + 1u);
            float u2 = rng_slot_uniform(key, slot + 2u);
            newDirection = Lambertian::sampleBSDF(r, min_p, material, u1, u2).normalized();
            f = Lambertian::evalBSDF(r, newDirection, min_p, material);
            pdf = Lambertian::EvalPDF(r, newDirection, min_p, material);
        }
        if (pdf > 0.0f) coefficient = coefficient.cwiseProduct(f / pdf);
        else break;
        float p = fmaxf(coefficient.x(), fmaxf(coefficient.y(), coefficient.z()));
        float u = rng_slot_uniform(key, slot + 5u);
        if (p < u) break;
        coefficient = coefficient * (1 / p);
        r = ray(newPosition, newDirection);
    }
    return contribution;
}

// ----------------------------------------------------------------------------
// main.cu:22-28 config_t plus the literals the reference hard-codes
// (n_samples main.cu:54, n_bounces bsdf.h:75) and a row range for tiling.
// ----------------------------------------------------------------------------
struct config_t {
    vec3 camera_pos;
    int max_x;
    int max_y;
    float focal_length;
    float view_port_height;
};

// main.cu:229-232 — host-side focal length; nx/ny is an INTEGER division there.
inline float host_focal_length(int nx, int ny, float v_height, float fov_deg) {
    float v_width = (float)(nx / ny) * v_height;
    return (float)((double)(v_width / 2) / tan(DEG_TO_RAD((double)(fov_deg / 2))));
}

// main.cu:30-69 — one pixel of `render` (row j = 0 is the bottom of the image).
inline vec3 renderPixel(const config_t& config, Shape* const* scene, int n_objects, int i, int j,
                        int n_samples, int n_bounces, uint32_t seed, int bsdf_model = 0) {
    int max_x = config.max_x, max_y = config.max_y;
    int pixel_index = j * max_x + i;
    vec3 camera_pos = config.camera_pos;
    float focal_length = config.focal_length;
    float viewport_height = config.view_port_height;
    float viewport_width = viewport_height * (static_cast<float>(max_x) / max_y);
    vec3 viewport_u = vec3(viewport_width, 0, 0);
    vec3 viewport_v = vec3(0, -viewport_height, 0);
    vec3 pixel_delta_u = viewport_u / (float)max_x;
    vec3 pixel_delta_v = viewport_v / (float)max_y;
    vec3 viewport_upper_left = camera_pos - vec3(0, 0, focal_length) - viewport_u / 2.0f - viewport_v / 2.0f;
    vec3 pixel00_loc = viewport_upper_left + (pixel_delta_u + pixel_delta_v) * 0.5f;
    vec3 totalLight(0, 0, 0);
    for (int sample = 0; sample < n_samples; sample++) {
        pt_rng s = rng_init(seed, (uint32_t)pixel_index, (uint32_t)sample);
        float jx = rng_uniform(&s);
        float jy = rng_uniform(&s);
        vec3 p_center = pixel00_loc + (pixel_delta_u * (i - 0.5f + jx)) + (pixel_delta_v * (max_y - j - 0.5f + jy));
        vec3 direction = (p_center - camera_pos).normalized();
        ray r = ray(camera_pos, direction);
        totalLight = totalLight + (bsdf_model == 1 ? tracePathAlternative(scene, n_objects, r, &s, n_bounces)
                                                   : tracePath(scene, n_objects, r, &s, n_bounces));
    }
    return totalLight / ((float)n_samples);
}

}  // namespace ptor
