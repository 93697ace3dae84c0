// linalg.h — host-side value types under the reference's names (vec_math.h:5-30):
// PI, DEG_TO_RAD, SMALL_NUMBER, IDENTITY, vec3, lerp, reflect, refract, and the
// small slice of Eigen's Affine3f / AngleAxisf / Translation3f interface that the
// reference's scene construction touches (main.cu:73-134, shapes.h:36-38).
// Eigen itself is not needed.  These types only build scenes and carry results;
// the hot path (getRayPath, getNearestIntersection, tracePath, render) runs on
// the GPU through the C ABI of pt_hip.h — see api.h.
#pragma once

#include <cmath>

#define PI 3.14159265358979323846
#define DEG_TO_RAD(X) (X * PI) / 180
#define SMALL_NUMBER 0.00001
#define IDENTITY Affine3f::Identity()

// Eigen::Vector3<float> stand-in (vec_math.h:14): reductions sum as x + (y + z),
// normalized() leaves the zero vector alone.
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
    float squaredNorm() const { return dot(*this); }
    float norm() const { return sqrtf(squaredNorm()); }
    vec3 normalized() const {
        const float n = squaredNorm();
        if (n > 0.0f) { const float s = sqrtf(n); return vec3(v[0] / s, v[1] / s, v[2] / s); }
        return *this;
    }
    void normalize() { *this = normalized(); }
    vec3 cwiseProduct(const vec3& o) const { return vec3(v[0] * o.v[0], v[1] * o.v[1], v[2] * o.v[2]); }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]); }
inline vec3 operator-(const vec3& a) { return vec3(-a.v[0], -a.v[1], -a.v[2]); }
inline vec3 operator*(const vec3& a, float s) { return vec3(a.v[0] * s, a.v[1] * s, a.v[2] * s); }
inline vec3 operator*(float s, const vec3& a) { return vec3(s * a.v[0], s * a.v[1], s * a.v[2]); }
inline vec3 operator/(const vec3& a, float s) { return vec3(a.v[0] / s, a.v[1] / s, a.v[2] / s); }
inline vec3& operator+=(vec3& a, const vec3& b) { a = a + b; return a; }

template <typename T> inline T lerp(T v0, T v1, float t) { return (1 - t) * v0 + t * v1; }   // vec_math.h:10-12

inline vec3 reflect(vec3 original, vec3 normal) {   // vec_math.h:19-21
    return original - 2 * (original.dot(normal) * normal);
}

inline vec3 refract(vec3 I, vec3 N, float eta) {    // vec_math.h:23-30 (GLSL semantics, zero vector on TIR)
    const double k = 1.0 - (double)(eta * eta) * (1.0 - (double)((N.dot(I)) * (N.dot(I))));
    const float kf = (float)k;
    if (kf < 0.0f) return vec3(0, 0, 0);
    return eta * I - (eta * (N.dot(I)) + sqrtf(kf)) * N;
}

struct Matrix3f {
    float m[3][3];
};

// vec4, matrix4 and Q4 (vec_math.h:15-17) are declared by the reference and used nowhere in it; they are here as
// plain value types under the same names so that code which mentions them keeps compiling.
struct vec4 {
    float v[4];
    vec4() : v{0.0f, 0.0f, 0.0f, 0.0f} {}
    vec4(float x, float y, float z, float w) : v{x, y, z, w} {}
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float z() const { return v[2]; }
    float w() const { return v[3]; }
    float operator[](int i) const { return v[i]; }
    float& operator[](int i) { return v[i]; }
    float dot(const vec4& o) const { return (v[0] * o.v[0] + v[1] * o.v[1]) + (v[2] * o.v[2] + v[3] * o.v[3]); }
};
struct matrix4 {
    float m[4][4];
    static matrix4 Identity() { matrix4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = i == j ? 1.0f : 0.0f; return r; }
    float operator()(int i, int j) const { return m[i][j]; }
    float& operator()(int i, int j) { return m[i][j]; }
    vec4 operator*(const vec4& p) const {
        vec4 r;
        for (int i = 0; i < 4; i++) r.v[i] = (m[i][0] * p.v[0] + m[i][1] * p.v[1]) + (m[i][2] * p.v[2] + m[i][3] * p.v[3]);
        return r;
    }
};
struct Q4 {                                    // Eigen::Quaternion<float>: (w, x, y, z)
    float qw, qx, qy, qz;
    Q4() : qw(1.0f), qx(0.0f), qy(0.0f), qz(0.0f) {}
    Q4(float w, float x, float y, float z) : qw(w), qx(x), qy(y), qz(z) {}
    float w() const { return qw; }
    float x() const { return qx; }
    float y() const { return qy; }
    float z() const { return qz; }
    static Q4 Identity() { return Q4(); }
};

// Eigen::AngleAxisf(angle, axis).toRotationMatrix()
struct AngleAxisf {
    float angle;
    vec3 axis;
    AngleAxisf(double a, vec3 ax) : angle((float)a), axis(ax) {}
    Matrix3f toRotationMatrix() const {
        Matrix3f r;
        const float s = sinf(angle), c = cosf(angle);
        const vec3 sin_axis = s * axis;
        const vec3 cos1_axis = (1.0f - c) * axis;
        float tmp;
        tmp = cos1_axis.x() * axis.y(); r.m[0][1] = tmp - sin_axis.z(); r.m[1][0] = tmp + sin_axis.z();
        tmp = cos1_axis.x() * axis.z(); r.m[0][2] = tmp + sin_axis.y(); r.m[2][0] = tmp - sin_axis.y();
        tmp = cos1_axis.y() * axis.z(); r.m[1][2] = tmp - sin_axis.x(); r.m[2][1] = tmp + sin_axis.x();
        r.m[0][0] = cos1_axis.x() * axis.x() + c;
        r.m[1][1] = cos1_axis.y() * axis.y() + c;
        r.m[2][2] = cos1_axis.z() * axis.z() + c;
        return r;
    }
};

// Eigen::Translation3f(x, y, z).translation()
struct Translation3f {
    vec3 t;
    Translation3f(double x, double y, double z) : t((float)x, (float)y, (float)z) {}
    vec3 translation() const { return t; }
};

// Eigen::Affine3f stand-in: rows of [L | t].
struct Affine3f {
    float m[3][4];
    struct LinearRef {
        Affine3f* a;
        LinearRef& operator=(const Matrix3f& r) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a->m[i][j] = r.m[i][j]; return *this; }
    };
    struct TranslationRef {
        Affine3f* a;
        TranslationRef& operator=(const vec3& t) { a->m[0][3] = t.v[0]; a->m[1][3] = t.v[1]; a->m[2][3] = t.v[2]; return *this; }
        operator vec3() const { return vec3(a->m[0][3], a->m[1][3], a->m[2][3]); }
    };
    static Affine3f Identity() {
        Affine3f a;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) a.m[i][j] = (i == j) ? 1.0f : 0.0f;
        return a;
    }
    LinearRef linear() { return LinearRef{this}; }
    TranslationRef translation() { return TranslationRef{this}; }
    vec3 translation() const { return vec3(m[0][3], m[1][3], m[2][3]); }
    // scale(): post-multiplies the linear block, translation untouched (main.cu:119,134)
    Affine3f& scale(float s) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] *= s; return *this; }
    Affine3f& scale(const vec3& s) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] *= s.v[j]; return *this; }
    vec3 operator*(const vec3& p) const {
        return vec3(m[0][0] * p.v[0] + (m[0][1] * p.v[1] + m[0][2] * p.v[2]) + m[0][3],
                    m[1][0] * p.v[0] + (m[1][1] * p.v[1] + m[1][2] * p.v[2]) + m[1][3],
                    m[2][0] * p.v[0] + (m[2][1] * p.v[1] + m[2][2] * p.v[2]) + m[2][3]);
    }
};
