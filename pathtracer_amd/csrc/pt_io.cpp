// Host-only input/output around the hot path: a text scene description that
// feeds the same pt_prim / pt_material tables as the built-in scenes (the
// reference hard-codes its scene in constructScene, main.cu:71-198, and its
// readme mentions a `scenes` folder that is not in the tree), and binary image
// writers beside the reference's ASCII PPM (main.cu:240-253).
//
// Scene text, one statement per line, '#' starts a comment:
//   material NAME [albedo r g b] [emissive r g b] [f0 x] [specularRoughness x] [specularColor r g b]
//                 [IOR x] [transparency x] [refractionRoughness x] [refractionColor r g b] [bsdfType n]
//   plane    MATERIAL [bounds xMin yMin xMax yMax] OPS...
//   sphere | cube | cylinder  MATERIAL OPS...
// OPS build the shape's Affine3f the way the reference's constructScene does, in the order written:
//   rotate ANGLE ax ay az   linear() = AngleAxisf(ANGLE, axis)   ANGLE in radians: a number, pi, -pi, pi/N, -pi/N
//   translate x y z         translation() = (x, y, z)
//   scale s | scale sx sy sz    scale(...): post-multiplies the linear block
#include "pt_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

const double IO_PI = 3.14159265358979323846;

struct ParseError { std::string msg; };

bool parse_number(const std::string& tok, double* out) {
    // number | [-]pi | [-]pi/N
    std::string t = tok;
    double sign = 1.0;
    if (!t.empty() && t[0] == '-' && t.size() > 1 && t[1] == 'p') { sign = -1.0; t = t.substr(1); }
    if (t.rfind("pi", 0) == 0) {
        double v = IO_PI;
        if (t.size() > 2) {
            if (t[2] != '/') return false;
            char* end = nullptr;
            const double den = std::strtod(t.c_str() + 3, &end);
            if (end == t.c_str() + 3 || *end || den == 0.0) return false;
            v = IO_PI / den;
        }
        *out = sign * v;
        return true;
    }
    char* end = nullptr;
    const double v = std::strtod(tok.c_str(), &end);
    if (end == tok.c_str() || *end) return false;
    *out = v;
    return true;
}

float need_float(const std::vector<std::string>& t, size_t& i, int line) {
    double v;
    if (i >= t.size() || !parse_number(t[i], &v)) {
        char b[96]; std::snprintf(b, sizeof(b), "scene line %d: expected a number", line);
        throw ParseError{b};
    }
    i++;
    return (float)v;
}

void rotate(float m[12], float angle, float ax, float ay, float az) {   // AngleAxisf::toRotationMatrix into linear()
    const float s = sinf(angle), c = cosf(angle);
    const float sx = s * ax, sy = s * ay, sz = s * az, c1 = 1.0f - c;
    const float cx = c1 * ax, cy = c1 * ay, cz = c1 * az;
    float tmp;
    tmp = cx * ay; m[1] = tmp - sz; m[4] = tmp + sz;
    tmp = cx * az; m[2] = tmp + sy; m[8] = tmp - sy;
    tmp = cy * az; m[6] = tmp - sx; m[9] = tmp + sx;
    m[0] = cx * ax + c; m[5] = cy * ay + c; m[10] = cz * az + c;
}

}  // namespace

extern "C" {

int pt_scene_parse(const char* text, pt_prim* prims, int cap_prims, pt_material* mats, int cap_mats, int* n_mats) {
    if (!text || !prims || !mats || !n_mats) { pt_set_error(-1, "pt_scene_parse: null argument", __FILE__, __LINE__); return -1; }
    std::map<std::string, int> mat_index;
    int np = 0, nm = 0, line_no = 0;
    std::istringstream in(text);
    std::string line;
    try {
        while (std::getline(in, line)) {
            line_no++;
            const size_t hash = line.find('#');
            if (hash != std::string::npos) line.resize(hash);
            std::istringstream ls(line);
            std::vector<std::string> t;
            for (std::string w; ls >> w;) t.push_back(w);
            if (t.empty()) continue;
            auto fail = [&](const char* what) { char b[160]; std::snprintf(b, sizeof(b), "scene line %d: %s", line_no, what); throw ParseError{b}; };
            if (t[0] == "material") {
                if (t.size() < 2) fail("material needs a name");
                if (nm >= cap_mats) fail("too many materials");
                pt_material m;
                std::memset(&m, 0, sizeof(m));
                m.IOR = 1.0f;   // materials.h:27
                size_t i = 2;
                while (i < t.size()) {
                    const std::string k = t[i++];
                    auto v3 = [&](float* d) { d[0] = need_float(t, i, line_no); d[1] = need_float(t, i, line_no); d[2] = need_float(t, i, line_no); };
                    if (k == "albedo") v3(m.albedo);
                    else if (k == "emissive") v3(m.emissive);
                    else if (k == "specularColor") v3(m.specularColor);
                    else if (k == "refractionColor") v3(m.refractionColor);
                    else if (k == "f0") m.f0 = need_float(t, i, line_no);
                    else if (k == "specularRoughness") m.specularRoughness = need_float(t, i, line_no);
                    else if (k == "IOR") m.IOR = need_float(t, i, line_no);
                    else if (k == "transparency") m.transparency = need_float(t, i, line_no);
                    else if (k == "refractionRoughness") m.refractionRoughness = need_float(t, i, line_no);
                    else if (k == "bsdfType") m.bsdfType = (int32_t)need_float(t, i, line_no);
                    else fail("unknown material field");
                }
                mat_index[t[1]] = nm;
                mats[nm++] = m;
                continue;
            }
            int type;
            if (t[0] == "plane") type = PT_PLANE;
            else if (t[0] == "cube") type = PT_CUBE;
            else if (t[0] == "cylinder") type = PT_CYLINDER;
            else if (t[0] == "sphere") type = PT_SPHERE;
            else { fail("unknown statement"); return -1; }
            if (t.size() < 2 || !mat_index.count(t[1])) fail("shape needs a material defined above it");
            if (np >= cap_prims) fail("too many shapes");
            pt_prim p;
            std::memset(&p, 0, sizeof(p));
            p.type = type; p.material = mat_index[t[1]];
            float* m = p.transform;
            m[0] = m[5] = m[10] = 1.0f;
            const float big = 3.402823466e+38f;   // Plane's default bounds (shapes.h:111-112)
            if (type == PT_PLANE) { p.bounds[0] = -big; p.bounds[1] = -big; p.bounds[2] = big; p.bounds[3] = big; }
            else { p.bounds[0] = -0.0f; p.bounds[1] = -0.0f; }   // unused by the other shapes; same bits as the built-in scenes
            size_t i = 2;
            while (i < t.size()) {
                const std::string k = t[i++];
                if (k == "bounds") { for (int b = 0; b < 4; b++) p.bounds[b] = need_float(t, i, line_no); }
                else if (k == "rotate") {
                    const float ang = need_float(t, i, line_no), ax = need_float(t, i, line_no), ay = need_float(t, i, line_no), az = need_float(t, i, line_no);
                    rotate(m, ang, ax, ay, az);
                } else if (k == "translate") { m[3] = need_float(t, i, line_no); m[7] = need_float(t, i, line_no); m[11] = need_float(t, i, line_no); }
                else if (k == "scale") {
                    float s[3];
                    s[0] = need_float(t, i, line_no);
                    double probe;
                    if (i < t.size() && parse_number(t[i], &probe)) { s[1] = need_float(t, i, line_no); s[2] = need_float(t, i, line_no); }
                    else { s[1] = s[0]; s[2] = s[0]; }
                    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m[r * 4 + c] *= s[c];
                } else fail("unknown shape option");
            }
            prims[np++] = p;
        }
    } catch (const ParseError& e) {
        pt_set_error(-1, e.msg.c_str(), __FILE__, __LINE__);
        return -1;
    }
    *n_mats = nm;
    return np;
}

// Binary PPM (P6): rows top to bottom, 8 bits per channel, int(255.99*c) clamped to [0,255].
long pt_ppm_p6(const float* fb, int nx, int ny, unsigned char* out, long cap) {
    char head[64];
    const int h = std::snprintf(head, sizeof(head), "P6\n%d %d\n255\n", nx, ny);
    const long need = (long)h + (long)nx * ny * 3;
    if (cap < need) return -1;
    std::memcpy(out, head, (size_t)h);
    unsigned char* q = out + h;
    for (int j = ny - 1; j >= 0; j--)
        for (int i = 0; i < nx; i++)
            for (int c = 0; c < 3; c++) {
                const double v = 255.99 * (double)fb[((size_t)j * nx + i) * 3 + c];
                *q++ = (unsigned char)(v < 0.0 ? 0 : (v > 255.0 ? 255 : (int)v));
            }
    return need;
}

// Portable float map (PF, little endian): rows bottom to top, which is the framebuffer's own order (row 0 = bottom).
long pt_pfm(const float* fb, int nx, int ny, unsigned char* out, long cap) {
    char head[64];
    const int h = std::snprintf(head, sizeof(head), "PF\n%d %d\n-1.0\n", nx, ny);
    const long need = (long)h + (long)nx * ny * 12;
    if (cap < need) return -1;
    std::memcpy(out, head, (size_t)h);
    std::memcpy(out + h, fb, (size_t)nx * ny * 12);
    return need;
}

}  // extern "C"
