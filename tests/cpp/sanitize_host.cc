// Host-code run under AddressSanitizer + UndefinedBehaviorSanitizer (built and run by tests/test_sanitizers.py):
//   * the CPU oracle (oracle/oracle_capi.cpp): scene construction, every shape's getRayPath / getIntersections on
//     ordinary and degenerate rays (the two uninitialised reads of the reference, shapes.h:117-121,132, are defined
//     values here and must stay so), getNearestIntersection, both tracePath variants, render on two threads;
//   * the scene-text parser and the binary image writers of the product (pathtracer_amd/csrc/pt_io.cpp) on the shipped
//     scene file, on hand-written malformed input and on a few thousand seeded mutations of the scene file.
// Exit code 0 and no sanitizer report = pass.  Nothing here touches a GPU.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "pt_hip.h"

extern "C" {
struct oracle_material { float albedo[3], emissive[3], f0, specularRoughness, specularColor[3], IOR, transparency, refractionRoughness, refractionColor[3]; int bsdfType; };
struct oracle_prim { int type, material; float transform[12], bounds[4]; };
struct oracle_config { float camera_pos[3]; int max_x, max_y; float focal_length, view_port_height; int n_samples, n_bounces; unsigned seed; int row_begin, row_end, bsdf_model; };
struct oracle_stats;
void* oracle_scene_create(const oracle_prim*, int, const oracle_material*, int);
void oracle_scene_free(void*);
int oracle_default_scene(oracle_prim*, oracle_material*, int*);
float oracle_focal_length(int, int, float, float);
void oracle_raypath_batch(const oracle_prim*, const float*, const float*, int, int, int*, int*, float*, float*, float*, int*, float*, float*, float*);
void oracle_nearest_batch(void*, const float*, const float*, int, int*, float*, float*, float*, int*);
void oracle_trace_paths(void*, const float*, const float*, const unsigned*, const unsigned*, int, unsigned, int, float*);
void oracle_trace_paths_alt(void*, const float*, const float*, const unsigned*, const unsigned*, int, unsigned, int, float*);
void oracle_rng(unsigned, unsigned, unsigned, int, unsigned*, float*);
int oracle_render(void*, const oracle_config*, float*, int, oracle_stats*);
long oracle_ppm(const float*, int, int, char*, long);
}

// pt_io.cpp reports through the library's error slot; this program stands in for it.
static char g_err[512];
int pt_set_error(int code, const char* expr, const char* file, int line) {
    std::snprintf(g_err, sizeof(g_err), "HIP error = %u at %s:%d '%s' ", (unsigned)code, file, line, expr);
    return code ? code : -1;
}

static uint32_t g_state = 0x1234567u;
static uint32_t rnd() { g_state = g_state * 1664525u + 1013904223u; return g_state >> 8; }
static float frand(float lo, float hi) { return lo + (hi - lo) * (float)(rnd() & 0xffff) / 65535.0f; }

static int check(bool ok, const char* what) {
    if (!ok) { std::fprintf(stderr, "sanitize_host: FAILED %s\n", what); std::exit(1); }
    return 0;
}

static void run_oracle() {
    oracle_prim prims[11];
    oracle_material mats[8];
    int nm = 0;
    const int n = oracle_default_scene(prims, mats, &nm);
    check(n == 11 && nm == 8, "default scene size");
    void* sc = oracle_scene_create(prims, n, mats, nm);
    check(sc != nullptr, "scene create");

    // rays: random, plus the degenerate ones (zero direction, axis-parallel, origin on the surface / inside)
    const int N = 4096;
    std::vector<float> o(3 * N), d(3 * N);
    for (int i = 0; i < N; i++) {
        for (int k = 0; k < 3; k++) { o[3 * i + k] = frand(-2.0f, 2.0f); d[3 * i + k] = frand(-1.0f, 1.0f); }
        if (i % 16 == 1) { d[3 * i] = 0.0f; d[3 * i + 1] = 0.0f; d[3 * i + 2] = 0.0f; }
        if (i % 16 == 2) { d[3 * i + 2] = 0.0f; }
        if (i % 16 == 3) { d[3 * i] = 0.0f; d[3 * i + 1] = 0.0f; }
        if (i % 16 == 4) { o[3 * i] = 0.0f; o[3 * i + 1] = 0.0f; o[3 * i + 2] = 0.0f; }
        if (i % 16 == 5) { o[3 * i] = 0.5f; o[3 * i + 1] = 0.0f; o[3 * i + 2] = 0.0f; }
    }
    std::vector<int> ni(N), h1(N), h2(N), prim(N), inside(N);
    std::vector<float> t1(N), t2(N), p1(3 * N), p2(3 * N), n1(3 * N), n2(3 * N), t(N), pos(3 * N), nrm(3 * N);
    for (int type = 0; type < 4; type++) {
        oracle_prim p;
        std::memset(&p, 0, sizeof(p));
        p.type = type;
        p.transform[0] = p.transform[5] = p.transform[10] = 1.0f;
        p.bounds[0] = -1.0f; p.bounds[1] = -1.0f; p.bounds[2] = 1.0f; p.bounds[3] = 1.0f;
        for (int filter = 0; filter < 2; filter++) {
            oracle_raypath_batch(&p, o.data(), d.data(), N, filter, ni.data(), h1.data(), t1.data(), p1.data(), n1.data(),
                                 h2.data(), t2.data(), p2.data(), n2.data());
            for (int i = 0; i < N; i++) check(ni[i] >= 0 && ni[i] <= 2, "n_intersections in 0..2");
        }
    }
    for (int k = 0; k < n; k++)
        oracle_raypath_batch(&prims[k], o.data(), d.data(), N, 1, ni.data(), h1.data(), t1.data(), p1.data(), n1.data(),
                             h2.data(), t2.data(), p2.data(), n2.data());
    oracle_nearest_batch(sc, o.data(), d.data(), N, prim.data(), t.data(), pos.data(), nrm.data(), inside.data());
    for (int i = 0; i < N; i++) check(prim[i] >= -1 && prim[i] < n, "nearest primitive index");

    std::vector<unsigned> px(N), smp(N);
    for (int i = 0; i < N; i++) { px[i] = rnd(); smp[i] = rnd() & 1023u; }
    std::vector<float> rgb(3 * N);
    oracle_trace_paths(sc, o.data(), d.data(), px.data(), smp.data(), N, 7u, 5, rgb.data());
    oracle_trace_paths(sc, o.data(), d.data(), px.data(), smp.data(), N, 7u, 0, rgb.data());
    oracle_trace_paths(sc, o.data(), d.data(), px.data(), smp.data(), N, 7u, 31, rgb.data());
    oracle_trace_paths_alt(sc, o.data(), d.data(), px.data(), smp.data(), N, 7u, 5, rgb.data());

    unsigned bits[64]; float u[64];
    oracle_rng(0u, 0xffffffffu, 0xffffffffu, 64, bits, u);
    for (int k = 0; k < 64; k++) check(u[k] > 0.0f && u[k] <= 1.0f, "uniform in (0,1]");

    for (int model = 0; model < 2; model++) {
        oracle_config cfg;
        std::memset(&cfg, 0, sizeof(cfg));
        cfg.max_x = 48; cfg.max_y = 32;
        cfg.view_port_height = 5.0f; cfg.focal_length = oracle_focal_length(48, 32, 5.0f, 45.0f);
        cfg.n_samples = 4; cfg.n_bounces = 5; cfg.seed = 0u; cfg.row_begin = 0; cfg.row_end = 32; cfg.bsdf_model = model;
        std::vector<float> fb(48 * 32 * 3, -1.0f);
        check(oracle_render(sc, &cfg, fb.data(), 2, nullptr) == 0, "render");
        for (float v : fb) check(v >= 0.0f, "rendered radiance is non-negative");
        std::vector<char> ppm(48 * 32 * 16 + 64);
        check(oracle_ppm(fb.data(), 48, 32, ppm.data(), (long)ppm.size()) > 0, "ppm");
        check(oracle_ppm(fb.data(), 48, 32, ppm.data(), 8) < 0, "ppm with a short buffer is refused");
    }
    // an empty scene and a scene whose material index is out of range
    void* empty = oracle_scene_create(prims, 0, mats, nm);
    if (empty) {
        oracle_nearest_batch(empty, o.data(), d.data(), 64, prim.data(), t.data(), pos.data(), nrm.data(), inside.data());
        for (int i = 0; i < 64; i++) check(prim[i] == -1, "empty scene: every ray misses");
        oracle_scene_free(empty);
    }
    oracle_prim bad = prims[0];
    bad.material = 99;
    void* refused = oracle_scene_create(&bad, 1, mats, nm);
    if (refused) oracle_scene_free(refused);
    oracle_scene_free(sc);
}

static void parse(const std::string& text, int cap_prims, int cap_mats, int* n_out = nullptr) {
    std::vector<pt_prim> prims((size_t)(cap_prims > 0 ? cap_prims : 1));
    std::vector<pt_material> mats((size_t)(cap_mats > 0 ? cap_mats : 1));
    int nm = 0;
    const int n = pt_scene_parse(text.c_str(), prims.data(), cap_prims, mats.data(), cap_mats, &nm);
    check(n <= cap_prims && nm <= cap_mats, "parser stays inside the caller's tables");
    for (int k = 0; k < n; k++) check(prims[(size_t)k].material >= 0 && prims[(size_t)k].material < nm, "parsed material index");
    if (n_out) *n_out = n;
}

static void run_parser(const char* scene_path) {
    std::ifstream f(scene_path);
    check((bool)f, "scene file opens");
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    int n = 0;
    parse(text, 64, 64, &n);
    check(n == 11, "default.scene has 11 shapes");
    {   // a null argument is an error (-1), never a primitive count
        pt_prim one; pt_material m; int nm = 0;
        check(pt_scene_parse(nullptr, &one, 1, &m, 1, &nm) == -1, "null text is refused");
        check(pt_scene_parse(text.c_str(), nullptr, 1, &m, 1, &nm) == -1, "null primitive table is refused");
        check(pt_scene_parse(text.c_str(), &one, 1, nullptr, 1, &nm) == -1, "null material table is refused");
        check(pt_scene_parse(text.c_str(), &one, 1, &m, 1, nullptr) == -1, "null material count is refused");
    }
    parse(text, 3, 64, &n);   check(n == -1, "too few shape slots is an error");
    parse(text, 64, 2, &n);   check(n == -1, "too few material slots is an error");
    parse(text, 0, 0, &n);
    const char* malformed[] = {
        "", "\n\n\n", "#", "material", "material m albedo", "material m albedo 1 2", "material m albedo 1 2 x",
        "sphere", "sphere nosuch", "material m\nsphere m scale", "material m\nsphere m scale 1 2", "material m\nplane m bounds 1 2 3",
        "material m\nsphere m rotate pi/2", "material m\nsphere m rotate pi/0 1 0 0", "material m\nsphere m translate 1e999 -1e999 nan",
        "material m IOR 1e-999 f0 inf\ncube m scale 0\ncylinder m scale -1 0 1", "material m bsdfType 1e30", "teapot m",
        "material m\nsphere m frobnicate 3", "material \xff\xfe albedo 1 1 1\nsphere \xff\xfe", "material m\rsphere m",
        "material m albedo 1 1 1 # trailing\nsphere m#x\n#\n", "material m\nsphere m scale 1 1", "material a\nmaterial a\nsphere a",
    };
    for (const char* m : malformed) parse(m, 8, 8);
    std::string longline = "material m albedo";
    for (int k = 0; k < 20000; k++) longline += " 1";
    parse(longline, 8, 8);
    // seeded mutations of the shipped scene: bytes replaced, lines cut, tokens doubled
    for (int round = 0; round < 3000; round++) {
        std::string s = text;
        const int edits = 1 + (int)(rnd() % 6u);
        for (int e = 0; e < edits && !s.empty(); e++) {
            const size_t at = rnd() % s.size();
            switch (rnd() % 5u) {
                case 0: s[at] = (char)(rnd() & 0xff); if (s[at] == '\0') s[at] = ' '; break;
                case 1: s.erase(at, 1 + rnd() % 9u); break;
                case 2: s.insert(at, s.substr(at, 1 + rnd() % 12u)); break;
                case 3: s.resize(at); break;
                default: s[at] = "\n #-.e9"[rnd() % 7u]; break;
            }
        }
        parse(s, 16, 16);
    }
    // image writers: exact-size, short and zero-size buffers
    std::vector<float> fb(7 * 5 * 3);
    for (size_t k = 0; k < fb.size(); k++) fb[k] = frand(-1.0f, 3.0f);
    fb[0] = 1e30f; fb[1] = -1e30f; fb[2] = 0.0f / 1.0f;
    std::vector<unsigned char> out(64 + 7 * 5 * 12);
    const long p6 = pt_ppm_p6(fb.data(), 7, 5, out.data(), (long)out.size());
    check(p6 > 0 && pt_ppm_p6(fb.data(), 7, 5, out.data(), p6) == p6 && pt_ppm_p6(fb.data(), 7, 5, out.data(), p6 - 1) < 0, "p6 sizes");
    const long pf = pt_pfm(fb.data(), 7, 5, out.data(), (long)out.size());
    check(pf > 0 && pt_pfm(fb.data(), 7, 5, out.data(), pf) == pf && pt_pfm(fb.data(), 7, 5, out.data(), pf - 1) < 0, "pfm sizes");
    check(pt_ppm_p6(fb.data(), 7, 5, out.data(), 0) < 0, "p6 with no room");
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: sanitize_host scenes/default.scene\n"); return 2; }
    run_oracle();
    run_parser(argv[1]);
    std::printf("sanitize_host: ok\n");
    return 0;
}
