// pathtracer — the reference's host flow (main.cu:200-257) on MI355X: build the
// scene with the Shape constructors, render through the C ABI, print the image as
// ASCII PPM (P3) on stdout with scanline progress on stderr.  The default scene is the
// library's table of main.cu:73-195 (pt_default_scene); examples/reference_main.cpp is
// the same flow with the reference's own constructScene body.  Every literal of
// the reference's main() is a flag here whose default is that literal:
//   --width 800 --height 800 --spp 2000 --depth 5 --seed 0 --math fast --bsdf reference|classes
//   --scene default|glass[N]|synthetic  --scene-file FILE  --format p3|p6|pfm  --gpus 1  --stats
//   --spp-chunk S: the sample loop as launches of at most S samples each (same image)
//   --checkpoint FILE: the pixels' exact sums and the samples done are written to FILE after every launch (FILE.k for
//     row set k of several); a run that finds them goes on from there.  --stop-after N ends the run (exit status 3, no
//     image) once N samples are done, for a later run with the same flags to finish
//   --tiles T (default = gpus): the image is split into T interleaved row sets, set k rendered on GPU k % gpus
//   --gather host|rccl: tiles copied to the host image one by one (default), or all-gathered between the GPUs with RCCL
// Usage, as in the reference's readme:  pathtracer > out.ppm
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <fstream>
#include <sstream>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pt/api.h"

// Shapes from a flat description (the library's glass-grid and synthetic builders).
static int shapesFromTable(const std::vector<pt_prim>& prims, const std::vector<pt_material>& mats, std::vector<Shape*>& scene) {
    for (const pt_prim& p : prims) {
        Material m;
        const pt_material& a = mats[p.material];
        m.albedo = vec3(a.albedo[0], a.albedo[1], a.albedo[2]); m.emissive = vec3(a.emissive[0], a.emissive[1], a.emissive[2]);
        m.f0 = a.f0; m.specularRoughness = a.specularRoughness; m.specularColor = vec3(a.specularColor[0], a.specularColor[1], a.specularColor[2]);
        m.IOR = a.IOR; m.transparency = a.transparency; m.refractionRoughness = a.refractionRoughness;
        m.refractionColor = vec3(a.refractionColor[0], a.refractionColor[1], a.refractionColor[2]); m.bsdfType = a.bsdfType;
        Affine3f t;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) t.m[i][j] = p.transform[i * 4 + j];
        switch (p.type) {
            case PT_PLANE: scene.push_back(new Plane(m, t, p.bounds[0], p.bounds[1], p.bounds[2], p.bounds[3])); break;
            case PT_CUBE: scene.push_back(new Cube(m, t)); break;
            case PT_CYLINDER: scene.push_back(new Cylinder(m, t)); break;
            default: scene.push_back(new Sphere(m, t)); break;
        }
    }
    return (int)scene.size();
}

int main(int argc, char** argv) {
    int nx = 800, ny = 800, n_gpus = 1, n_tiles = 0;
    render_options opt;                 // n_samples 2000, n_bounces 5 as in the reference
    std::string scene_name = "default", scene_file, format = "p3";
    bool stats = false, gather_rccl = false;
    std::string checkpoint;
    int stop_after = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::cerr << "missing value for " << a << "\n"; std::exit(2); } return argv[++i]; };
        if (a == "--width") nx = std::atoi(next());
        else if (a == "--height") ny = std::atoi(next());
        else if (a == "--spp") opt.n_samples = std::atoi(next());
        else if (a == "--depth") opt.n_bounces = std::atoi(next());
        else if (a == "--spp-chunk") opt.spp_chunk = std::atoi(next());
        else if (a == "--seed") opt.seed = (unsigned)std::strtoul(next(), nullptr, 10);
        else if (a == "--math") opt.math_mode = std::string(next()) == "parity" ? PT_MATH_PARITY : PT_MATH_FAST;
        else if (a == "--bsdf") opt.bsdf_model = std::string(next()) == "classes" ? PT_BSDF_CLASSES : PT_BSDF_REFERENCE;
        else if (a == "--scene") scene_name = next();
        else if (a == "--scene-file") scene_file = next();
        else if (a == "--format") format = next();
        else if (a == "--gpus") n_gpus = std::atoi(next());
        else if (a == "--tiles") n_tiles = std::atoi(next());
        else if (a == "--gather") gather_rccl = std::string(next()) == "rccl";
        else if (a == "--stats") stats = true;
        else if (a == "--checkpoint") checkpoint = next();
        else if (a == "--stop-after") stop_after = std::atoi(next());
        else { std::cerr << "unknown flag " << a << "\n"; return 2; }
    }
    if (nx <= 0 || ny <= 0 || n_gpus < 1) { std::cerr << "bad size or gpu count\n"; return 2; }
    int n_dev = 0;
    checkHipErrors(hipGetDeviceCount(&n_dev));
    if (n_gpus > n_dev) { std::cerr << "asked for " << n_gpus << " GPUs, " << n_dev << " visible\n"; return 2; }

    /* Scene */
    std::vector<Shape*> scene;
    if (format != "p3" && format != "p6" && format != "pfm") { std::cerr << "unknown format " << format << "\n"; return 2; }
    if (!scene_file.empty()) {
        std::ifstream f(scene_file);
        if (!f) { std::cerr << "cannot open " << scene_file << "\n"; return 2; }
        std::stringstream ss;
        ss << f.rdbuf();
        const std::string text = ss.str();
        int cap = 1;
        for (char c : text) cap += c == '\n';
        std::vector<pt_prim> prims((size_t)cap);
        std::vector<pt_material> mats((size_t)cap);
        int nm = 0;
        const int n = pt_scene_parse(text.c_str(), prims.data(), cap, mats.data(), cap, &nm);
        if (n <= 0) { std::cerr << (n < 0 ? pt_last_error() : "scene file has no shapes") << "\n"; return 2; }
        prims.resize((size_t)n);
        shapesFromTable(prims, mats, scene);
    } else {
        std::vector<pt_prim> prims;
        std::vector<pt_material> mats(8);
        int nm = 0, n = 0;
        if (scene_name == "default") {
            prims.resize(11);
            n = pt_default_scene(prims.data(), mats.data(), &nm);
        } else if (scene_name.rfind("glass", 0) == 0) {
            const int g = scene_name.size() > 5 ? std::atoi(scene_name.c_str() + 5) : 5;
            prims.resize(8 + (size_t)g * g);
            n = pt_glass_grid_scene(g, prims.data(), mats.data(), &nm);
        } else if (scene_name == "synthetic") {
            prims.resize(10000);
            n = pt_synthetic_scene(5000, 4989, 12345u, prims.data(), mats.data(), &nm);
        } else { std::cerr << "unknown scene " << scene_name << "\n"; return 2; }
        prims.resize((size_t)n);
        shapesFromTable(prims, mats, scene);
    }
    const int n_objs = (int)scene.size();

    /* Camera (main.cu:229-234; nx/ny is an integer division there) */
    float v_height = 5.0;
    float v_width = nx / ny * v_height;
    float fov = 45;
    float focal_length = (v_width / 2) / tan(DEG_TO_RAD(fov / 2));
    config_t config = {vec3(0, 0, 0), nx, ny, focal_length, v_height};

    /* Render: row set k = rows k, k + n_tiles, ... (rows through the glass cost more than wall rows, so interleaving
       balances the devices) goes to GPU k % n_gpus, rendered into a compact tile; a strided copy puts it into the
       host image */
    if (n_tiles < n_gpus) n_tiles = n_gpus;
    const size_t num_pixels = (size_t)nx * ny;
    std::vector<vec3> fb(num_pixels);
    std::vector<double> ms(n_tiles, 0.0);
    std::vector<unsigned long long> segments((size_t)n_tiles, 0ull);
    /* --gather rccl: one row set per GPU; the tiles are all-gathered over xGMI (pt_gather_framebuffer), every GPU ends
       up with the whole frame and GPU 0's copy goes to the host.  Default: each tile is copied to the host image. */
    std::vector<pt_comm*> comms;
    if (gather_rccl) {
        n_tiles = n_gpus;
        std::vector<int> devices((size_t)n_gpus);
        for (int g = 0; g < n_gpus; g++) devices[(size_t)g] = g;
        comms.assign((size_t)n_gpus, nullptr);
        /* RCCL may print a version banner on stdout while it starts; stdout carries the image, so it goes to stderr. */
        std::cout.flush(); std::fflush(stdout);
        const int saved_stdout = dup(STDOUT_FILENO);
        dup2(STDERR_FILENO, STDOUT_FILENO);
        const int rc = pt_comm_init_all(n_gpus, devices.data(), comms.data());
        std::fflush(stdout);
        dup2(saved_stdout, STDOUT_FILENO);
        close(saved_stdout);
        if (rc != 0) { std::cerr << pt_last_error() << "\n"; return 99; }
    }
    const int tile_rows = pt_tile_rows(ny, n_tiles);
    std::vector<float*> d_gathered((size_t)n_tiles, nullptr), d_full((size_t)n_tiles, nullptr);
    std::vector<vec3*> d_tiles((size_t)n_tiles, nullptr);
    std::vector<unsigned long long*> d_counts((size_t)n_tiles, nullptr);
    std::vector<int> samples_done((size_t)n_tiles, opt.n_samples);
    const unsigned long long tag = checkpoint.empty() ? 0ull : pt::scene_tag(scene.data(), n_objs);
    auto worker = [&](int k) {
        const int g = k % n_gpus;
        checkHipErrors(hipSetDevice(g));
        render_options o = opt;
        o.device = g;
        o.row_begin = k; o.row_end = ny; o.row_step = n_tiles; o.fb_compact = 1;
        const int my_rows = k < ny ? (ny - k + n_tiles - 1) / n_tiles : 0;
        if (my_rows == 0 && !gather_rccl) return;
        pt::DeviceScene dscene(scene.data(), n_objs, g);
        vec3* d_fb = nullptr;
        checkHipErrors(hipMalloc((void**)&d_fb, (size_t)tile_rows * nx * sizeof(vec3)));
        /* The collective's buffers exist before anything is launched and are freed only after every thread has
           joined: hipMalloc / hipFree synchronise the device, and doing that while another rank's RCCL kernel of the
           same collective is in flight is how a process that drives several GPUs deadlocks. */
        const size_t row_bytes = (size_t)nx * sizeof(vec3);
        if (gather_rccl) {
            checkHipErrors(hipMalloc((void**)&d_gathered[(size_t)k], (size_t)n_tiles * tile_rows * row_bytes));
            checkHipErrors(hipMalloc((void**)&d_full[(size_t)k], (size_t)ny * row_bytes));
        }
        d_tiles[(size_t)k] = d_fb;
        unsigned long long* d_counters = nullptr;
        if (stats && o.spp_chunk <= 0 && checkpoint.empty()) {
            checkHipErrors(hipMalloc((void**)&d_counters, PT_COUNTER_COUNT * sizeof(unsigned long long)));
            checkHipErrors(hipMemset(d_counters, 0, PT_COUNTER_COUNT * sizeof(unsigned long long)));
            o.d_counters = d_counters;
            d_counts[(size_t)k] = d_counters;             // (freed with the other buffers, after every thread has joined)
        }
        hipEvent_t e0, e1;
        checkHipErrors(hipEventCreate(&e0)); checkHipErrors(hipEventCreate(&e1));
        checkHipErrors(hipEventRecord(e0, nullptr));
        if (my_rows > 0 && !checkpoint.empty()) {
            render_checkpoint ck;
            ck.path = n_tiles > 1 ? checkpoint + "." + std::to_string(k) : checkpoint;
            ck.scene_tag = tag; ck.stop_after = stop_after;
            samples_done[(size_t)k] = render_resumable(d_fb, config, dscene, o, ck);
        } else if (my_rows > 0) {
            render(d_fb, config, dscene, o);
        }
        checkHipErrors(hipEventRecord(e1, nullptr));
        checkHipErrors(hipGetLastError());
        checkHipErrors(hipDeviceSynchronize());
        float t = 0.0f;
        checkHipErrors(hipEventElapsedTime(&t, e0, e1));
        ms[k] = t;
        if (d_counters) {
            unsigned long long c[PT_COUNTER_COUNT];
            checkHipErrors(hipMemcpy(c, d_counters, sizeof(c), hipMemcpyDeviceToHost));
            segments[(size_t)k] = c[PT_COUNTER_SEGMENTS];
        }
        if (gather_rccl) {
            if (pt_gather_framebuffer(comms[(size_t)k], reinterpret_cast<const float*>(d_fb), d_gathered[(size_t)k], d_full[(size_t)k], ny, nx, nullptr) != 0) {
                std::cerr << pt_last_error() << "\n";
                std::exit(99);
            }
            checkHipErrors(hipDeviceSynchronize());
            if (k == 0) checkHipErrors(hipMemcpy(fb.data(), d_full[0], (size_t)ny * row_bytes, hipMemcpyDeviceToHost));
        } else {
            checkHipErrors(hipMemcpy2D(fb.data() + (size_t)k * nx, row_bytes * n_tiles, d_fb, row_bytes, row_bytes, my_rows, hipMemcpyDeviceToHost));
        }
    };
    std::vector<std::thread> threads;
    for (int k = 1; k < n_tiles; k++) threads.emplace_back(worker, k);
    worker(0);
    for (auto& t : threads) t.join();
    for (int k = 0; k < n_tiles; k++) {
        if (!d_tiles[(size_t)k]) continue;
        checkHipErrors(hipSetDevice(k % n_gpus));
        checkHipErrors(hipFree(d_tiles[(size_t)k]));
        if (d_counts[(size_t)k]) checkHipErrors(hipFree(d_counts[(size_t)k]));
        if (d_gathered[(size_t)k]) checkHipErrors(hipFree(d_gathered[(size_t)k]));
        if (d_full[(size_t)k]) checkHipErrors(hipFree(d_full[(size_t)k]));
    }
    for (pt_comm* c : comms) pt_comm_free(c);
    for (int done : samples_done) {
        if (done < 0) return 2;                   // the state file belongs to another frame, or cannot be written
        if (done < opt.n_samples) {
            std::cerr << "stopped after " << done << " of " << opt.n_samples << " samples; the same command goes on from " << checkpoint << "\n";
            for (Shape* s : scene) delete s;
            return 3;
        }
    }

    /* --gather rccl --stats: the gathered frame against the frame one GPU renders alone, byte for byte */
    int gathered_equal = -1;
    if (stats && gather_rccl) {
        checkHipErrors(hipSetDevice(0));
        render_options o = opt;
        o.device = 0;
        pt::DeviceScene dscene(scene.data(), n_objs, 0);
        vec3* d_ref = nullptr;
        checkHipErrors(hipMalloc((void**)&d_ref, num_pixels * sizeof(vec3)));
        render(d_ref, config, dscene, o);
        std::vector<vec3> ref(num_pixels);
        checkHipErrors(hipMemcpy(ref.data(), d_ref, num_pixels * sizeof(vec3), hipMemcpyDeviceToHost));
        checkHipErrors(hipFree(d_ref));
        gathered_equal = std::memcmp(ref.data(), fb.data(), num_pixels * sizeof(vec3)) == 0 ? 1 : 0;
    }
    if (stats) {
        double worst = 0.0;
        for (double t : ms) worst = std::max(worst, t);
        const double samples = (double)num_pixels * opt.n_samples;
        double segs = 0.0;
        for (unsigned long long s : segments) segs += (double)s;
        // Arithmetic per sample by the event table the benchmark prices its roofline with (add/sub/mul/min/max/div/sqrt/
        // transcendental = 1, FMA = 2; camera ray 28; per segment a 33-flop transform + the shape's test per primitive
        // (plane 7, sphere 21, cube 25, cylinder 49), the winner's record 53, shading 20 + a diffuse direction 23): the
        // terms every segment has.  Glass hits and roots of hit spheres add a few per cent on the default scene.
        double per_segment = 53.0 + 20.0 + 23.0;
        for (int k = 0; k < n_objs; k++) {
            const int type = scene[(size_t)k]->description().type;
            per_segment += 33.0 + (type == PT_PLANE ? 7.0 : type == PT_SPHERE ? 21.0 : type == PT_CUBE ? 25.0 : 49.0);
        }
        const double flops_per_sample = segs > 0.0 ? 28.0 + per_segment * segs / samples : 0.0;
        const double tflops = flops_per_sample * samples / (worst * 1e-3) / 1e12;
        std::clog << "{\"width\": " << nx << ", \"height\": " << ny << ", \"spp\": " << opt.n_samples << ", \"depth\": " << opt.n_bounces
                  << ", \"gpus\": " << n_gpus << ", \"samples\": " << samples << ", \"segments\": " << segs
                  << ", \"segments_per_sample\": " << segs / samples << ", \"kernel_ms\": " << worst
                  << ", \"msamples_per_s\": " << samples / worst / 1e3 << ", \"flops_per_sample_event_table\": " << flops_per_sample
                  << ", \"algorithmic_tflops\": " << tflops << ", \"pct_of_fp32_vector_peak\": " << 100.0 * tflops / (157.3 * n_gpus);
        if (gathered_equal >= 0) std::clog << ", \"gather\": \"pt_gather_framebuffer\", \"bitwise_equal_to_single_gpu\": " << (gathered_equal ? "true" : "false");
        std::clog << "}\n";
        if (gathered_equal == 0) { std::cerr << "the gathered frame differs from the single-GPU frame\n"; return 1; }
    }

    if (format != "p3") {
        const long cap = 64 + (long)num_pixels * 12;
        std::vector<unsigned char> buf((size_t)cap);
        const float* f = reinterpret_cast<const float*>(fb.data());
        const long n = format == "p6" ? pt_ppm_p6(f, nx, ny, buf.data(), cap) : pt_pfm(f, nx, ny, buf.data(), cap);
        std::cout.write(reinterpret_cast<const char*>(buf.data()), n);
        for (Shape* s : scene) delete s;
        return 0;
    }

    /* Output as PPM Image (main.cu:239-253) */
    std::cout << "P3\n" << nx << " " << ny << "\n255\n";
    for (int j = ny - 1; j >= 0; j--) {
        std::clog << "\rScanlines remaining: " << (ny - j) << ' ' << std::flush;
        for (int i = 0; i < nx; i++) {
            size_t pixel_index = (size_t)j * nx + i;
            float r = fb[pixel_index].x();
            float g = fb[pixel_index].y();
            float b = fb[pixel_index].z();
            int ir = int(255.99 * r);
            int ig = int(255.99 * g);
            int ib = int(255.99 * b);
            std::cout << ir << " " << ig << " " << ib << "\n";
        }
    }
    for (Shape* s : scene) delete s;
    return 0;
}
