// The reference's host program (main.cu:71-257) with its CUDA calls replaced by this library's source-level API:
// constructScene keeps its body as a host function (same transforms, same materials, same eleven `new`
// expressions), the render<<<>>> launch becomes render(fb, config, scene, n_objs), the PPM loop is unchanged.
// The literals of main() that make the reference a many-minute render (800 x 800, 2000 samples) can be overridden:
//   reference_main [width height spp] > out.ppm
// The product's own program is pathtracer_amd/host/main.cpp; both print the same file for the same size and samples.
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "../include/pt/api.h"

// constructScene (main.cu:71-198) as a host function: same transforms, same
// materials, same eleven `new` expressions.
static int constructScene(Shape** scene) {
    Affine3f floor = IDENTITY;
    floor.linear() = AngleAxisf(PI / 2, vec3(1, 0, 0)).toRotationMatrix();
    floor.translation() = Translation3f(0.0, -1.0, 0.0).translation();
    Affine3f back = IDENTITY;
    back.translation() = Translation3f(0.0, 0, -5.0).translation();
    Affine3f left = IDENTITY;
    left.linear() = AngleAxisf(PI / 2, vec3(0, 1, 0)).toRotationMatrix();
    left.translation() = Translation3f(-1.0, 0.0, -5.0).translation();
    Affine3f right = IDENTITY;
    right.linear() = AngleAxisf(-PI / 2, vec3(0, 1, 0)).toRotationMatrix();
    right.translation() = Translation3f(1.0, 0.0, -5.0).translation();
    Affine3f ceiling = IDENTITY;
    ceiling.linear() = AngleAxisf(-PI / 2, vec3(1, 0, 0)).toRotationMatrix();
    ceiling.translation() = Translation3f(0.0, 1.0, 0.0).translation();
    Affine3f light = IDENTITY;
    light.linear() = AngleAxisf(-PI / 2, vec3(1, 0, 0)).toRotationMatrix();
    light.translation() = Translation3f(0.0, 0.97, -3.25).translation();
    Affine3f tplane6 = IDENTITY;
    tplane6.linear() = AngleAxisf(PI, vec3(0, 1, 0)).toRotationMatrix();
    tplane6.translation() = Translation3f(0.0, 0.0, 7.0).translation();
    Affine3f t5 = IDENTITY;
    t5.translation() = Translation3f(0.0, -0.5, -3.0).translation();
    t5.scale(0.4f);
    Affine3f t6 = IDENTITY;
    t6.translation() = Translation3f(0.6, -0.5, -3.0).translation();
    t6.scale(0.4f);
    Affine3f t7 = IDENTITY;
    t7.translation() = Translation3f(-0.6, -0.5, -3.0).translation();
    t7.scale(0.4f);
    Affine3f t8 = IDENTITY;
    t8.translation() = Translation3f(0.0f, -0.5f, -4.2f).translation();
    t8.scale(vec3(2.5f, 0.3f, 0.5f));

    Material light_material;
    light_material.emissive = vec3(1.0f, 0.9f, 0.7f) * 20.0f;
    Material base;
    base.albedo = vec3(0.4f, 0.4f, 0.4f);
    Material green;
    green.albedo = vec3(0.0f, 1.0f, 0.0f);
    Material red;
    red.albedo = vec3(1.0f, 0.0f, 0.0f);
    Material glass;
    glass.f0 = 0.02f;
    glass.specularRoughness = 0;
    glass.specularColor = vec3(1.0f, 1.0f, 1.0f) * 0.8f;
    glass.IOR = 1.5f;
    glass.transparency = 1.0f;
    glass.refractionRoughness = 0.0f;
    glass.bsdfType = 1;
    Material diffuse_pure;
    diffuse_pure.albedo = vec3(0.6f, 0.6f, 0.6f);
    Material glass1 = glass, glass2 = glass, glass3 = glass;
    glass2.refractionColor = vec3(0.9f, 0.0f, 0.0f);
    glass3.refractionColor = vec3(2.0f, 0.0f, 0.0f);

    scene[0] = new Plane(base, floor, -15.0f, -15.0f, 15.0f, 15.0f);
    scene[1] = new Plane(red, left, -15.0f, -15.0f, 15.0f, 15.0f);
    scene[2] = new Plane(green, right, -15.0f, -15.0f, 15.0f, 15.0f);
    scene[3] = new Plane(base, back, -15.0f, -15.0f, 15.0f, 15.0f);
    scene[4] = new Plane(base, ceiling, -15.0f, -15.0f, 15.0f, 15.0f);
    scene[5] = new Plane(light_material, light, -0.5f, -0.5f, 0.5f, 0.5f);
    scene[6] = new Plane(base, tplane6, -15.0f, -15.0f, 15.0f, 15.0f);
    scene[7] = new Sphere(glass1, t7);
    scene[8] = new Sphere(glass2, t5);
    scene[9] = new Sphere(glass3, t6);
    scene[10] = new Cube(diffuse_pure, t8);
    return 11;
}

int main(int argc, char** argv) {
    int nx = 800;
    int ny = 800;
    render_options opt;                 // n_samples 2000 (main.cu:54), n_bounces 5 (bsdf.h:75)
    if (argc >= 4) { nx = std::atoi(argv[1]); ny = std::atoi(argv[2]); opt.n_samples = std::atoi(argv[3]); }
    if (nx <= 0 || ny <= 0 || opt.n_samples <= 0) { std::cerr << "usage: reference_main [width height spp]\n"; return 2; }

    std::cerr << "Rendering a " << nx << "x" << ny << " image ";
    std::cerr << "in " << 8 << "x" << 8 << " blocks.\n";

    int n_objs = 11;
    std::vector<Shape*> scene((size_t)n_objs);
    constructScene(scene.data());

    size_t num_pixels = (size_t)nx * ny;
    vec3* d_fb = nullptr;
    checkHipErrors(hipMalloc((void**)&d_fb, num_pixels * sizeof(vec3)));

    float v_height = 5.0;
    float v_width = nx / ny * v_height;
    float fov = 45;
    float focal_length = (v_width / 2) / tan(DEG_TO_RAD(fov / 2));
    config_t config = {vec3(0, 0, 0), nx, ny, focal_length, v_height};

    render(d_fb, config, scene.data(), n_objs, opt);
    checkHipErrors(hipGetLastError());
    checkHipErrors(hipDeviceSynchronize());
    std::vector<vec3> fb(num_pixels);
    checkHipErrors(hipMemcpy(fb.data(), d_fb, num_pixels * sizeof(vec3), hipMemcpyDeviceToHost));

    // Output as PPM Image
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
    checkHipErrors(hipFree(d_fb));
    for (Shape* s : scene) delete s;
    return 0;
}
