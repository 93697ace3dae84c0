"""pathtracer_amd — MI355X (gfx950) kernels for the sri-30/pathtracer hot path
(render -> tracePath -> getNearestIntersection -> Shape::getRayPath) behind the
C ABI of include/pt_hip.h.  `binding` loads libpt_hip.so and fails loudly when it
is missing; nothing here computes on the CPU."""
from . import binding  # noqa: F401
from .binding import (Comm, PT_BSDF_CLASSES, PT_BSDF_REFERENCE, PT_CUBE, PT_CYLINDER, PT_FB_AOS, PT_GENERAL_ALL, PT_GENERAL_CUBES, PT_GENERAL_PLANES, PT_GENERAL_SPHERES, PT_FB_SOA, PT_MATH_FAST, PT_MATH_FAST_LDS, PT_MATH_FAST_SCALAR, PT_MATH_PARITY, PT_PLANE,  # noqa: F401
                      PT_SPHERE, PtError, Scene, assemble_rows, comm_unique_id, default_scene, focal_length, glass_grid_scene, make_config, parse_scene, pfm, ppm_p6,
                      ppm_p3, pt_config, pt_material, pt_prim, rendered_rows, rng_draws, synthetic_scene)
