/*
 * minimal.c — the C ABI end to end from plain C: build the arguments of Scene::new by hand (example_scene.h: one matte
 * sphere under a disk light, the reference's scenes/simple.cry in miniature), run Scene::new (cray_host.h), upload, render
 * (cray.h) and write the film as OpenEXR (cray_io.h).  No Python, no torch.
 *
 *   gcc -std=c11 -Iinclude examples/minimal.c -Lcraytracer_amd/csrc -lcray_hip -Wl,-rpath,$PWD/craytracer_amd/csrc -lm -o minimal
 *   ./minimal out.exr
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cray.h"
#include "cray_host.h"
#include "cray_io.h"
#include "example_scene.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != CRAY_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, cray_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    const char* out_path = argc > 1 ? argv[1] : "minimal.exr";

    example_scene ex;
    example_scene_init(&ex);
    const cray_scene_desc desc = ex.desc;

    cray_ctx* ctx = NULL;
    CHECK(cray_ctx_create(0, NULL, &ctx));
    cray_host_scene* host = NULL;
    CHECK(cray_host_scene_new_on(&desc, CRAY_SPLIT_SAH, ctx, &host));   /* Scene::new, Bvh::new on the GPU */
    cray_scene* scene = NULL;
    CHECK(cray_scene_upload(ctx, cray_host_scene_flat(host), &scene));

    cray_render_params prm;
    cray_render_params_default(&prm);
    prm.seed = 2;
    const size_t n = (size_t)desc.camera.film_width * desc.camera.film_height * 3;
    float* film = (float*)malloc(n * sizeof(float));
    cray_stats st;
    CHECK(cray_render(ctx, scene, &prm, film, &st));
    CHECK(cray_write_exr(out_path, desc.camera.film_width, desc.camera.film_height, film));
    double mean = 0.0;
    for (size_t i = 0; i < n; i++) mean += film[i];
    printf("%llu paths, %llu + %llu rays, mean radiance %.9g, %.3f ms -> %s\n", (unsigned long long)st.paths,
           (unsigned long long)st.closest_rays, (unsigned long long)st.shadow_rays, mean / (double)n, st.seconds * 1e3, out_path);

    free(film);
    cray_scene_free(scene);
    cray_host_scene_free(host);
    cray_ctx_destroy(ctx);
    return 0;
}
