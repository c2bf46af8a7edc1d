/*
 * minimal.c — the C ABI end to end from plain C: build the arguments of Scene::new by hand (one matte sphere under a
 * disk light, the reference's scenes/simple.cry in miniature), run Scene::new (cray_host.h), upload, render
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

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != CRAY_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, cray_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    const char* out_path = argc > 1 ? argv[1] : "minimal.exr";

    /* Material::new_matte(Color(0.8, 0.6, 0.4), sigma = 0): one Lambertian BxDF over a constant texture */
    cray_texture textures[1];
    memset(textures, 0, sizeof(textures));
    textures[0].kind = CRAY_TEX_CONSTANT; textures[0].image = -1;
    textures[0].a.r = 0.8; textures[0].a.g = 0.6; textures[0].a.b = 0.4;
    cray_bxdf bxdfs[1];
    memset(bxdfs, 0, sizeof(bxdfs));
    bxdfs[0].kind = CRAY_BXDF_LAMBERTIAN; bxdfs[0].tex_a = 0; bxdfs[0].tex_b = -1;
    cray_material materials[1] = {{0 /* Material::BxDF */, 1, 0, 0}};

    cray_sphere_desc spheres[1] = {{{0.0, 0.0, 0.0}, 1.0}};
    cray_disk_desc disks[1] = {{{0.0, 3.0, 0.0}, 90.0, 0.0, 1.5, 0.0}};   /* Shape::new_disk(origin, rotate_x, rotate_y, r, r_in) */

    /* primitives: the emissive disk (AreaLightPrimitive), then the sphere; lights: area lights in primitive order */
    cray_prim prims[2] = {{CRAY_SHAPE_DISK, 0, -1, 0}, {CRAY_SHAPE_SPHERE, 0, 0, -1}};
    cray_light lights[1];
    memset(lights, 0, sizeof(lights));
    lights[0].kind = CRAY_LIGHT_AREA; lights[0].prim = 0;
    lights[0].c.r = 4.0; lights[0].c.g = 4.0; lights[0].c.b = 4.0;

    cray_scene_desc desc;
    memset(&desc, 0, sizeof(desc));
    desc.max_depth = 4; desc.num_samples = 8;
    desc.camera.type = CRAY_CAMERA_PERSPECTIVE;
    desc.camera.film_width = 48; desc.camera.film_height = 32;
    desc.camera.origin.x = 0.0; desc.camera.origin.y = 1.0; desc.camera.origin.z = -6.0;
    desc.camera.target.x = 0.0; desc.camera.target.y = 0.5; desc.camera.target.z = 0.0;
    desc.camera.up.y = 1.0;
    desc.camera.fov = 50.0; desc.camera.lens_radius = 0.0; desc.camera.focal_distance = 1e6;
    desc.n_spheres = 1; desc.spheres = spheres;
    desc.n_disks = 1; desc.disks = disks;
    desc.n_prims = 2; desc.prims = prims;
    desc.n_lights = 1; desc.lights = lights;
    desc.n_materials = 1; desc.materials = materials;
    desc.n_bxdfs = 1; desc.bxdfs = bxdfs;
    desc.n_textures = 1; desc.textures = textures;

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
