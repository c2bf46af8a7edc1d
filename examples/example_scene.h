/*
 * example_scene.h — the arguments of Scene::new for the examples, built by hand: one matte sphere under a disk light
 * (the reference's scenes/simple.cry in miniature).  The arrays live in the caller's `example_scene` so that the
 * cray_scene_desc (and the flat scene that borrows from it) stay valid.
 */
#ifndef CRAY_EXAMPLE_SCENE_H
#define CRAY_EXAMPLE_SCENE_H

#include <string.h>

#include "cray.h"

typedef struct {
    cray_texture textures[1];
    cray_bxdf bxdfs[1];
    cray_material materials[1];
    cray_sphere_desc spheres[1];
    cray_disk_desc disks[1];
    cray_prim prims[2];
    cray_light lights[1];
    cray_scene_desc desc;
} example_scene;

static void example_scene_init(example_scene* s) {
    memset(s, 0, sizeof(*s));
    /* Material::new_matte(Color(0.8, 0.6, 0.4), sigma = 0): one Lambertian BxDF over a constant texture */
    s->textures[0].kind = CRAY_TEX_CONSTANT; s->textures[0].image = -1;
    s->textures[0].a.r = 0.8; s->textures[0].a.g = 0.6; s->textures[0].a.b = 0.4;
    s->bxdfs[0].kind = CRAY_BXDF_LAMBERTIAN; s->bxdfs[0].tex_a = 0; s->bxdfs[0].tex_b = -1;
    s->materials[0].is_bsdf = 0 /* Material::BxDF */; s->materials[0].n_bxdfs = 1; s->materials[0].first_bxdf = 0;

    s->spheres[0].radius = 1.0;                                   /* Shape::new_sphere(origin (0,0,0), 1) */
    s->disks[0].origin.y = 3.0; s->disks[0].rotate_x = 90.0; s->disks[0].radius = 1.5;   /* Shape::new_disk(origin, rotate_x, rotate_y, r, r_in) */

    /* primitives: the emissive disk (AreaLightPrimitive), then the sphere; lights: area lights in primitive order */
    s->prims[0].shape_kind = CRAY_SHAPE_DISK; s->prims[0].shape = 0; s->prims[0].material = -1; s->prims[0].light = 0;
    s->prims[1].shape_kind = CRAY_SHAPE_SPHERE; s->prims[1].shape = 0; s->prims[1].material = 0; s->prims[1].light = -1;
    s->lights[0].kind = CRAY_LIGHT_AREA; s->lights[0].prim = 0;
    s->lights[0].c.r = 4.0; s->lights[0].c.g = 4.0; s->lights[0].c.b = 4.0;

    cray_scene_desc* d = &s->desc;
    d->max_depth = 4; d->num_samples = 8;
    d->camera.type = CRAY_CAMERA_PERSPECTIVE;
    d->camera.film_width = 48; d->camera.film_height = 32;
    d->camera.origin.x = 0.0; d->camera.origin.y = 1.0; d->camera.origin.z = -6.0;
    d->camera.target.x = 0.0; d->camera.target.y = 0.5; d->camera.target.z = 0.0;
    d->camera.up.y = 1.0;
    d->camera.fov = 50.0; d->camera.lens_radius = 0.0; d->camera.focal_distance = 1e6;
    d->n_spheres = 1; d->spheres = s->spheres;
    d->n_disks = 1; d->disks = s->disks;
    d->n_prims = 2; d->prims = s->prims;
    d->n_lights = 1; d->lights = s->lights;
    d->n_materials = 1; d->materials = s->materials;
    d->n_bxdfs = 1; d->bxdfs = s->bxdfs;
    d->n_textures = 1; d->textures = s->textures;
}

#endif
