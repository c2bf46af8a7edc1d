/*
 * cray_scene_desc.h — plain-C description of a craytracer scene *before* BVH
 * construction: the same information the reference's `Scene::new` receives
 * (reference src/scene.rs:25-31: max_depth, num_samples, camera, lights,
 * primitives), written as POD arrays so that it can cross a C ABI.
 *
 * It is an input *format*, not code: both the product's host layer
 * (include/cray_host.h) and the test oracle (oracle/) read it, so that both
 * sides see identical bytes.
 *
 * Conventions: all floating point is f64 exactly as in the reference
 * (src/geometry.rs:33,219,375 — Vector/Point/Normal are 3×f64; src/color.rs:7 —
 * Color is 3×f64). Indices are 32-bit. Arrays are owned by the caller.
 */
#ifndef CRAY_SCENE_DESC_H
#define CRAY_SCENE_DESC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double x, y, z; } cray_vec3;   /* Point / Vector / Normal */
typedef struct { double r, g, b; } cray_color;  /* src/color.rs:6-11 */

/* ---- Texture<T> (src/texture.rs:7-12) ---------------------------------- */
enum { CRAY_TEX_CONSTANT = 0, CRAY_TEX_CHECKERBOARD = 1, CRAY_TEX_IMAGE = 2 };
typedef struct {
    int32_t kind;
    int32_t image;      /* CRAY_TEX_IMAGE: index into images[] */
    cray_color a;       /* Constant value, or Checkerboard `a`. Texture<f64>: value in .r */
    cray_color b;       /* Checkerboard `b` */
    double scale;       /* Checkerboard scale */
} cray_texture;

/* RgbImage (image crate `to_rgb8()`, src/texture.rs:57-58): RGB8, row-major */
typedef struct {
    uint32_t width, height;
    uint64_t offset;    /* byte offset of pixel (0,0) in image_pool */
} cray_image;

/* ---- BxDF (src/bxdf.rs:23-50) ------------------------------------------ */
enum {
    CRAY_BXDF_LAMBERTIAN = 0,
    CRAY_BXDF_OREN_NAYAR = 1,
    CRAY_BXDF_FRESNEL_CONDUCTOR = 2,
    CRAY_BXDF_SPECULAR_BRDF = 3,
    CRAY_BXDF_SPECULAR_BTDF = 4,
    CRAY_BXDF_FRESNEL_SPECULAR = 5
};
enum { CRAY_FRESNEL_DIELECTRIC = 0, CRAY_FRESNEL_CONDUCTOR = 1 }; /* src/bxdf.rs:333-336 */
typedef struct {
    int32_t kind;
    int32_t tex_a;  /* Lambertian/OrenNayar/SpecularBRDF/FresnelSpecular: reflectance;
                       FresnelConductor: eta; SpecularBTDF: transmittance   (Texture<Color>) */
    int32_t tex_b;  /* OrenNayar: sigma (Texture<f64>); FresnelConductor: k;
                       FresnelSpecular: transmittance; else -1 */
    int32_t fresnel_kind;           /* SpecularBRDF only */
    double eta_i, eta_t;            /* Dielectric / SpecularBTDF / FresnelSpecular */
    cray_color c_eta_i, c_eta_t, c_k; /* Conductor (SpecularBRDF with Fresnel::Conductor) */
} cray_bxdf;

/* ---- Material (src/material.rs:13-17) ---------------------------------- */
typedef struct {
    int32_t is_bsdf;    /* 0: Material::BxDF(bxdf) (n_bxdfs == 1); 1: Material::BSDF(BSDF{bxdfs}) */
    int32_t n_bxdfs;    /* BSDF may hold 0..n lobes (src/material.rs:39-63) */
    int32_t first_bxdf; /* index of the first lobe in bxdfs[] */
    int32_t pad_;
} cray_material;

/* ---- Shapes (src/shape.rs:24-47, constructors :55-153) ------------------ */
enum { CRAY_SHAPE_SPHERE = 0, CRAY_SHAPE_TRIANGLE = 1, CRAY_SHAPE_DISK = 2 };
typedef struct { cray_vec3 origin; double radius; } cray_sphere_desc;
typedef struct {
    cray_vec3 origin;
    double rotate_x, rotate_y;  /* degrees (src/shape.rs:142-143) */
    double radius, inner_radius;
} cray_disk_desc;
/* Shape::Triangle stores (v0,e1,e2,n0,n01,n02,uv0,uv01,uv02) = 24 doubles
 * (src/shape.rs:30-40); the constructors' arithmetic (src/shape.rs:70-132) is
 * done by whoever fills this array. */
typedef struct {
    cray_vec3 v0, e1, e2, n0, n01, n02;
    double uv0[2], uv01[2], uv02[2];
} cray_triangle;

/* ---- Primitive (src/primitive.rs:14-25) --------------------------------- */
typedef struct {
    int32_t shape_kind;
    uint32_t shape;     /* index into spheres[] / triangles[] / disks[] */
    int32_t material;   /* ShapePrimitive: index into materials[]; AreaLightPrimitive: -1
                           (its material is the black matte of src/primitive.rs:40-46) */
    int32_t light;      /* AreaLightPrimitive: index into lights[]; else -1 */
} cray_prim;

/* ---- Light (src/light.rs:25-43) ------------------------------------------ */
enum { CRAY_LIGHT_POINT = 0, CRAY_LIGHT_DISTANT = 1, CRAY_LIGHT_INFINITE = 2, CRAY_LIGHT_AREA = 3 };
typedef struct {
    int32_t kind;
    int32_t prim;       /* Area: index into prims[] of the primitive whose shape emits; else -1 */
    cray_vec3 v;        /* Point: origin; Distant: direction (normalised, scene_parser.rs:888) */
    cray_color c;       /* intensity / emittance */
} cray_light;

/* ---- Camera (src/camera.rs:56-129, src/film.rs) --------------------------- */
enum { CRAY_CAMERA_PERSPECTIVE = 0, CRAY_CAMERA_ORTHOGRAPHIC = 1 };
typedef struct {
    int32_t type;
    uint32_t film_width, film_height;
    int32_t pad_;
    cray_vec3 origin, target, up;
    double fov;             /* degrees; perspective only */
    double lens_radius;     /* default 0 (scene_parser.rs:815) */
    double focal_distance;  /* default 1e6 (scene_parser.rs:798) */
} cray_camera_desc;

/* ---- Scene::new arguments (src/scene.rs:25-31) ----------------------------- */
typedef struct {
    uint32_t max_depth;     /* default 8 (scene_parser.rs:796) */
    uint32_t num_samples;   /* default 4 (scene_parser.rs:797) */
    cray_camera_desc camera;

    uint32_t n_spheres;   const cray_sphere_desc* spheres;
    uint32_t n_disks;     const cray_disk_desc* disks;
    uint32_t n_triangles; const cray_triangle* triangles;

    uint32_t n_prims;     const cray_prim* prims;       /* primitive order = BVH input order */
    uint32_t n_lights;    const cray_light* lights;     /* explicit lights, then area lights in
                                                           primitive order (scene_parser.rs:1093-1101) */
    uint32_t n_materials; const cray_material* materials;
    uint32_t n_bxdfs;     const cray_bxdf* bxdfs;
    uint32_t n_textures;  const cray_texture* textures;
    uint32_t n_images;    const cray_image* images;
    uint64_t image_pool_bytes; const uint8_t* image_pool;
} cray_scene_desc;

#ifdef __cplusplus
}
#endif
#endif /* CRAY_SCENE_DESC_H */
