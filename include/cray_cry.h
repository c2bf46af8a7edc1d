/*
 * cray_cry.h — C ABI of the `.cry` scene reader and OBJ/MTL ingest.
 *
 * Host-side only (no GPU involved); same grammar, defaults and error messages as the reference:
 *   tokenizer::tokenize              src/scene_parser.rs:13-253
 *   parser::RawValue / maps / arrays src/scene_parser.rs:255-773
 *   scene_parser::parse_scene        src/scene_parser.rs:775-1118
 *   obj::load_obj                    src/obj.rs:26-220 (tobj 4.0.0 GPU_LOAD_OPTIONS semantics:
 *                                    triangulate by fan, one vertex per face corner)
 * The result is a `cray_scene_desc` (cray_scene_desc.h), i.e. the arguments of Scene::new, ready
 * for cray_host_scene_new (cray_host.h).
 */
#ifndef CRAY_CRY_H
#define CRAY_CRY_H

#include <stddef.h>
#include <stdint.h>

#include "cray_scene_desc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ParserError {message, location: Option<Location>} (scene_parser.rs:65-84) */
typedef struct {
    int32_t has_location;
    uint32_t line, column;
    char message[512];
} cray_parser_error;

/* TokenValue (scene_parser.rs:17-31) */
enum {
    CRAY_TOK_IDENTIFIER = 0, CRAY_TOK_NUMBER = 1, CRAY_TOK_STRING = 2, CRAY_TOK_LEFT_BRACE = 3,
    CRAY_TOK_RIGHT_BRACE = 4, CRAY_TOK_LEFT_BRACKET = 5, CRAY_TOK_RIGHT_BRACKET = 6, CRAY_TOK_LEFT_PAREN = 7,
    CRAY_TOK_RIGHT_PAREN = 8, CRAY_TOK_COMMA = 9, CRAY_TOK_COLON = 10, CRAY_TOK_EOF = 11
};
typedef struct {
    int32_t kind;
    uint32_t line, column;  /* Location, 1-based */
    double number;          /* CRAY_TOK_NUMBER */
    const char* text;       /* Identifier / String payload (owned by the token array), else NULL */
} cray_token;

/* tokenizer::tokenize. Returns 0 and a token array ending in EOF, or -1 and *err. */
int cray_cry_tokenize(const char* input, cray_token** tokens, size_t* n_tokens, cray_parser_error* err);
void cray_cry_free_tokens(cray_token* tokens, size_t n_tokens);

/* RawValue::from_tokens on the token stream of `input`; on success *dump is a canonical text form
 * of the value tree (test hook): Number(1.23) String("s") Vector(x,y,z) Point(..) Color(..)
 * Map@line:col{key:value,...} (keys sorted) Typed:Name@line:col{...} Array[...]. */
int cray_cry_parse_value(const char* input, char** dump, cray_parser_error* err);
void cray_cry_free_string(char* s);

/* Texture files referenced by MTL files are decoded by the caller (the reference uses the `image`
 * crate): return 0 and a malloc()-ed RGB8 buffer (row-major, w*h*3), or non-zero on failure. */
typedef int (*cray_image_loader)(const char* path, void* user, uint32_t* width, uint32_t* height, uint8_t** rgb8);

/* The reference CLI has no film/spp/depth flags (craytracer.rs:321-334); every BASELINE config
 * overrides them, so the reader can: 0 = keep the file's value. */
typedef struct { uint32_t width, height, num_samples, max_depth; } cray_scene_overrides;

typedef struct cray_owned_scene cray_owned_scene;
/* scene_parser::parse_scene. Mesh file names are resolved against base_dir (NULL = cwd). */
int cray_cry_parse_scene(const char* input, const char* base_dir, cray_image_loader loader, void* loader_user,
                         const cray_scene_overrides* overrides, cray_owned_scene** out, cray_parser_error* err);
const cray_scene_desc* cray_owned_scene_desc(const cray_owned_scene* scene);
/* number of "unused key" warnings the reference would log on Drop (scene_parser.rs:571-586) */
uint32_t cray_owned_scene_warnings(const cray_owned_scene* scene);
void cray_owned_scene_free(cray_owned_scene* scene);

#ifdef __cplusplus
}
#endif
#endif
