/*
 * cray_io.h — output of the Film the render seam returns.
 *
 * The reference hands the Vec<f32> of `render` to image::Rgb32FImage and saves it by file extension
 * (src/bin/craytracer.rs:366-370; the CLI default is an .exr path, :327-330): linear, un-clamped f32 RGB.
 * cray_write_exr writes the same pixels as a single-part scanline OpenEXR 2.0 file with three FLOAT
 * channels (B, G, R), no compression, INCREASING_Y — every reader accepts it; byte-for-byte equality with
 * the `exr` crate's file (which may compress) is not a goal, pixel equality is.
 */
#ifndef CRAY_IO_H
#define CRAY_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rgb: width*height*3 floats, row-major, y down (the layout cray_render fills). Returns 0 or CRAY_ERR_INVALID
 * (message in cray_last_error). */
int cray_write_exr(const char* path, uint32_t width, uint32_t height, const float* rgb);

/* Reads back a file written by cray_write_exr (uncompressed scanline FLOAT B/G/R only): for tests and tools.
 * Call with rgb == NULL to get the size. */
int cray_read_exr(const char* path, uint32_t* width, uint32_t* height, float* rgb, uint64_t capacity_floats);

/* The preview buffer of `render` (src/bin/craytracer.rs:69-93, 190-205) without the window: what a host blits while a frame
 * is in flight.  minifb's window itself (:45-66, 95-144) stays with the host.
 *   cray_preview_checkerboard: the initial pattern, 0x999999 / 0xaaaaaa per tile (tile_x + tile_y even / odd), :78-91
 *   cray_preview_pixels: pixel i = to_rgb(Color(rgb[3i..3i+3]) / divisor) packed 0x00RRGGBB (:192-204; Color::to_rgb,
 *     src/color.rs:47-54: powf(1 / 2.2), clamp, * 255, `as u8`).  With the film cray_render returns for samples [0, s)
 *     (already divided by num_samples) pass divisor = s / num_samples to get the running average the reference shows. */
void cray_preview_checkerboard(uint32_t width, uint32_t height, uint32_t tile_width, uint32_t tile_height, uint32_t* out);
void cray_preview_pixels(const float* rgb, uint64_t n_pixels, double divisor, uint32_t* out);

/* Texture files for hosts without an image library — what the reference does with the `image` crate
 * (`image::io::Reader::open(path).decode()` + `.to_rgb8()`, src/obj.rs:16-24, src/texture.rs:57-58):
 *   PNM (P6 / P3 / P5 / P2), PNG (every colour type and bit depth, Adam7, own inflate; alpha dropped, 16-bit scaled by
 *   (x + 128) / 257 like the crate's to_rgb8) and JPEG (8-bit Huffman: baseline, extended sequential and progressive).
 * JPEG arithmetic is the IJG reference decoder's (islow IDCT, fancy upsampling), i.e. libjpeg-turbo's pixels; the Rust
 * jpeg-decoder may differ from it by a level or two.  *rgb8: malloc()-ed width*height*3 bytes, row-major; release with
 * cray_free_image (or free).  Returns 0, CRAY_ERR_INVALID (unreadable / corrupt) or CRAY_ERR_UNSUPPORTED (other formats). */
int cray_load_image(const char* path, uint32_t* width, uint32_t* height, uint8_t** rgb8);
void cray_free_image(uint8_t* rgb8);
/* The same as a cray_image_loader (cray_cry.h): cray_cry_parse_scene uses it when the caller passes no loader. */
int cray_default_image_loader(const char* path, void* user, uint32_t* width, uint32_t* height, uint8_t** rgb8);

#ifdef __cplusplus
}
#endif
#endif
