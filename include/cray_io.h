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

#ifdef __cplusplus
}
#endif
#endif
