// cray_io.cpp — OpenEXR writer for the Film (include/cray_io.h; reference src/bin/craytracer.rs:366-370).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cray.h"
#include "../../include/cray_io.h"

namespace cray {
void set_last_error(const char* fmt, ...);
}

namespace {

void put_u8(std::vector<uint8_t>& b, uint8_t v) { b.push_back(v); }
void put_i32(std::vector<uint8_t>& b, int32_t v) { for (int k = 0; k < 4; k++) b.push_back((uint8_t)((uint32_t)v >> (8 * k))); }
void put_f32(std::vector<uint8_t>& b, float f) { int32_t v; memcpy(&v, &f, 4); put_i32(b, v); }
void put_u64(std::vector<uint8_t>& b, uint64_t v) { for (int k = 0; k < 8; k++) b.push_back((uint8_t)(v >> (8 * k))); }
void put_str(std::vector<uint8_t>& b, const char* s) { while (*s) b.push_back((uint8_t)*s++); b.push_back(0); }
void attr(std::vector<uint8_t>& b, const char* name, const char* type, const std::vector<uint8_t>& value) {
    put_str(b, name); put_str(b, type); put_i32(b, (int32_t)value.size());
    b.insert(b.end(), value.begin(), value.end());
}

}  // namespace

// The preview buffer of `render` (craytracer.rs:69-93, 190-205), headless.
extern "C" void cray_preview_checkerboard(uint32_t w, uint32_t h, uint32_t tw, uint32_t th, uint32_t* out) {
    if (!out || tw == 0 || th == 0) return;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) out[(size_t)y * w + x] = ((x / tw) + (y / th)) % 2 == 0 ? 0x999999u : 0xaaaaaau;
}
extern "C" void cray_preview_pixels(const float* rgb, uint64_t n, double divisor, uint32_t* out) {
    if (!rgb || !out) return;
    auto ch = [&](float v) -> uint32_t {   // Color::to_rgb (color.rs:47-54) on (v as f64) / divisor
        double c = pow((double)v / divisor, 1.0 / 2.2);
        c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);            // f64::clamp; NaN stays NaN ...
        const double s = c * 255.0;
        return s != s ? 0u : (uint32_t)(uint8_t)s;           // ... and `NaN as u8` is 0
    };
    for (uint64_t i = 0; i < n; i++) out[i] = (ch(rgb[3 * i]) << 16) | (ch(rgb[3 * i + 1]) << 8) | ch(rgb[3 * i + 2]);
}

extern "C" int cray_write_exr(const char* path, uint32_t w, uint32_t h, const float* rgb) {
    if (!path || !rgb || w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24)) { cray::set_last_error("cray_write_exr: bad argument"); return CRAY_ERR_INVALID; }
    std::vector<uint8_t> hd;
    const uint8_t magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};  // magic, version 2, single-part scanline
    hd.insert(hd.end(), magic, magic + 8);
    {
        std::vector<uint8_t> v;
        for (const char* ch : {"B", "G", "R"}) {  // channels are stored in alphabetical order
            put_str(v, ch); put_i32(v, 2 /* FLOAT */); put_u8(v, 0); put_u8(v, 0); put_u8(v, 0); put_u8(v, 0);
            put_i32(v, 1); put_i32(v, 1);
        }
        put_u8(v, 0);
        attr(hd, "channels", "chlist", v);
    }
    { std::vector<uint8_t> v; put_u8(v, 0); attr(hd, "compression", "compression", v); }
    for (const char* nm : {"dataWindow", "displayWindow"}) {
        std::vector<uint8_t> v; put_i32(v, 0); put_i32(v, 0); put_i32(v, (int32_t)w - 1); put_i32(v, (int32_t)h - 1);
        attr(hd, nm, "box2i", v);
    }
    { std::vector<uint8_t> v; put_u8(v, 0); attr(hd, "lineOrder", "lineOrder", v); }
    { std::vector<uint8_t> v; put_f32(v, 1.0f); attr(hd, "pixelAspectRatio", "float", v); }
    { std::vector<uint8_t> v; put_f32(v, 0.0f); put_f32(v, 0.0f); attr(hd, "screenWindowCenter", "v2f", v); }
    { std::vector<uint8_t> v; put_f32(v, 1.0f); attr(hd, "screenWindowWidth", "float", v); }
    put_u8(hd, 0);
    const uint64_t row_bytes = (uint64_t)w * 3 * 4, block = 8 + row_bytes;
    const uint64_t first = hd.size() + (uint64_t)h * 8;
    for (uint32_t y = 0; y < h; y++) put_u64(hd, first + (uint64_t)y * block);
    FILE* f = fopen(path, "wb");
    if (!f) { cray::set_last_error("cray_write_exr: cannot open %s", path); return CRAY_ERR_INVALID; }
    bool ok = fwrite(hd.data(), 1, hd.size(), f) == hd.size();
    std::vector<uint8_t> line;
    std::vector<float> plane((size_t)w * 3);
    for (uint32_t y = 0; y < h && ok; y++) {
        line.clear();
        put_i32(line, (int32_t)y); put_i32(line, (int32_t)row_bytes);
        const float* src = rgb + (size_t)y * w * 3;
        for (uint32_t x = 0; x < w; x++) { plane[x] = src[3 * x + 2]; plane[w + x] = src[3 * x + 1]; plane[2 * (size_t)w + x] = src[3 * x]; }
        ok = fwrite(line.data(), 1, line.size(), f) == line.size() && fwrite(plane.data(), 4, plane.size(), f) == plane.size();
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { cray::set_last_error("cray_write_exr: write to %s failed", path); return CRAY_ERR_INVALID; }
    return CRAY_OK;
}

extern "C" int cray_read_exr(const char* path, uint32_t* w_out, uint32_t* h_out, float* rgb, uint64_t cap) {
    if (!path || !w_out || !h_out) { cray::set_last_error("cray_read_exr: null argument"); return CRAY_ERR_INVALID; }
    FILE* f = fopen(path, "rb");
    if (!f) { cray::set_last_error("cray_read_exr: cannot open %s", path); return CRAY_ERR_INVALID; }
    std::vector<uint8_t> d;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + got);
    fclose(f);
    auto fail = [&](const char* why) { cray::set_last_error("cray_read_exr: %s: %s", path, why); return CRAY_ERR_INVALID; };
    if (d.size() < 8 || d[0] != 0x76 || d[1] != 0x2f || d[2] != 0x31 || d[3] != 0x01) return fail("not an OpenEXR file");
    if (d[4] != 2 || d[5] != 0) return fail("only single-part scanline files are supported");
    size_t p = 8;
    auto rd_str = [&](std::string& s) { s.clear(); while (p < d.size() && d[p]) s.push_back((char)d[p++]); if (p >= d.size()) return false; p++; return true; };
    auto rd_i32 = [&](size_t at) { int32_t v; memcpy(&v, &d[at], 4); return v; };
    int32_t x0 = 0, y0 = 0, x1 = -1, y1 = -1;
    int compression = -1;
    std::string chans;
    for (;;) {
        std::string name, type;
        if (!rd_str(name)) return fail("truncated header");
        if (name.empty()) break;
        if (!rd_str(type) || p + 4 > d.size()) return fail("truncated header");
        const int32_t sz = rd_i32(p); p += 4;
        if (sz < 0 || p + (size_t)sz > d.size()) return fail("truncated attribute");
        if (name == "dataWindow" && sz == 16) { x0 = rd_i32(p); y0 = rd_i32(p + 4); x1 = rd_i32(p + 8); y1 = rd_i32(p + 12); }
        if (name == "compression" && sz == 1) compression = d[p];
        if (name == "channels") {
            const size_t end = p + (size_t)sz;   // every read below stays inside the attribute
            size_t q = p;
            while (q < end && d[q]) {
                std::string cn;
                while (q < end && d[q]) cn.push_back((char)d[q++]);
                if (q >= end) return fail("truncated channel list");
                q++;
                if (q + 16 > end) return fail("truncated channel list");
                if (rd_i32(q) != 2) return fail("channel is not FLOAT");
                q += 16;
                chans += cn; chans += ',';
            }
        }
        p += (size_t)sz;
    }
    if (compression != 0) return fail("compressed files are not supported");
    if (chans != "B,G,R,") return fail("expected channels B,G,R");
    if (x0 != 0 || y0 != 0 || x1 < 0 || y1 < 0) return fail("bad dataWindow");
    const uint32_t w = (uint32_t)x1 + 1, h = (uint32_t)y1 + 1;
    // the file must be able to hold what the window announces: offset table + h uncompressed scan-line blocks
    if ((uint64_t)h * (16 + (uint64_t)w * 12) > d.size() - p) return fail("dataWindow larger than the file");
    *w_out = w; *h_out = h;
    if (!rgb) return CRAY_OK;
    if (cap < (uint64_t)w * h * 3) return fail("output buffer too small");
    const size_t table = p;
    if (table + (size_t)h * 8 > d.size()) return fail("truncated offset table");
    for (uint32_t i = 0; i < h; i++) {
        uint64_t off; memcpy(&off, &d[table + (size_t)i * 8], 8);
        if (off > d.size() || (uint64_t)w * 12 + 8 > d.size() - off) return fail("truncated scan line");   // no wrap-around
        const int32_t y = rd_i32(off), nbytes = rd_i32(off + 4);
        if (y < 0 || (uint32_t)y >= h || (uint64_t)nbytes != (uint64_t)w * 12) return fail("bad scan line block");
        const uint8_t* src = &d[off + 8];
        float* dst = rgb + (size_t)y * w * 3;
        for (uint32_t x = 0; x < w; x++) {
            memcpy(&dst[3 * x + 2], src + 4 * (size_t)x, 4);
            memcpy(&dst[3 * x + 1], src + 4 * ((size_t)w + x), 4);
            memcpy(&dst[3 * x], src + 4 * (2 * (size_t)w + x), 4);
        }
    }
    return CRAY_OK;
}
