// cray_image.cpp — texture file decoding for hosts without an image library (include/cray_io.h: cray_load_image).
//
// The reference decodes `map_Kd` / `map_Ks` files with the `image` crate and converts to RGB8
// (src/obj.rs:16-24, src/texture.rs:57-58: `image::io::Reader::open(path).decode()`, `.to_rgb8()`).  This file reads
// the two formats the path needs without any dependency:
//   * binary / ASCII PNM (P6, P3, P5, P2) — exact by definition;
//   * PNG: all colour types and bit depths, interlaced or not, with its own inflate (no zlib); alpha is dropped and 16-bit
//     samples are scaled like the `image` crate's `to_rgb8` ((x + 128) / 257);
//   * JPEG, 8-bit Huffman, baseline / extended sequential (SOF0, SOF1) and progressive (SOF2), 1 or 3 components,
//     any sampling factors, restart intervals.  Arithmetic follows the IJG reference decoder (integer "islow" IDCT,
//     triangle-filter "fancy" chroma upsampling for 2x1 / 2x2, the ycc -> rgb integer tables), so the pixels equal
//     libjpeg / libjpeg-turbo's (what Pillow returns; tested on the ten textures of objs/staircase/textures).
//     The Rust `jpeg-decoder` behind the `image` crate uses its own IDCT / upsampling: its pixels may differ from any
//     libjpeg's by a level or two, so texture-mapped pixels are pinned to the *decoded* RGB8 data, not to the .jpg.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <new>
#include <stdexcept>
#include <vector>

#include "../../include/cray.h"
#include "../../include/cray_io.h"

namespace cray {
void set_last_error(const char* fmt, ...);
}

namespace {

bool read_whole_file(const char* path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
    out.resize((size_t)n);
    bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

// ------------------------------------------------------------------------------------------- PNM
bool pnm_token(const std::vector<uint8_t>& d, size_t& p, long& v) {
    for (;;) {
        while (p < d.size() && (d[p] == ' ' || d[p] == '\t' || d[p] == '\n' || d[p] == '\r')) p++;
        if (p < d.size() && d[p] == '#') { while (p < d.size() && d[p] != '\n') p++; continue; }
        break;
    }
    if (p >= d.size() || d[p] < '0' || d[p] > '9') return false;
    v = 0;
    while (p < d.size() && d[p] >= '0' && d[p] <= '9') { v = v * 10 + (d[p] - '0'); if (v > (1L << 30)) return false; p++; }
    return true;
}

int decode_pnm(const std::vector<uint8_t>& d, uint32_t* w, uint32_t* h, uint8_t** rgb) {
    const int kind = d[1] - '0';  // 2 ASCII gray, 3 ASCII rgb, 5 binary gray, 6 binary rgb
    size_t p = 2;
    long W, H, maxv;
    if (!pnm_token(d, p, W) || !pnm_token(d, p, H) || !pnm_token(d, p, maxv) || W < 1 || H < 1 || maxv < 1 || maxv > 65535 || (uint64_t)W * (uint64_t)H > (1ull << 31)) {
        cray::set_last_error("PNM: bad header");
        return CRAY_ERR_INVALID;
    }
    const int ch = (kind == 3 || kind == 6) ? 3 : 1;
    const size_t n = (size_t)W * H;
    uint8_t* out = (uint8_t*)malloc(n * 3);
    if (!out) { cray::set_last_error("PNM: out of memory"); return CRAY_ERR_INVALID; }
    auto scale = [&](long v) -> uint8_t {
        if (v > maxv) v = maxv;
        return maxv == 255 ? (uint8_t)v : (uint8_t)((v * 255 + maxv / 2) / maxv);
    };
    bool ok = true;
    if (kind == 5 || kind == 6) {
        p++;  // the single whitespace byte after maxval
        const size_t bps = maxv > 255 ? 2 : 1;
        if (p + n * ch * bps > d.size()) ok = false;
        for (size_t i = 0; ok && i < n; i++)
            for (int c = 0; c < 3; c++) {
                const size_t at = p + (i * ch + (ch == 3 ? c : 0)) * bps;
                const long v = bps == 2 ? ((long)d[at] << 8 | d[at + 1]) : d[at];
                out[3 * i + c] = scale(v);
            }
    } else {
        for (size_t i = 0; ok && i < n; i++) {
            long v[3] = {0, 0, 0};
            for (int c = 0; c < ch; c++) if (!pnm_token(d, p, v[c])) { ok = false; break; }
            for (int c = 0; c < 3; c++) out[3 * i + c] = scale(v[ch == 3 ? c : 0]);
        }
    }
    if (!ok) { free(out); cray::set_last_error("PNM: truncated pixel data"); return CRAY_ERR_INVALID; }
    *w = (uint32_t)W; *h = (uint32_t)H; *rgb = out;
    return CRAY_OK;
}

// ------------------------------------------------------------------------------------------- PNG
// RFC 1951 inflate: stored, fixed and dynamic Huffman blocks
struct Inflate {
    const uint8_t* d; size_t n, pos = 0;
    uint32_t bits = 0; int nbits = 0;
    bool ok = true;
    size_t limit = ~(size_t)0;   // the caller knows how many bytes the stream may legitimately expand to
    int bit() {
        if (nbits == 0) { if (pos >= n) { ok = false; return 0; } bits = d[pos++]; nbits = 8; }
        int b = bits & 1; bits >>= 1; nbits--; return b;
    }
    uint32_t take(int k) { uint32_t v = 0; for (int i = 0; i < k; i++) v |= (uint32_t)bit() << i; return v; }
    struct Table { uint16_t count[16], symbol[320]; };
    static void build(Table& t, const uint8_t* len, int n_sym) {
        memset(t.count, 0, sizeof(t.count));
        for (int i = 0; i < n_sym; i++) t.count[len[i]]++;
        t.count[0] = 0;
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + t.count[l]);
        for (int i = 0; i < n_sym; i++) if (len[i]) t.symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(const Table& t) {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; l++) {
            code |= bit();
            const int count = t.count[l];
            if (code - count < first) return t.symbol[index + (code - first)];
            index += count; first += count; first <<= 1; code <<= 1;
            if (!ok) return -1;
        }
        ok = false;
        return -1;
    }
    bool run(std::vector<uint8_t>& out) {
        static const uint16_t lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
        static const uint8_t lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
        static const uint16_t dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
        static const uint8_t dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
        for (;;) {
            const int last = bit(), type = (int)take(2);
            if (!ok) return false;
            if (type == 0) {
                nbits = 0;
                if (pos + 4 > n) return false;
                const uint32_t len = d[pos] | (d[pos + 1] << 8), nlen = d[pos + 2] | (d[pos + 3] << 8);
                pos += 4;
                if ((len ^ 0xffffu) != nlen || pos + len > n) return false;
                if (out.size() + len > limit) return false;
                out.insert(out.end(), d + pos, d + pos + len);
                pos += len;
            } else if (type == 1 || type == 2) {
                Table lt, dt;
                uint8_t lens[320];
                if (type == 1) {
                    for (int i = 0; i < 144; i++) lens[i] = 8;
                    for (int i = 144; i < 256; i++) lens[i] = 9;
                    for (int i = 256; i < 280; i++) lens[i] = 7;
                    for (int i = 280; i < 288; i++) lens[i] = 8;
                    build(lt, lens, 288);
                    for (int i = 0; i < 30; i++) lens[i] = 5;
                    build(dt, lens, 30);
                } else {
                    const int nl = (int)take(5) + 257, nd = (int)take(5) + 1, nc = (int)take(4) + 4;
                    static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
                    uint8_t cl[19] = {0};
                    for (int i = 0; i < nc; i++) cl[order[i]] = (uint8_t)take(3);
                    if (!ok || nl > 286 || nd > 30) return false;
                    Table ct;
                    build(ct, cl, 19);
                    int idx = 0;
                    while (idx < nl + nd) {
                        const int sym = decode(ct);
                        if (sym < 0) return false;
                        if (sym < 16) lens[idx++] = (uint8_t)sym;
                        else {
                            int prev = 0, rep;
                            if (sym == 16) { if (idx == 0) return false; prev = lens[idx - 1]; rep = 3 + (int)take(2); }
                            else if (sym == 17) rep = 3 + (int)take(3);
                            else rep = 11 + (int)take(7);
                            if (idx + rep > nl + nd) return false;
                            while (rep--) lens[idx++] = (uint8_t)prev;
                        }
                    }
                    if (lens[256] == 0) return false;
                    build(lt, lens, nl);
                    build(dt, lens + nl, nd);
                }
                for (;;) {
                    int sym = decode(lt);
                    if (sym < 0 || !ok) return false;
                    if (out.size() > limit) return false;
                    if (sym < 256) out.push_back((uint8_t)sym);
                    else if (sym == 256) break;
                    else {
                        sym -= 257;
                        if (sym >= 29) return false;
                        const int len = lbase[sym] + (int)take(lext[sym]);
                        const int ds = decode(dt);
                        if (ds < 0 || ds >= 30) return false;
                        const size_t dist = dbase[ds] + take(dext[ds]);
                        if (!ok || dist > out.size()) return false;
                        const size_t from = out.size() - dist;
                        for (int i = 0; i < len; i++) out.push_back(out[from + i]);
                    }
                }
            } else return false;
            if (last) return ok;
        }
    }
};

int decode_png(const std::vector<uint8_t>& d, uint32_t* w_out, uint32_t* h_out, uint8_t** rgb) {
    auto fail = [](const char* m) { cray::set_last_error("PNG: %s", m); return CRAY_ERR_INVALID; };
    auto be32 = [&](size_t at) { return ((uint32_t)d[at] << 24) | ((uint32_t)d[at + 1] << 16) | ((uint32_t)d[at + 2] << 8) | d[at + 3]; };
    size_t p = 8;
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool seen_end = false;
    while (p + 12 <= d.size() && !seen_end) {
        const uint32_t len = be32(p);
        if (len > d.size() - p - 12) return fail("truncated chunk");
        const char* ty = (const char*)&d[p + 4];
        const size_t at = p + 8;
        if (!memcmp(ty, "IHDR", 4)) {
            if (len < 13) return fail("bad IHDR");
            W = be32(at); H = be32(at + 4); depth = d[at + 8]; ctype = d[at + 9]; interlace = d[at + 12];
            if (d[at + 10] != 0 || d[at + 11] != 0 || interlace > 1) return fail("unsupported compression / filter / interlace method");
            if ((uint64_t)W * (uint64_t)H > (1ull << 28)) return fail("image too large (more than 2^28 pixels)");
        } else if (!memcmp(ty, "PLTE", 4)) plte.assign(d.begin() + at, d.begin() + at + len);
        else if (!memcmp(ty, "IDAT", 4)) idat.insert(idat.end(), d.begin() + at, d.begin() + at + len);
        else if (!memcmp(ty, "IEND", 4)) seen_end = true;
        p += 12 + (size_t)len;
    }
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!channels || !depth_ok || W == 0 || H == 0 || (uint64_t)W * H > (1ull << 31)) return fail("bad header");
    if (ctype == 3 && plte.size() < 3) return fail("palette image without PLTE");
    if (idat.size() < 6) return fail("no image data");
    Inflate z;
    z.d = idat.data() + 2; z.n = idat.size() - 2;   // zlib header: CMF, FLG (no preset dictionary in PNG)
    if ((idat[0] & 15) != 8 || (idat[1] & 0x20)) return fail("bad zlib header");
    std::vector<uint8_t> raw;
    const size_t row_bytes = ((size_t)W * channels * depth + 7) / 8;
    z.limit = (size_t)H * (row_bytes + 1) + (interlace ? 8 * ((size_t)H + row_bytes) + 64 : 0) + 258;   // filter bytes included; Adam7 pads every sub-row
    raw.reserve(z.limit < ((size_t)1 << 28) ? z.limit : ((size_t)1 << 28));
    if (!z.run(raw)) return fail("corrupt deflate stream");
    uint8_t* out = (uint8_t*)malloc((size_t)W * H * 3);
    if (!out) return fail("out of memory");
    const int bpp_bits = channels * depth, bpp = (bpp_bits + 7) / 8;   // bytes per complete pixel for the filters
    auto sample16 = [](const uint8_t* q) { const uint32_t v = ((uint32_t)q[0] << 8) | q[1]; return (uint8_t)((v + 128) / 257); };
    // one pass (the whole image, or one of the seven Adam7 sub-images): unfilter, then scatter into `out`
    size_t rp = 0;
    std::vector<uint8_t> prev, cur;
    auto pass = [&](uint32_t x0, uint32_t y0, uint32_t dx, uint32_t dy) -> bool {
        if (x0 >= W || y0 >= H) return true;
        const uint32_t pw = (W - x0 + dx - 1) / dx, ph = (H - y0 + dy - 1) / dy;
        const size_t stride = ((size_t)pw * bpp_bits + 7) / 8;
        prev.assign(stride, 0); cur.resize(stride);
        for (uint32_t r = 0; r < ph; r++) {
            if (rp + 1 + stride > raw.size()) return false;
            const int ft = raw[rp++];
            const uint8_t* src = &raw[rp];
            rp += stride;
            for (size_t i = 0; i < stride; i++) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
                int pred = 0;
                if (ft == 1) pred = a;
                else if (ft == 2) pred = b;
                else if (ft == 3) pred = (a + b) >> 1;
                else if (ft == 4) { const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
                else if (ft != 0) return false;
                cur[i] = (uint8_t)(src[i] + pred);
            }
            const uint32_t y = y0 + r * dy;
            for (uint32_t i = 0; i < pw; i++) {
                uint8_t* o = out + ((size_t)y * W + (x0 + i * dx)) * 3;
                if (depth < 8) {
                    const size_t bit = (size_t)i * depth;
                    const int v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                    if (ctype == 3) { if ((size_t)v * 3 + 2 < plte.size()) { o[0] = plte[v * 3]; o[1] = plte[v * 3 + 1]; o[2] = plte[v * 3 + 2]; } else { o[0] = o[1] = o[2] = 0; } }
                    else { o[0] = o[1] = o[2] = (uint8_t)(v * 255 / ((1 << depth) - 1)); }
                } else {
                    const uint8_t* q = &cur[(size_t)i * bpp];
                    const int sb = depth / 8;
                    if (ctype == 3) { const size_t v = q[0]; if (v * 3 + 2 < plte.size()) { o[0] = plte[v * 3]; o[1] = plte[v * 3 + 1]; o[2] = plte[v * 3 + 2]; } else { o[0] = o[1] = o[2] = 0; } }
                    else if (ctype == 0 || ctype == 4) { o[0] = o[1] = o[2] = depth == 8 ? q[0] : sample16(q); }
                    else { for (int k = 0; k < 3; k++) o[k] = depth == 8 ? q[k] : sample16(q + k * sb); }
                }
            }
            prev.swap(cur);
            cur.resize(stride);
        }
        return true;
    };
    bool ok = true;
    if (!interlace) ok = pass(0, 0, 1, 1);
    else {
        static const uint32_t ax[7] = {0, 4, 0, 2, 0, 1, 0}, ay[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int k = 0; k < 7 && ok; k++) ok = pass(ax[k], ay[k], adx[k], ady[k]);
    }
    if (!ok) { free(out); return fail("truncated or corrupt image data"); }
    *w_out = W; *h_out = H; *rgb = out;
    return CRAY_OK;
}

// ------------------------------------------------------------------------------------------- JPEG
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int mincode[18], maxcode[18], valptr[18];
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0;
    int bw = 0, bh = 0;      // blocks per row / column incl. MCU padding
    int dw = 0, dh = 0;      // downsampled size in samples (real image data)
    std::vector<int16_t> coef;  // bw * bh * 64, natural (de-zigzagged) order
    int dc_tab = 0, ac_tab = 0, pred = 0;
};

struct Jpeg {
    const uint8_t* d;
    size_t n, pos = 0;
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1;
    bool progressive = false, have_frame = false;
    Comp comp[4];
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int restart_interval = 0;
    int adobe_transform = -1;
    // bit reader
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    bool hit_marker = false;
    int eobrun = 0;
    const char* err = nullptr;

    int fill() {
        while (bitcnt <= 24) {
            int b = 0;
            if (!hit_marker && pos < n) {
                b = d[pos];
                if (b == 0xff) {
                    int b2 = pos + 1 < n ? d[pos + 1] : 0xd9;
                    if (b2 == 0) pos += 2;              // stuffed zero
                    else { hit_marker = true; b = 0; }  // a marker: feed zeros, leave pos at the 0xff
                } else pos++;
            } else b = 0;
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
        return 0;
    }
    inline int getbits(int k) {
        if (k == 0) return 0;
        if (bitcnt < k) fill();
        int v = (int)(bitbuf >> (32 - k));
        bitbuf <<= k;
        bitcnt -= k;
        return v;
    }
    inline int getbit() { return getbits(1); }
    inline int decode(const Huff& h) {
        if (bitcnt < 16) fill();
        int code = 0;
        for (int l = 1; l <= 16; l++) {
            code = (code << 1) | (int)(bitbuf >> 31);
            bitbuf <<= 1;
            bitcnt--;
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
        err = "bad Huffman code";
        return 0;
    }
    static inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }
    void reset_bits() { bitbuf = 0; bitcnt = 0; hit_marker = false; }

    int u16(size_t at) const { return (d[at] << 8) | d[at + 1]; }

    // ---- one block, sequential (baseline) scan
    void block_seq(Comp& c, int16_t* blk) {
        int t = decode(dc[c.dc_tab]);
        if (t > 11) { err = "bad Huffman code"; return; }   // DC difference categories of 8-bit JPEG: 0..11 (T.81 F.1.2.1.1)
        int diff = t ? extend(getbits(t), t) : 0;
        c.pred += diff;
        blk[0] = (int16_t)c.pred;
        for (int k = 1; k < 64;) {
            int rs = decode(ac[c.ac_tab]);
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r == 15) { k += 16; continue; }
                break;
            }
            k += r;
            if (k > 63) { err = "AC index out of range"; return; }
            blk[kZigzag[k]] = (int16_t)extend(getbits(s), s);
            k++;
        }
    }
    // ---- progressive scans (ITU T.81 G.1.2)
    void block_dc_first(Comp& c, int16_t* blk, int al) {
        int t = decode(dc[c.dc_tab]);
        if (t > 11) { err = "bad Huffman code"; return; }
        int diff = t ? extend(getbits(t), t) : 0;
        c.pred += diff;
        blk[0] = (int16_t)(c.pred * (1 << al));
    }
    void block_dc_refine(int16_t* blk, int al) {
        if (getbit()) blk[0] |= (int16_t)(1 << al);
    }
    void block_ac_first(Comp& c, int16_t* blk, int ss, int se, int al) {
        if (eobrun > 0) { eobrun--; return; }
        for (int k = ss; k <= se;) {
            int rs = decode(ac[c.ac_tab]);
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobrun = (1 << r) - 1;
                    if (r) eobrun += getbits(r);
                    break;
                }
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) { err = "AC index out of range"; return; }
            blk[kZigzag[k]] = (int16_t)(extend(getbits(s), s) * (1 << al));
            k++;
        }
    }
    void block_ac_refine(Comp& c, int16_t* blk, int ss, int se, int al) {
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun <= 0) {
            for (; k <= se;) {
                int rs = decode(ac[c.ac_tab]);
                int r = rs >> 4, s = rs & 15;
                int val = 0;
                if (s == 0) {
                    if (r < 15) {
                        eobrun = (1 << r);
                        if (r) eobrun += getbits(r);
                        break;
                    }
                    // r == 15: skip 16 zero coefficients (refining the non-zero ones passed on the way)
                } else {
                    if (s != 1) { err = "bad refinement code"; return; }
                    val = getbit() ? p1 : m1;
                }
                while (k <= se) {
                    int16_t* cf = &blk[kZigzag[k]];
                    if (*cf != 0) {
                        if (getbit() && (*cf & p1) == 0) *cf = (int16_t)(*cf >= 0 ? *cf + p1 : *cf + m1);
                    } else {
                        if (r == 0) {
                            if (val) *cf = (int16_t)val;
                            k++;
                            break;
                        }
                        r--;
                    }
                    k++;
                }
                if (err) return;
            }
        }
        if (eobrun > 0) {
            for (; k <= se; k++) {
                int16_t* cf = &blk[kZigzag[k]];
                if (*cf != 0 && getbit() && (*cf & p1) == 0) *cf = (int16_t)(*cf >= 0 ? *cf + p1 : *cf + m1);
            }
            eobrun--;
        }
    }

    bool restart_if_due(int& todo) {
        if (!restart_interval) return true;
        if (--todo > 0) return true;
        // expect RSTn at pos (byte aligned)
        reset_bits();
        while (pos + 1 < n && !(d[pos] == 0xff && d[pos + 1] >= 0xd0 && d[pos + 1] <= 0xd7)) {
            if (d[pos] == 0xff && d[pos + 1] != 0 && d[pos + 1] != 0xff) return true;  // some other marker: end of scan data
            pos++;
        }
        if (pos + 1 < n) pos += 2;
        for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
        eobrun = 0;
        todo = restart_interval;
        return true;
    }

    bool scan(size_t at, size_t len) {
        if (len < 1) { err = "bad SOS"; return false; }   // (a file that ends in FF DA 00 02 has at == n here)
        const int ns = d[at];
        if (ns < 1 || ns > 4 || len < (size_t)(1 + 2 * ns + 3)) { err = "bad SOS"; return false; }
        int idx[4];
        for (int i = 0; i < ns; i++) {
            int cid = d[at + 1 + 2 * i], tabs = d[at + 2 + 2 * i];
            int ci = -1;
            for (int j = 0; j < ncomp; j++) if (comp[j].id == cid) ci = j;
            if (ci < 0) { err = "SOS names an unknown component"; return false; }
            idx[i] = ci;
            comp[ci].dc_tab = tabs >> 4; comp[ci].ac_tab = tabs & 15;
            if (comp[ci].dc_tab > 3 || comp[ci].ac_tab > 3) { err = "bad table index"; return false; }
        }
        const int ss = d[at + 1 + 2 * ns], se = d[at + 2 + 2 * ns], ah = d[at + 3 + 2 * ns] >> 4, al = d[at + 3 + 2 * ns] & 15;
        if (progressive) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1)) { err = "bad progressive scan parameters"; return false; }
        }
        for (int i = 0; i < ns; i++) {
            const Comp& c = comp[idx[i]];
            if ((!progressive || ss == 0) && !(progressive && ah) && !dc[c.dc_tab].present) { err = "missing DC Huffman table"; return false; }
            if ((!progressive || ss > 0) && !ac[c.ac_tab].present) { err = "missing AC Huffman table"; return false; }
        }
        pos = at + len;
        reset_bits();
        for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
        eobrun = 0;
        int todo = restart_interval;
        auto one = [&](Comp& c, int bx, int by) {
            int16_t* blk = &c.coef[((size_t)by * c.bw + bx) * 64];
            if (!progressive) block_seq(c, blk);
            else if (ss == 0) { if (ah == 0) block_dc_first(c, blk, al); else block_dc_refine(blk, al); }
            else { if (ah == 0) block_ac_first(c, blk, ss, se, al); else block_ac_refine(c, blk, ss, se, al); }
        };
        if (ns == 1) {  // non-interleaved: the component's own block grid, ceil(size / 8) blocks
            Comp& c = comp[idx[0]];
            const int nbx = (c.dw + 7) / 8, nby = (c.dh + 7) / 8;
            for (int by = 0; by < nby; by++)
                for (int bx = 0; bx < nbx; bx++) {
                    one(c, bx, by);
                    if (err) return false;
                    restart_if_due(todo);
                }
        } else {
            const int mx = (W + 8 * hmax - 1) / (8 * hmax), my = (H + 8 * vmax - 1) / (8 * vmax);
            for (int m = 0; m < mx * my; m++) {
                const int mcx = m % mx, mcy = m / mx;
                for (int i = 0; i < ns; i++) {
                    Comp& c = comp[idx[i]];
                    for (int y = 0; y < c.v; y++)
                        for (int x = 0; x < c.h; x++) {
                            one(c, mcx * c.h + x, mcy * c.v + y);
                            if (err) return false;
                        }
                }
                restart_if_due(todo);
            }
        }
        // leave pos at the next marker
        if (!hit_marker) {
            while (pos + 1 < n && !(d[pos] == 0xff && d[pos + 1] != 0 && !(d[pos + 1] >= 0xd0 && d[pos + 1] <= 0xd7))) pos++;
        }
        return true;
    }

    bool parse() {
        if (n < 4 || d[0] != 0xff || d[1] != 0xd8) { err = "not a JPEG file"; return false; }
        pos = 2;
        bool seen_scan = false;
        while (pos + 3 < n) {
            if (d[pos] != 0xff) { pos++; continue; }
            const int m = d[pos + 1];
            if (m == 0xff) { pos++; continue; }
            if (m == 0xd9) break;                                   // EOI
            if (m == 0x01 || (m >= 0xd0 && m <= 0xd7) || m == 0) { pos += 2; continue; }
            const size_t len = (size_t)u16(pos + 2);
            if (len < 2 || pos + 2 + len > n) { err = "truncated segment"; return false; }
            const size_t at = pos + 4, body = len - 2;
            if (m == 0xdb) {  // DQT
                size_t q = at;
                while (q < at + body) {
                    const int pq = d[q] >> 4, tq = d[q] & 15;
                    q++;
                    if (tq > 3 || q + (pq ? 128 : 64) > at + body) { err = "bad DQT"; return false; }
                    for (int i = 0; i < 64; i++) { qt[tq][kZigzag[i]] = pq ? (uint16_t)u16(q + 2 * i) : d[q + i]; }
                    qt_present[tq] = true;
                    q += pq ? 128 : 64;
                }
            } else if (m == 0xc4) {  // DHT
                size_t q = at;
                while (q < at + body) {
                    const int tc = d[q] >> 4, th = d[q] & 15;
                    q++;
                    if (tc > 1 || th > 3 || q + 16 > at + body) { err = "bad DHT"; return false; }
                    Huff& h = tc ? ac[th] : dc[th];
                    int total = 0;
                    h.bits[0] = 0;
                    for (int i = 1; i <= 16; i++) { h.bits[i] = d[q + i - 1]; total += h.bits[i]; }
                    q += 16;
                    if (total > 256 || q + total > at + body) { err = "bad DHT"; return false; }
                    memcpy(h.vals, d + q, (size_t)total);
                    q += total;
                    h.build();
                    h.present = true;
                }
            } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {  // SOF0/1/2
                if (have_frame) { err = "more than one frame"; return false; }
                if (body < 6 || d[at] != 8) { err = "only 8-bit JPEG is supported"; return false; }
                progressive = m == 0xc2;
                H = u16(at + 1); W = u16(at + 3); ncomp = d[at + 5];
                if (W < 1 || H < 1 || (ncomp != 1 && ncomp != 3) || body < (size_t)(6 + 3 * ncomp)) { err = "unsupported frame (need 1 or 3 components)"; return false; }
                if ((uint64_t)W * (uint64_t)H > (1ull << 28)) { err = "image too large (more than 2^28 pixels)"; return false; }
                for (int i = 0; i < ncomp; i++) {
                    Comp& c = comp[i];
                    c.id = d[at + 6 + 3 * i]; c.h = d[at + 7 + 3 * i] >> 4; c.v = d[at + 7 + 3 * i] & 15; c.tq = d[at + 8 + 3 * i];
                    if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) { err = "bad sampling factors"; return false; }
                    if (c.h > hmax) hmax = c.h;
                    if (c.v > vmax) vmax = c.v;
                }
                const int mx = (W + 8 * hmax - 1) / (8 * hmax), my = (H + 8 * vmax - 1) / (8 * vmax);
                for (int i = 0; i < ncomp; i++) {
                    Comp& c = comp[i];
                    c.bw = mx * c.h; c.bh = my * c.v;
                    c.dw = (W * c.h + hmax - 1) / hmax; c.dh = (H * c.v + vmax - 1) / vmax;
                    c.coef.assign((size_t)c.bw * c.bh * 64, 0);
                }
                have_frame = true;
            } else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) {
                err = "unsupported JPEG process (lossless / hierarchical / arithmetic)";
                return false;
            } else if (m == 0xdd) {
                if (body >= 2) restart_interval = u16(at);
            } else if (m == 0xee && body >= 12 && memcmp(d + at, "Adobe", 5) == 0) {
                adobe_transform = d[at + 11];
            } else if (m == 0xda) {
                if (!have_frame) { err = "scan before frame"; return false; }
                if (!scan(at, body)) return false;
                seen_scan = true;
                continue;  // scan() left pos at the next marker
            }
            pos += 2 + len;
        }
        if (!seen_scan) { err = "no scan"; return false; }
        return true;
    }
};

// jidctint.c "islow": 8x8 inverse DCT, 13-bit constants, two passes, output range-limited around +128
inline uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr int F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
                  F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    auto descale = [](long x, int nb) { return (int)((x + (1L << (nb - 1))) >> nb); };
    int ws[64];
    for (int c = 0; c < 8; c++) {
        const int16_t* ip = in + c;
        const uint16_t* qp = q + c;
        int* wp = ws + c;
        if (ip[8] == 0 && ip[16] == 0 && ip[24] == 0 && ip[32] == 0 && ip[40] == 0 && ip[48] == 0 && ip[56] == 0) {
            const int dcval = (int)((long)ip[0] * qp[0]) * (1 << P1);
            for (int r = 0; r < 8; r++) wp[8 * r] = dcval;
            continue;
        }
        long z2 = (long)ip[16] * qp[16], z3 = (long)ip[48] * qp[48];
        long z1 = (z2 + z3) * F0_541;
        long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        z2 = (long)ip[0] * qp[0]; z3 = (long)ip[32] * qp[32];
        long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = (long)ip[56] * qp[56]; tmp1 = (long)ip[40] * qp[40]; tmp2 = (long)ip[24] * qp[24]; tmp3 = (long)ip[8] * qp[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3, z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        wp[0] = descale(tmp10 + tmp3, CB - P1); wp[56] = descale(tmp10 - tmp3, CB - P1);
        wp[8] = descale(tmp11 + tmp2, CB - P1); wp[48] = descale(tmp11 - tmp2, CB - P1);
        wp[16] = descale(tmp12 + tmp1, CB - P1); wp[40] = descale(tmp12 - tmp1, CB - P1);
        wp[24] = descale(tmp13 + tmp0, CB - P1); wp[32] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; r++) {
        const int* wp = ws + 8 * r;
        uint8_t* op = out + (size_t)r * stride;
        long z2 = wp[2], z3 = wp[6];
        long z1 = (z2 + z3) * F0_541;
        long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        long tmp0 = ((long)wp[0] + wp[4]) * (1L << CB), tmp1 = ((long)wp[0] - wp[4]) * (1L << CB);
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = wp[7]; tmp1 = wp[5]; tmp2 = wp[3]; tmp3 = wp[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3, z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        constexpr int S = CB + P1 + 3;
        op[0] = clamp255(descale(tmp10 + tmp3, S) + 128); op[7] = clamp255(descale(tmp10 - tmp3, S) + 128);
        op[1] = clamp255(descale(tmp11 + tmp2, S) + 128); op[6] = clamp255(descale(tmp11 - tmp2, S) + 128);
        op[2] = clamp255(descale(tmp12 + tmp1, S) + 128); op[5] = clamp255(descale(tmp12 - tmp1, S) + 128);
        op[3] = clamp255(descale(tmp13 + tmp0, S) + 128); op[4] = clamp255(descale(tmp13 - tmp0, S) + 128);
    }
}

// component plane (dw x dh real samples inside a bw*8-wide buffer) -> full-resolution plane W x H (jdsample.c)
void upsample(const Comp& c, const std::vector<uint8_t>& src, int src_stride, int hmax, int vmax, int W, int H, std::vector<uint8_t>& dst) {
    dst.assign((size_t)W * H, 0);
    const int hx = hmax / c.h, vx = vmax / c.v;
    const bool exact = hmax % c.h == 0 && vmax % c.v == 0;
    auto row = [&](int y) { return &src[(size_t)(y < 0 ? 0 : (y >= c.dh ? c.dh - 1 : y)) * src_stride]; };
    if (exact && hx == 1 && vx == 1) {
        for (int y = 0; y < H; y++) memcpy(&dst[(size_t)y * W], row(y), (size_t)W);
        return;
    }
    std::vector<uint8_t> line((size_t)c.dw * 2 + 2);
    if (exact && hx == 2 && vx == 1 && c.dw > 2) {  // h2v1_fancy_upsample
        for (int y = 0; y < H; y++) {
            const uint8_t* in = row(y);
            line[0] = in[0];
            line[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
            for (int x = 1; x < c.dw - 1; x++) {
                const int v = in[x] * 3;
                line[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2);
                line[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2);
            }
            const int x = c.dw - 1;
            line[2 * x] = (uint8_t)((in[x] * 3 + in[x - 1] + 1) >> 2);
            line[2 * x + 1] = in[x];
            memcpy(&dst[(size_t)y * W], line.data(), (size_t)W);
        }
        return;
    }
    if (exact && hx == 2 && vx == 2 && c.dw > 2) {  // h2v2_fancy_upsample: 3/4 nearer row + 1/4 further, then the same horizontally
        for (int y = 0; y < H; y++) {
            const int sy = y >> 1;
            const uint8_t* in0 = row(sy);
            const uint8_t* in1 = (y & 1) ? row(sy + 1) : row(sy - 1);
            int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
            line[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
            line[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol; thiscol = nextcol;
            for (int x = 1; x < c.dw - 1; x++) {
                nextcol = in0[x + 1] * 3 + in1[x + 1];
                line[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                line[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                lastcol = thiscol; thiscol = nextcol;
            }
            const int x = c.dw - 1;
            line[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
            line[2 * x + 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
            memcpy(&dst[(size_t)y * W], line.data(), (size_t)W);
        }
        return;
    }
    // everything else: replication (int_upsample); non-integral ratios fall back to nearest
    for (int y = 0; y < H; y++) {
        const uint8_t* in = row(exact ? y / vx : (int)((long)y * c.v / vmax));
        uint8_t* o = &dst[(size_t)y * W];
        for (int x = 0; x < W; x++) {
            int sx = exact ? x / hx : (int)((long)x * c.h / hmax);
            o[x] = in[sx >= c.dw ? c.dw - 1 : sx];
        }
    }
}

int decode_jpeg(const std::vector<uint8_t>& d, uint32_t* w, uint32_t* h, uint8_t** rgb) {
    Jpeg j;
    j.d = d.data(); j.n = d.size();
    if (!j.parse() || j.err) { cray::set_last_error("JPEG: %s", j.err ? j.err : "decode error"); return CRAY_ERR_INVALID; }
    const int W = j.W, H = j.H;
    std::vector<uint8_t> full[3];
    for (int i = 0; i < j.ncomp; i++) {
        Comp& c = j.comp[i];
        if (!j.qt_present[c.tq]) { cray::set_last_error("JPEG: missing quantisation table"); return CRAY_ERR_INVALID; }
        const int stride = c.bw * 8;
        std::vector<uint8_t> plane((size_t)stride * c.bh * 8);
        for (int by = 0; by < c.bh; by++)
            for (int bx = 0; bx < c.bw; bx++)
                idct_islow(&c.coef[((size_t)by * c.bw + bx) * 64], j.qt[c.tq], &plane[(size_t)by * 8 * stride + (size_t)bx * 8], stride);
        upsample(c, plane, stride, j.hmax, j.vmax, W, H, full[i]);
        c.coef.clear(); c.coef.shrink_to_fit();
    }
    uint8_t* out = (uint8_t*)malloc((size_t)W * H * 3);
    if (!out) { cray::set_last_error("JPEG: out of memory"); return CRAY_ERR_INVALID; }
    const size_t n = (size_t)W * H;
    if (j.ncomp == 1) {
        for (size_t i = 0; i < n; i++) { out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = full[0][i]; }
    } else if (j.adobe_transform == 0) {  // Adobe marker says the components are R, G, B
        for (size_t i = 0; i < n; i++) { out[3 * i] = full[0][i]; out[3 * i + 1] = full[1][i]; out[3 * i + 2] = full[2][i]; }
    } else {  // jdcolor.c ycc_rgb_convert with its 16-bit fixed-point tables
        int cr_r[256], cb_b[256];
        long cr_g[256], cb_g[256];
        for (int i = 0; i < 256; i++) {
            const long x = i - 128;
            cr_r[i] = (int)((91881L * x + 32768L) >> 16);    // FIX(1.40200)
            cb_b[i] = (int)((116130L * x + 32768L) >> 16);   // FIX(1.77200)
            cr_g[i] = -46802L * x;                            // FIX(0.71414)
            cb_g[i] = -22554L * x + 32768L;                   // FIX(0.34414) + ONE_HALF
        }
        for (size_t i = 0; i < n; i++) {
            const int y = full[0][i], cb = full[1][i], cr = full[2][i];
            out[3 * i] = clamp255(y + cr_r[cr]);
            out[3 * i + 1] = clamp255(y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
            out[3 * i + 2] = clamp255(y + cb_b[cb]);
        }
    }
    *w = (uint32_t)W; *h = (uint32_t)H; *rgb = out;
    return CRAY_OK;
}

}  // namespace

extern "C" int cray_load_image(const char* path, uint32_t* width, uint32_t* height, uint8_t** rgb8) {
    if (!path || !width || !height || !rgb8) { cray::set_last_error("cray_load_image: null argument"); return CRAY_ERR_INVALID; }
    *rgb8 = nullptr;
    // Nothing may leave an extern "C" function by exception: a header that declares 65535 x 65535 pixels makes the decoders'
    // vectors throw std::bad_alloc, which would reach std::terminate in a C, Rust or ctypes host.
    try {
        std::vector<uint8_t> d;
        if (!read_whole_file(path, d)) { cray::set_last_error("cray_load_image: cannot read %s", path); return CRAY_ERR_INVALID; }
        if (d.size() >= 8 && d[0] == 'P' && (d[1] == '2' || d[1] == '3' || d[1] == '5' || d[1] == '6')) return decode_pnm(d, width, height, rgb8);
        if (d.size() >= 4 && d[0] == 0xff && d[1] == 0xd8) return decode_jpeg(d, width, height, rgb8);
        static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
        if (d.size() >= 8 && memcmp(d.data(), png_sig, 8) == 0) return decode_png(d, width, height, rgb8);
        cray::set_last_error("cray_load_image: %s is not PNM, PNG or JPEG (other formats need a caller-supplied cray_image_loader)", path);
        return CRAY_ERR_UNSUPPORTED;
    } catch (const std::bad_alloc&) {
        if (*rgb8) { free(*rgb8); *rgb8 = nullptr; }
        cray::set_last_error("cray_load_image: %s needs more memory than is available (corrupt header?)", path);
        return CRAY_ERR_INVALID;
    } catch (const std::exception& ex) {
        if (*rgb8) { free(*rgb8); *rgb8 = nullptr; }
        cray::set_last_error("cray_load_image: %s: %s", path, ex.what());
        return CRAY_ERR_INVALID;
    }
}

extern "C" void cray_free_image(uint8_t* rgb8) { free(rgb8); }

// cray_image_loader (cray_cry.h) over cray_load_image: what cray_cry_parse_scene uses when the caller passes no loader
extern "C" int cray_default_image_loader(const char* path, void* user, uint32_t* width, uint32_t* height, uint8_t** rgb8) {
    (void)user;
    return cray_load_image(path, width, height, rgb8) == CRAY_OK ? 0 : 1;
}
