// cray_cry.cpp — `.cry` scene reader and OBJ/MTL ingest (include/cray_cry.h), host side only.
//
// Restates, with the same grammar, defaults, error messages and locations:
//   tokenizer::tokenize                 src/scene_parser.rs:174-253 (+ tokenize_number :132-159, tokenize_string :161-172)
//   parser::{RawValue, RawValueMap, TypedRawValueMap, RawValueArray}   src/scene_parser.rs:320-586
//   scene_parser::{TryFrom impls, create_primitives, parse_scene}      src/scene_parser.rs:799-1117
//   obj::load_obj                       src/obj.rs:26-220 (on top of tobj 4.0.0's GPU_LOAD_OPTIONS behaviour)
//   Material::new_* / Texture::is_black/is_zero                        src/material.rs:19-70, src/texture.rs:83-101
// The output is the argument list of Scene::new as a cray_scene_desc.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/cray_cry.h"
#include "../../include/cray_io.h"

namespace {

struct Loc { uint32_t line = 0, column = 0; };
struct Err {
    std::string message;
    bool has_loc = false;
    Loc loc;
};
static Err err_at(const std::string& m, Loc l) { Err e; e.message = m; e.has_loc = true; e.loc = l; return e; }
static Err err_noloc(const std::string& m) { Err e; e.message = m; return e; }
static std::string fmt(const char* f, ...) {
    char buf[1024];
    va_list ap; va_start(ap, f); vsnprintf(buf, sizeof(buf), f, ap); va_end(ap);
    return buf;
}
// Rust `{}` of an f64: shortest digits that round-trip, never scientific
static std::string rust_f64(double v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[64];
    int prec = 1;
    for (; prec <= 17; prec++) {
        snprintf(buf, sizeof(buf), "%.*e", prec - 1, v);
        if (strtod(buf, nullptr) == v) break;
    }
    // digits and exponent -> plain decimal
    std::string s(buf);
    size_t epos = s.find('e');
    int exp10 = atoi(s.c_str() + epos + 1);
    std::string mant = s.substr(0, epos);
    bool neg = mant[0] == '-';
    if (neg) mant = mant.substr(1);
    std::string digits;
    for (char c : mant) if (c != '.') digits.push_back(c);
    std::string out;
    int point = exp10 + 1;  // position of the decimal point within digits
    if (point <= 0) { out = "0." + std::string((size_t)(-point), '0') + digits; }
    else if ((size_t)point >= digits.size()) { out = digits + std::string((size_t)point - digits.size(), '0'); }
    else { out = digits.substr(0, (size_t)point) + "." + digits.substr((size_t)point); }
    if (out.find('.') != std::string::npos) {
        while (out.back() == '0') out.pop_back();
        if (out.back() == '.') out.pop_back();
    }
    return (neg ? "-" : "") + out;
}

// ------------------------------------------------------------------ tokenizer
enum Tok { T_IDENT = 0, T_NUMBER, T_STRING, T_LBRACE, T_RBRACE, T_LBRACKET, T_RBRACKET, T_LPAREN, T_RPAREN, T_COMMA, T_COLON, T_EOF };
struct Token {
    Tok kind;
    Loc loc;
    double number = 0.0;
    std::string text;
};
static std::string tok_display(const Token& t) {  // impl Display for TokenValue (:39-56)
    switch (t.kind) {
    case T_IDENT: return "'" + t.text + "'";
    case T_NUMBER: return "'" + rust_f64(t.number) + "'";
    case T_STRING: return "'" + t.text + "'";
    case T_LBRACE: return "'{'";
    case T_RBRACE: return "'}'";
    case T_LBRACKET: return "'['";
    case T_RBRACKET: return "']'";
    case T_LPAREN: return "'('";
    case T_RPAREN: return "')'";
    case T_COMMA: return "','";
    case T_COLON: return "':'";
    default: return "EOF";
    }
}
static const char* tok_kind_display(Tok k) {
    static const char* names[] = {"''", "'0'", "''", "'{'", "'}'", "'['", "']'", "'('", "')'", "','", "':'", "EOF"};
    return names[k];
}

// CharsWithLocation (:86-130): iterates Unicode scalar values, column counts characters
struct Chars {
    const std::string& s;
    size_t pos = 0;
    Loc loc;
    explicit Chars(const std::string& str) : s(str) { loc.line = 1; loc.column = 1; }
    static size_t len_at(unsigned char c) { return c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 1; }
    bool peek(std::string* ch) const {
        if (pos >= s.size()) return false;
        size_t n = len_at((unsigned char)s[pos]);
        if (pos + n > s.size()) n = s.size() - pos;
        *ch = s.substr(pos, n);
        return true;
    }
    bool next(std::string* ch) {
        std::string c;
        if (!peek(&c)) return false;
        pos += c.size();
        if (c == "\n") { loc.line += 1; loc.column = 1; } else loc.column += 1;
        if (ch) *ch = c;
        return true;
    }
};

static bool tokenize(const std::string& input, std::vector<Token>& out, Err& err) {
    Chars ch(input);
    std::string c;
    while (ch.peek(&c)) {
        const char c0 = c.size() == 1 ? c[0] : '\0';
        bool single = false;
        Tok kind = T_EOF;
        switch (c0) {
        case ' ': case '\t': case '\n': case '\r': break;
        case '/': {
            ch.next(nullptr);
            std::string d;
            if (ch.peek(&d) && d == "/") {
                ch.next(nullptr);
                while (ch.peek(&d)) {
                    if (d == "\n") break;
                    ch.next(nullptr);
                }
            } else {
                err = err_at("Expected a second '/' to start a comment", ch.loc);
                return false;
            }
            break;
        }
        case '{': single = true; kind = T_LBRACE; break;
        case '}': single = true; kind = T_RBRACE; break;
        case '[': single = true; kind = T_LBRACKET; break;
        case ']': single = true; kind = T_RBRACKET; break;
        case '(': single = true; kind = T_LPAREN; break;
        case ')': single = true; kind = T_RPAREN; break;
        case ',': single = true; kind = T_COMMA; break;
        case ':': single = true; kind = T_COLON; break;
        case '"': case '\'': {  // tokenize_string
            Token t; t.kind = T_STRING; t.loc = ch.loc;
            std::string start, d;
            ch.next(&start);
            bool closed = false;
            while (ch.next(&d)) {
                if (d == start) { closed = true; break; }
                t.text += d;
            }
            if (!closed) { err = err_at("Unterminated string", t.loc); return false; }
            out.push_back(t);
            continue;
        }
        default:
            if ((c0 >= '0' && c0 <= '9') || c0 == '+' || c0 == '-') {  // tokenize_number
                Token t; t.kind = T_NUMBER; t.loc = ch.loc;
                std::string num, d;
                bool has_dot = false;
                if (c0 == '+' || c0 == '-') { ch.next(&d); num += d; }
                while (ch.peek(&d)) {
                    if (d.size() == 1 && d[0] >= '0' && d[0] <= '9') { ch.next(nullptr); num += d; }
                    else if (!has_dot && d == ".") { has_dot = true; ch.next(nullptr); num += d; }
                    else break;
                }
                bool any_digit = false;
                for (char x : num) if (x >= '0' && x <= '9') any_digit = true;
                if (!any_digit) { err = err_at("Cannot parse '" + num + "' as number", t.loc); return false; }
                t.number = strtod(num.c_str(), nullptr);
                out.push_back(t);
                continue;
            }
            if ((c0 >= 'a' && c0 <= 'z') || (c0 >= 'A' && c0 <= 'Z') || c0 == '_') {
                Token t; t.kind = T_IDENT; t.loc = ch.loc;
                std::string d;
                ch.next(&d); t.text += d;
                while (ch.peek(&d)) {
                    char x = d.size() == 1 ? d[0] : '\0';
                    if ((x >= 'a' && x <= 'z') || (x >= 'A' && x <= 'Z') || (x >= '0' && x <= '9') || x == '_') { ch.next(nullptr); t.text += d; }
                    else break;
                }
                out.push_back(t);
                continue;
            }
            err = err_at("Unexpected character: '" + c + "'", ch.loc);
            return false;
        }
        if (single) { Token t; t.kind = kind; t.loc = ch.loc; out.push_back(t); }
        ch.next(nullptr);
    }
    Token e; e.kind = T_EOF; e.loc = ch.loc;
    out.push_back(e);
    return true;
}

// ------------------------------------------------------------------ raw values
struct RawValue;
using RawPtr = std::unique_ptr<RawValue>;
struct RawMap {
    Loc loc;
    std::map<std::string, RawPtr> map;  // HashMap in the reference; only lookups by key matter
};
enum RawKind { R_NUMBER, R_STRING, R_VECTOR, R_POINT, R_COLOR, R_MAP, R_TYPED, R_ARRAY };
struct RawValue {
    RawKind kind = R_NUMBER;
    double number = 0.0;
    std::string text;        // String / TypedMap name
    double v[3] = {0, 0, 0}; // Vector / Point / Color
    RawMap map;              // Map / TypedMap
    std::set<std::string> used_keys;  // TypedRawValueMap::used_keys
    std::vector<RawPtr> array;
};
static const char* raw_kind_name(const RawValue& v) {
    switch (v.kind) {
    case R_NUMBER: return "Number"; case R_STRING: return "String"; case R_VECTOR: return "Vector"; case R_POINT: return "Point";
    case R_COLOR: return "Color"; case R_MAP: return "Map"; case R_TYPED: return "TypedMap"; default: return "Array";
    }
}

struct TokenStream {
    const std::vector<Token>& t;
    size_t i = 0;
    const Token& peek() const { return t[i < t.size() ? i : t.size() - 1]; }
    const Token& next() { const Token& r = peek(); if (i < t.size() - 1) i++; else i = t.size(); return r; }
};

static bool expect_variant(TokenStream& ts, Tok expected, Token* got, Err& err) {  // :276-296
    const Token& tk = ts.next();
    if (tk.kind == expected) { if (got) *got = tk; return true; }
    err = err_at(std::string("Expected ") + tok_kind_display(expected) + ", got " + tok_display(tk), tk.loc);
    return false;
}
static bool expect_number(TokenStream& ts, double* out, Err& err) {  // :298-308
    const Token& tk = ts.next();
    if (tk.kind == T_NUMBER) { *out = tk.number; return true; }
    err = err_at("Expected number, got " + tok_display(tk), tk.loc);
    return false;
}
static bool parse_value(TokenStream& ts, RawValue& out, Err& err);

static bool parse_map(TokenStream& ts, RawMap& out, Err& err) {  // RawValueMap::from_tokens :366-413
    Token start;
    if (!expect_variant(ts, T_LBRACE, &start, err)) return false;
    out.loc = start.loc;
    for (;;) {
        const Token& tk = ts.peek();
        if (tk.kind == T_RPAREN) break;  // sic
        if (tk.kind == T_IDENT) {
            std::string key = tk.text;
            ts.next();
            if (!expect_variant(ts, T_COLON, nullptr, err)) return false;
            RawPtr v(new RawValue());
            if (!parse_value(ts, *v, err)) return false;
            if (out.map.count(key)) { err = err_at("Duplicate key " + key, out.loc); return false; }
            out.map[key] = std::move(v);
        } else break;
        const Token& sep = ts.peek();
        if (sep.kind == T_RBRACE) break;
        if (sep.kind == T_COMMA) ts.next(); else break;
    }
    return expect_variant(ts, T_RBRACE, nullptr, err);
}
static bool parse_triple(TokenStream& ts, double v[3], Err& err) {
    return expect_variant(ts, T_LPAREN, nullptr, err) && expect_number(ts, &v[0], err) && expect_variant(ts, T_COMMA, nullptr, err) &&
           expect_number(ts, &v[1], err) && expect_variant(ts, T_COMMA, nullptr, err) && expect_number(ts, &v[2], err) &&
           expect_variant(ts, T_RPAREN, nullptr, err);
}
static bool parse_value(TokenStream& ts, RawValue& out, Err& err) {  // RawValue::from_tokens :333-402
    const Token tk = ts.peek();
    switch (tk.kind) {
    case T_NUMBER: ts.next(); out.kind = R_NUMBER; out.number = tk.number; return true;
    case T_STRING: ts.next(); out.kind = R_STRING; out.text = tk.text; return true;
    case T_IDENT: {
        ts.next();
        const Token opener = ts.peek();
        if (opener.kind == T_LPAREN) {
            if (tk.text == "Vector") { out.kind = R_VECTOR; return parse_triple(ts, out.v, err); }
            if (tk.text == "Point") { out.kind = R_POINT; return parse_triple(ts, out.v, err); }
            if (tk.text == "Color") { out.kind = R_COLOR; return parse_triple(ts, out.v, err); }
            // TypedRawValueMap::from_tokens: expects an identifier, but it was already consumed
            const Token& n = ts.next();
            if (n.kind != T_IDENT) { err = err_at("Expected identifier, got " + tok_display(n), n.loc); return false; }
            out.kind = R_TYPED; out.text = n.text;
            return parse_map(ts, out.map, err);
        }
        if (opener.kind == T_LBRACE) { out.kind = R_TYPED; out.text = tk.text; return parse_map(ts, out.map, err); }
        err = err_at("Expected '(' or '{', got " + tok_display(opener), opener.loc);
        return false;
    }
    case T_LBRACE: out.kind = R_MAP; return parse_map(ts, out.map, err);
    case T_LBRACKET: {  // RawValueArray::from_tokens :594-626
        out.kind = R_ARRAY;
        if (!expect_variant(ts, T_LBRACKET, nullptr, err)) return false;
        for (;;) {
            if (ts.peek().kind == T_RBRACKET) break;
            RawPtr v(new RawValue());
            if (!parse_value(ts, *v, err)) return false;
            out.array.push_back(std::move(v));
            const Token& sep = ts.peek();
            if (sep.kind == T_RBRACKET) break;
            if (sep.kind == T_COMMA) ts.next(); else break;
        }
        return expect_variant(ts, T_RBRACKET, nullptr, err);
    }
    default:
        err = err_at("Expected a raw value. Got " + tok_display(tk), tk.loc);
        return false;
    }
}

static void dump_value(const RawValue& v, std::string& o);
static void dump_map(const RawMap& m, std::string& o) {
    o += fmt("@%u:%u{", m.loc.line, m.loc.column);
    bool first = true;
    for (auto& kv : m.map) {
        if (!first) o += ",";
        first = false;
        o += kv.first + ":";
        dump_value(*kv.second, o);
    }
    o += "}";
}
static void dump_value(const RawValue& v, std::string& o) {
    switch (v.kind) {
    case R_NUMBER: o += "Number(" + rust_f64(v.number) + ")"; break;
    case R_STRING: o += "String(\"" + v.text + "\")"; break;
    case R_VECTOR: o += "Vector(" + rust_f64(v.v[0]) + "," + rust_f64(v.v[1]) + "," + rust_f64(v.v[2]) + ")"; break;
    case R_POINT: o += "Point(" + rust_f64(v.v[0]) + "," + rust_f64(v.v[1]) + "," + rust_f64(v.v[2]) + ")"; break;
    case R_COLOR: o += "Color(" + rust_f64(v.v[0]) + "," + rust_f64(v.v[1]) + "," + rust_f64(v.v[2]) + ")"; break;
    case R_MAP: o += "Map"; dump_map(v.map, o); break;
    case R_TYPED: o += "Typed:" + v.text; dump_map(v.map, o); break;
    case R_ARRAY:
        o += "Array[";
        for (size_t i = 0; i < v.array.size(); i++) { if (i) o += ","; dump_value(*v.array[i], o); }
        o += "]";
        break;
    }
}

// ------------------------------------------------------------------ typed access (RawValueMap::get / get_or :415-459)
// `conv` converts the raw value; its error is wrapped with the key name, keeping the inner location if any.
template <class F>
static bool map_get(RawMap& m, std::set<std::string>* used, const std::string& key, bool required, bool* present, F conv, Err& err) {
    if (used) used->insert(key);
    auto it = m.map.find(key);
    if (it == m.map.end()) {
        if (present) *present = false;
        if (!required) return true;
        err = err_at(key + " not found in map", m.loc);
        return false;
    }
    if (present) *present = true;
    Err inner;
    if (conv(*it->second, inner)) return true;
    err = err_at("Error converting map value for '" + key + "' to expected type: " + inner.message, inner.has_loc ? inner.loc : m.loc);
    return false;
}
static bool as_number(RawValue& v, double* out, Err& e) {
    if (v.kind == R_NUMBER) { *out = v.number; return true; }
    e = err_noloc(std::string("Cannot get Number, found ") + raw_kind_name(v));
    return false;
}
static uint64_t as_usize_sat(double x) { return !(x > 0.0) ? 0 : (x >= 18446744073709551616.0 ? ~0ull : (uint64_t)x); }
static bool as_string(RawValue& v, std::string* out, Err& e) {
    if (v.kind == R_STRING) { *out = v.text; return true; }
    e = err_noloc(std::string("Cannot get String, found ") + raw_kind_name(v));
    return false;
}
static bool as_triple(RawValue& v, RawKind want, const char* name, double out[3], Err& e) {
    if (v.kind == want) { out[0] = v.v[0]; out[1] = v.v[1]; out[2] = v.v[2]; return true; }
    e = err_noloc(std::string("Cannot get ") + name + ", found " + raw_kind_name(v));
    return false;
}

// ------------------------------------------------------------------ scene under construction
struct Builder {
    std::vector<cray_sphere_desc> spheres;
    std::vector<cray_disk_desc> disks;
    std::vector<cray_triangle> triangles;
    std::vector<cray_prim> prims;
    std::vector<cray_light> lights;
    std::vector<cray_material> materials;
    std::vector<cray_bxdf> bxdfs;
    std::vector<cray_texture> textures;
    std::vector<cray_image> images;
    std::vector<uint8_t> pool;
    std::unordered_map<std::string, int32_t> image_ids;
    cray_camera_desc camera;
    uint32_t max_depth = 8, num_samples = 4;
    uint32_t warnings = 0;
    std::string base_dir;
    cray_image_loader loader = nullptr;
    void* loader_user = nullptr;
};

struct Tex {  // Texture<T>
    int kind = CRAY_TEX_CONSTANT;
    double a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
    double scale = 1.0;
    int32_t image = -1;
    bool scalar = false;
};
static bool tex_is_black(const Tex& t) {  // texture.rs:83-91
    auto blk = [](const double* c) { return c[0] == 0.0 && c[1] == 0.0 && c[2] == 0.0; };
    if (t.kind == CRAY_TEX_CONSTANT) return blk(t.a);
    if (t.kind == CRAY_TEX_CHECKERBOARD) return blk(t.a) && blk(t.b);
    return false;
}
static bool tex_is_zero(const Tex& t) {  // texture.rs:93-101
    if (t.kind == CRAY_TEX_CONSTANT) return t.a[0] == 0.0;
    if (t.kind == CRAY_TEX_CHECKERBOARD) return t.a[0] == 0.0 && t.b[0] == 0.0;
    return false;
}
static int32_t add_texture(Builder& b, const Tex& t) {
    cray_texture c;
    memset(&c, 0, sizeof(c));
    c.kind = t.kind; c.image = t.image;
    c.a = {t.a[0], t.a[1], t.a[2]}; c.b = {t.b[0], t.b[1], t.b[2]};
    c.scale = t.scale;
    b.textures.push_back(c);
    return (int32_t)b.textures.size() - 1;
}
static cray_bxdf mk_bxdf(int kind, int32_t ta, int32_t tb) {
    cray_bxdf x;
    memset(&x, 0, sizeof(x));
    x.kind = kind; x.tex_a = ta; x.tex_b = tb;
    return x;
}
static int32_t add_material(Builder& b, bool is_bsdf, const std::vector<cray_bxdf>& lobes) {
    cray_material m;
    m.is_bsdf = is_bsdf ? 1 : 0; m.n_bxdfs = (int32_t)lobes.size(); m.first_bxdf = (int32_t)b.bxdfs.size(); m.pad_ = 0;
    for (auto& l : lobes) b.bxdfs.push_back(l);
    b.materials.push_back(m);
    return (int32_t)b.materials.size() - 1;
}
// Material::new_* (material.rs:19-70)
static int32_t new_matte(Builder& b, const Tex& reflectance, const Tex& sigma) {
    if (tex_is_zero(sigma)) return add_material(b, false, {mk_bxdf(CRAY_BXDF_LAMBERTIAN, add_texture(b, reflectance), -1)});
    return add_material(b, false, {mk_bxdf(CRAY_BXDF_OREN_NAYAR, add_texture(b, reflectance), add_texture(b, sigma))});
}
static int32_t new_glass(Builder& b, const Tex& reflectance, const Tex& transmittance, double eta) {
    cray_bxdf x = mk_bxdf(CRAY_BXDF_FRESNEL_SPECULAR, add_texture(b, reflectance), add_texture(b, transmittance));
    x.eta_i = 1.0; x.eta_t = eta;
    return add_material(b, false, {x});
}
static int32_t new_plastic(Builder& b, const Tex& diffuse, const Tex& specular, const Tex& roughness) {
    std::vector<cray_bxdf> lobes;
    if (!tex_is_black(diffuse)) {
        if (!tex_is_zero(roughness)) lobes.push_back(mk_bxdf(CRAY_BXDF_OREN_NAYAR, add_texture(b, diffuse), add_texture(b, roughness)));
        else lobes.push_back(mk_bxdf(CRAY_BXDF_LAMBERTIAN, add_texture(b, diffuse), -1));
    }
    if (!tex_is_black(specular)) {
        cray_bxdf x = mk_bxdf(CRAY_BXDF_SPECULAR_BRDF, add_texture(b, specular), -1);
        x.fresnel_kind = CRAY_FRESNEL_DIELECTRIC; x.eta_i = 1.0; x.eta_t = 1.5;
        lobes.push_back(x);
    }
    return add_material(b, true, lobes);
}
static int32_t new_metal(Builder& b, const Tex& eta, const Tex& k) {
    return add_material(b, true, {mk_bxdf(CRAY_BXDF_FRESNEL_CONDUCTOR, add_texture(b, eta), add_texture(b, k))});
}

// Texture<T> from a raw value (scene_parser.rs:904-939)
static bool as_texture(RawValue& v, bool scalar, Tex* out, Err& e) {
    out->scalar = scalar;
    if (scalar && v.kind == R_NUMBER) { out->kind = CRAY_TEX_CONSTANT; out->a[0] = v.number; return true; }
    if (!scalar && v.kind == R_COLOR) { out->kind = CRAY_TEX_CONSTANT; memcpy(out->a, v.v, sizeof(v.v)); return true; }
    if (v.kind == R_TYPED) {
        if (v.text == "Checkerboard") {
            out->kind = CRAY_TEX_CHECKERBOARD;
            auto elem = [&](const char* key, double* dst, Err& ee) {
                return map_get(v.map, &v.used_keys, key, true, nullptr, [&](RawValue& x, Err& ie) {
                    if (scalar) return as_number(x, &dst[0], ie);
                    return as_triple(x, R_COLOR, "Color", dst, ie);
                }, ee);
            };
            if (!elem("a", out->a, e) || !elem("b", out->b, e)) return false;
            out->scale = 1.0;
            return map_get(v.map, &v.used_keys, "scale", false, nullptr, [&](RawValue& x, Err& ie) { return as_number(x, &out->scale, ie); }, e);
        }
        e = err_at("Unknown material type: " + v.text, v.map.loc);  // sic
        return false;
    }
    e = err_noloc(std::string("Cannot get Color, found ") + raw_kind_name(v));  // sic
    return false;
}

// ------------------------------------------------------------------ OBJ / MTL (obj.rs + tobj semantics)
struct MtlMaterial {
    std::string name;
    bool has_kd = false, has_ks = false, has_ns = false, has_d = false, has_ni = false, has_illum = false;
    double kd[3] = {0, 0, 0}, ks[3] = {0, 0, 0}, ns = 0, d = 1, ni = 1;
    int illum = 0;
    std::string map_kd, map_ks;
    std::map<std::string, std::string> unknown;
};
static std::string dir_of(const std::string& path) {
    size_t p = path.find_last_of('/');
    return p == std::string::npos ? std::string() : path.substr(0, p);
}
static std::string join_path(const std::string& dir, const std::string& f) {
    if (dir.empty() || (!f.empty() && f[0] == '/')) return f;
    return dir + "/" + f;
}
static std::vector<std::string> split_ws(const std::string& s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && isspace((unsigned char)s[i])) i++;
        size_t j = i;
        while (j < s.size() && !isspace((unsigned char)s[j])) j++;
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}
static std::string rest_after(const std::string& line, const std::string& key) {
    size_t p = line.find(key);
    std::string r = line.substr(p + key.size());
    size_t a = r.find_first_not_of(" \t\r");
    size_t b = r.find_last_not_of(" \t\r");
    return a == std::string::npos ? std::string() : r.substr(a, b - a + 1);
}
static bool read_file(const std::string& path, std::string* out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out->append(buf, n);
    fclose(f);
    return true;
}
static void parse_mtl(const std::string& text, std::vector<MtlMaterial>& mats) {
    size_t pos = 0;
    while (pos < text.size()) {
        size_t e = text.find('\n', pos);
        std::string line = text.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
        pos = e == std::string::npos ? text.size() : e + 1;
        std::vector<std::string> w = split_ws(line);
        if (w.empty() || w[0][0] == '#') continue;
        if (w[0] == "newmtl") { MtlMaterial m; m.name = rest_after(line, "newmtl"); mats.push_back(m); continue; }
        if (mats.empty()) continue;
        MtlMaterial& m = mats.back();
        auto f3 = [&](double* dst) { for (int i = 0; i < 3 && (size_t)i + 1 < w.size(); i++) dst[i] = strtod(w[i + 1].c_str(), nullptr); };
        if (w[0] == "Kd") { f3(m.kd); m.has_kd = true; }
        else if (w[0] == "Ks") { f3(m.ks); m.has_ks = true; }
        else if (w[0] == "Ka") { }
        else if (w[0] == "Ns" && w.size() > 1) { m.ns = strtod(w[1].c_str(), nullptr); m.has_ns = true; }
        else if (w[0] == "Ni" && w.size() > 1) { m.ni = strtod(w[1].c_str(), nullptr); m.has_ni = true; }
        else if (w[0] == "d" && w.size() > 1) { m.d = strtod(w[1].c_str(), nullptr); m.has_d = true; }
        else if (w[0] == "illum" && w.size() > 1) { m.illum = atoi(w[1].c_str()); m.has_illum = true; }
        else if (w[0] == "map_Kd") m.map_kd = rest_after(line, "map_Kd");
        else if (w[0] == "map_Ks") m.map_ks = rest_after(line, "map_Ks");
        else if (w[0] == "map_Ka" || w[0] == "map_Ns" || w[0] == "map_Bump" || w[0] == "map_bump" || w[0] == "bump" || w[0] == "map_d" || w[0] == "norm") { }
        else m.unknown[w[0]] = rest_after(line, w[0]);
    }
}

static bool load_image_texture(Builder& b, const std::string& obj_file, const std::string& tex_name, Tex* out, Err& e) {
    std::string path = join_path(dir_of(obj_file), tex_name);  // load_texture, obj.rs:16-24
    auto it = b.image_ids.find(path);
    if (it == b.image_ids.end()) {
        if (!b.loader) { e = err_noloc("Could not find texture file \"" + tex_name + "\" (no image loader supplied)"); return false; }
        uint32_t w = 0, h = 0;
        uint8_t* px = nullptr;
        if (b.loader(path.c_str(), b.loader_user, &w, &h, &px) != 0 || !px) { e = err_noloc("Could not find texture file \"" + tex_name + "\""); return false; }
        cray_image im; im.width = w; im.height = h; im.offset = b.pool.size();
        b.pool.insert(b.pool.end(), px, px + (size_t)w * h * 3);
        free(px);
        b.images.push_back(im);
        it = b.image_ids.emplace(path, (int32_t)b.images.size() - 1).first;
    }
    out->kind = CRAY_TEX_IMAGE; out->image = it->second;
    return true;
}

static void sub3(const double* a, const double* b, double* r) { for (int i = 0; i < 3; i++) r[i] = a[i] - b[i]; }
static void cross3(const double* a, const double* b, double* r) {
    r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = a[2] * b[0] - a[0] * b[2]; r[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// Shape::new_triangle_with_normals_and_texture_coordinates (shape.rs:96-132); false = degenerate
static bool make_triangle(const double* v0, const double* v1, const double* v2, const double* n0, const double* n1, const double* n2,
                          const double* uv0, const double* uv1, const double* uv2, cray_triangle* t) {
    double e1[3], e2[3], c[3];
    sub3(v1, v0, e1); sub3(v2, v0, e2); cross3(e2, e1, c);
    if (dot3(c, c) == 0.0 || dot3(n0, n0) == 0.0 || dot3(n1, n1) == 0.0 || dot3(n2, n2) == 0.0) return false;
    t->v0 = {v0[0], v0[1], v0[2]}; t->e1 = {e1[0], e1[1], e1[2]}; t->e2 = {e2[0], e2[1], e2[2]};
    t->n0 = {n0[0], n0[1], n0[2]};
    t->n01 = {n1[0] - n0[0], n1[1] - n0[1], n1[2] - n0[2]};
    t->n02 = {n2[0] - n0[0], n2[1] - n0[1], n2[2] - n0[2]};
    t->uv0[0] = uv0[0]; t->uv0[1] = uv0[1];
    t->uv01[0] = uv1[0] - uv0[0]; t->uv01[1] = uv1[1] - uv0[1];
    t->uv02[0] = uv2[0] - uv0[0]; t->uv02[1] = uv2[1] - uv0[1];
    return true;
}

// obj::load_obj (obj.rs:26-220)
static bool load_obj(Builder& b, const std::string& file_name, int32_t fallback_material, Err& e) {
    std::string path = join_path(b.base_dir, file_name);
    std::string text;
    if (!read_file(path, &text)) { e = err_noloc("Could not open mesh file \"" + file_name + "\""); return false; }
    std::vector<double> P, N, T;  // raw v / vn / vt
    struct Corner { int v, t, n; };
    struct Mesh { std::vector<Corner> corners; int material = -1; bool any_n = false, any_t = false; };
    std::vector<Mesh> meshes;
    std::vector<MtlMaterial> mtl;
    std::unordered_map<std::string, int> mtl_ids;
    Mesh cur;
    auto flush = [&]() { if (!cur.corners.empty()) meshes.push_back(cur); int m = cur.material; cur = Mesh(); cur.material = m; };
    size_t pos = 0;
    while (pos < text.size()) {
        size_t le = text.find('\n', pos);
        std::string line = text.substr(pos, le == std::string::npos ? std::string::npos : le - pos);
        pos = le == std::string::npos ? text.size() : le + 1;
        std::vector<std::string> w = split_ws(line);
        if (w.empty() || w[0][0] == '#') continue;
        if (w[0] == "v" && w.size() >= 4) { for (int i = 1; i <= 3; i++) P.push_back(strtod(w[i].c_str(), nullptr)); }
        else if (w[0] == "vn" && w.size() >= 4) { for (int i = 1; i <= 3; i++) N.push_back(strtod(w[i].c_str(), nullptr)); }
        else if (w[0] == "vt" && w.size() >= 2) { T.push_back(strtod(w[1].c_str(), nullptr)); T.push_back(w.size() >= 3 ? strtod(w[2].c_str(), nullptr) : 0.0); }
        else if (w[0] == "f" && w.size() >= 4) {
            std::vector<Corner> face;
            for (size_t i = 1; i < w.size(); i++) {
                Corner c{0, 0, 0};
                int idx[3] = {0, 0, 0};
                int field = 0;
                std::string num;
                for (size_t k = 0; k <= w[i].size(); k++) {
                    if (k == w[i].size() || w[i][k] == '/') { if (!num.empty() && field < 3) idx[field] = atoi(num.c_str()); num.clear(); field++; }
                    else num.push_back(w[i][k]);
                }
                auto fix = [](int i, size_t count) { return i > 0 ? i - 1 : (i < 0 ? (int)count + i : -1); };
                c.v = fix(idx[0], P.size() / 3); c.t = fix(idx[1], T.size() / 2); c.n = fix(idx[2], N.size() / 3);
                if (c.v < 0 || (size_t)c.v >= P.size() / 3) { e = err_noloc("bad vertex index in " + file_name); return false; }
                if (c.t >= 0) cur.any_t = true;
                if (c.n >= 0) cur.any_n = true;
                face.push_back(c);
            }
            for (size_t i = 2; i < face.size(); i++) { cur.corners.push_back(face[0]); cur.corners.push_back(face[i - 1]); cur.corners.push_back(face[i]); }  // fan
        }
        else if (w[0] == "o" || w[0] == "g") flush();
        else if (w[0] == "usemtl") {
            std::string name = rest_after(line, "usemtl");
            auto it = mtl_ids.find(name);
            int id = it == mtl_ids.end() ? -1 : it->second;
            if (id != cur.material) { flush(); cur.material = id; }
        }
        else if (w[0] == "mtllib") {
            std::string mtext;
            if (read_file(join_path(dir_of(path), rest_after(line, "mtllib")), &mtext)) {
                size_t before = mtl.size();
                parse_mtl(mtext, mtl);
                for (size_t i = before; i < mtl.size(); i++) mtl_ids[mtl[i].name] = (int)i;
            }  // else: "Error loading materials ... skipping" (obj.rs:32-38)
        }
    }
    flush();

    // MTL -> Material (obj.rs:61-105)
    std::vector<int32_t> mat_index(mtl.size(), -1);
    std::vector<bool> emissive(mtl.size(), false);
    std::vector<cray_color> emittance(mtl.size());
    for (size_t i = 0; i < mtl.size(); i++) {
        const MtlMaterial& m = mtl[i];
        Tex diffuse, specular, rough;
        if (!m.map_kd.empty()) { if (!load_image_texture(b, path, m.map_kd, &diffuse, e)) return false; }
        else {
            if (!m.has_kd) { e = err_noloc("material '" + m.name + "' has no Kd (reference: unwrap on None, obj.rs:67)"); return false; }
            memcpy(diffuse.a, m.kd, sizeof(m.kd));
        }
        if (!m.map_ks.empty()) { if (!load_image_texture(b, path, m.map_ks, &specular, e)) return false; }
        else if (m.has_ks) memcpy(specular.a, m.ks, sizeof(m.ks));
        double ke[3] = {0, 0, 0};
        auto k = m.unknown.find("Ke");
        if (k != m.unknown.end()) { std::vector<std::string> w = split_ws(k->second); for (int j = 0; j < 3 && (size_t)j < w.size(); j++) ke[j] = strtod(w[j].c_str(), nullptr); }
        double shininess = m.has_ns ? m.ns : 0.0;
        rough.scalar = true;
        rough.a[0] = 180.0 * (1.0 - pow(2.718281828459045, -shininess / 100.0));  // obj.rs:84 (E.powf)
        double dissolve = m.has_d ? m.d : 1.0;
        if (!(ke[0] == 0.0 && ke[1] == 0.0 && ke[2] == 0.0)) {
            emissive[i] = true; emittance[i] = {ke[0], ke[1], ke[2]};
            mat_index[i] = fallback_material;
        } else if (dissolve < 1.0) {
            mat_index[i] = new_glass(b, diffuse, diffuse, m.has_ni ? m.ni : 1.0);
        } else if (m.has_illum && m.illum >= 3 && m.illum <= 9) {
            mat_index[i] = new_metal(b, diffuse, specular);
        } else {
            mat_index[i] = new_plastic(b, diffuse, specular, rough);
        }
    }

    for (const Mesh& mesh : meshes) {
        int32_t material = mesh.material >= 0 ? mat_index[mesh.material] : fallback_material;
        bool emit = mesh.material >= 0 && emissive[mesh.material];
        for (size_t c = 0; c + 2 < mesh.corners.size(); c += 3) {
            double v[3][3], n[3][3], uv[3][2] = {{0.0, 0.0}, {1.0, 0.0}, {1.0, 1.0}};
            for (int k = 0; k < 3; k++) {
                const Corner& cn = mesh.corners[c + k];
                v[k][0] = P[3 * cn.v]; v[k][1] = P[3 * cn.v + 1]; v[k][2] = -P[3 * cn.v + 2];  // RH -> LH (obj.rs:131)
            }
            double a[3], bb[3], fn[3];
            sub3(v[2], v[0], a); sub3(v[1], v[0], bb); cross3(a, bb, fn);  // (vk - vi) x (vj - vi), obj.rs:159
            double mag = sqrt(dot3(fn, fn));
            for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) n[k][j] = fn[j] / mag;
            if (mesh.any_n) for (int k = 0; k < 3; k++) {
                const Corner& cn = mesh.corners[c + k];
                if (cn.n >= 0 && (size_t)cn.n < N.size() / 3) { n[k][0] = N[3 * cn.n]; n[k][1] = N[3 * cn.n + 1]; n[k][2] = -N[3 * cn.n + 2]; }
            }
            if (mesh.any_t) for (int k = 0; k < 3; k++) {
                const Corner& cn = mesh.corners[c + k];
                if (cn.t >= 0 && (size_t)cn.t < T.size() / 2) { uv[k][0] = T[2 * cn.t]; uv[k][1] = 1.0 - T[2 * cn.t + 1]; }  // obj.rs:149
            }
            cray_triangle t;
            if (!make_triangle(v[0], v[1], v[2], n[0], n[1], n[2], uv[0], uv[1], uv[2], &t)) continue;  // degenerate: skipped
            cray_prim p;
            p.shape_kind = CRAY_SHAPE_TRIANGLE; p.shape = (uint32_t)b.triangles.size();
            b.triangles.push_back(t);
            if (emit) {
                cray_light l; memset(&l, 0, sizeof(l));
                l.kind = CRAY_LIGHT_AREA; l.prim = (int32_t)b.prims.size(); l.c = emittance[mesh.material];
                p.material = -1; p.light = (int32_t)b.lights.size();
                b.lights.push_back(l);
            } else { p.material = material; p.light = -1; }
            b.prims.push_back(p);
        }
    }
    return true;
}

// count TypedRawValueMap drops that would warn about unused keys (scene_parser.rs:571-586)
static void count_unused(const RawValue& v, uint32_t& n) {
    if (v.kind == R_TYPED) {
        bool unused = false;
        for (auto& kv : v.map.map) if (!v.used_keys.count(kv.first)) unused = true;
        if (unused) n++;
    }
    if (v.kind == R_TYPED || v.kind == R_MAP) for (auto& kv : v.map.map) count_unused(*kv.second, n);
    if (v.kind == R_ARRAY) for (auto& x : v.array) count_unused(*x, n);
}

static bool build_scene(RawMap& top, Builder& b, Err& e) {  // parse_scene :1078-1117
    double tmp;
    bool present;
    // max_depth / num_samples (`as usize`)
    if (!map_get(top, nullptr, "max_depth", false, &present, [&](RawValue& v, Err& ie) { return as_number(v, &tmp, ie); }, e)) return false;
    b.max_depth = present ? (uint32_t)as_usize_sat(tmp) : 8;      // DEFAULT_MAX_DEPTH :796
    if (!map_get(top, nullptr, "num_samples", false, &present, [&](RawValue& v, Err& ie) { return as_number(v, &tmp, ie); }, e)) return false;
    b.num_samples = present ? (uint32_t)as_usize_sat(tmp) : 4;    // DEFAULT_NUM_SAMPLES :797

    // camera (:801-845)
    auto camera_conv = [&](RawValue& v, Err& ie) -> bool {
        if (v.kind != R_TYPED) { ie = err_noloc(std::string("Cannot get Camera, found ") + raw_kind_name(v)); return false; }
        cray_camera_desc& c = b.camera;
        memset(&c, 0, sizeof(c));
        double w = 0, h = 0, o[3], t[3], u[3];
        if (!map_get(v.map, &v.used_keys, "film", true, nullptr, [&](RawValue& f, Err& fe) -> bool {
                if (f.kind != R_MAP) { fe = err_noloc(std::string("Cannot get Film, found ") + raw_kind_name(f)); return false; }
                return map_get(f.map, nullptr, "width", true, nullptr, [&](RawValue& x, Err& xe) { return as_number(x, &w, xe); }, fe) &&
                       map_get(f.map, nullptr, "height", true, nullptr, [&](RawValue& x, Err& xe) { return as_number(x, &h, xe); }, fe);
            }, ie)) return false;
        if (!map_get(v.map, &v.used_keys, "origin", true, nullptr, [&](RawValue& x, Err& xe) { return as_triple(x, R_POINT, "Point", o, xe); }, ie)) return false;
        if (!map_get(v.map, &v.used_keys, "target", true, nullptr, [&](RawValue& x, Err& xe) { return as_triple(x, R_POINT, "Point", t, xe); }, ie)) return false;
        if (!map_get(v.map, &v.used_keys, "up", true, nullptr, [&](RawValue& x, Err& xe) { return as_triple(x, R_VECTOR, "Vector", u, xe); }, ie)) return false;
        c.lens_radius = 0.0; c.focal_distance = 1e6;  // DEFAULT_FOCAL_DISTANCE :798
        if (!map_get(v.map, &v.used_keys, "lens_radius", false, nullptr, [&](RawValue& x, Err& xe) { return as_number(x, &c.lens_radius, xe); }, ie)) return false;
        if (!map_get(v.map, &v.used_keys, "focal_distance", false, nullptr, [&](RawValue& x, Err& xe) { return as_number(x, &c.focal_distance, xe); }, ie)) return false;
        c.film_width = (uint32_t)as_usize_sat(w); c.film_height = (uint32_t)as_usize_sat(h);
        c.origin = {o[0], o[1], o[2]}; c.target = {t[0], t[1], t[2]}; c.up = {u[0], u[1], u[2]};
        if (v.text == "Perspective") {
            c.type = CRAY_CAMERA_PERSPECTIVE;
            return map_get(v.map, &v.used_keys, "fov", true, nullptr, [&](RawValue& x, Err& xe) { return as_number(x, &c.fov, xe); }, ie);
        }
        if (v.text == "Orthographic") { c.type = CRAY_CAMERA_ORTHOGRAPHIC; return true; }
        ie = err_noloc("Unknown camera type: " + v.text);
        return false;
    };
    if (!map_get(top, nullptr, "camera", true, nullptr, camera_conv, e)) return false;

    // lights (:866-901)
    auto lights_conv = [&](RawValue& arr, Err& ie) -> bool {
        if (arr.kind != R_ARRAY) { ie = err_noloc(std::string("Cannot get Array, found ") + raw_kind_name(arr)); return false; }
        for (auto& pv : arr.array) {
            RawValue& v = *pv;
            if (v.kind != R_TYPED) { ie = err_noloc(std::string("Cannot get Light, found ") + raw_kind_name(v)); return false; }
            cray_light l; memset(&l, 0, sizeof(l)); l.prim = -1;
            double p3[3] = {0, 0, 0}, c3[3];
            auto color = [&](Err& xe2) { return map_get(v.map, &v.used_keys, "intensity", true, nullptr, [&](RawValue& x, Err& xe) { return as_triple(x, R_COLOR, "Color", c3, xe); }, xe2); };
            if (v.text == "Point") {
                l.kind = CRAY_LIGHT_POINT;
                if (!map_get(v.map, &v.used_keys, "origin", true, nullptr, [&](RawValue& x, Err& xe) { return as_triple(x, R_POINT, "Point", p3, xe); }, ie) || !color(ie)) return false;
            } else if (v.text == "Distant") {
                l.kind = CRAY_LIGHT_DISTANT;
                if (!map_get(v.map, &v.used_keys, "direction", true, nullptr, [&](RawValue& x, Err& xe) { return as_triple(x, R_VECTOR, "Vector", p3, xe); }, ie) || !color(ie)) return false;
                double mag = sqrt(p3[0] * p3[0] + p3[1] * p3[1] + p3[2] * p3[2]);  // direction.normalized(), :888
                p3[0] /= mag; p3[1] /= mag; p3[2] /= mag;
            } else if (v.text == "Infinite") {
                l.kind = CRAY_LIGHT_INFINITE;
                if (!color(ie)) return false;
            } else { ie = err_noloc("Unknown light type: " + v.text); return false; }
            l.v = {p3[0], p3[1], p3[2]}; l.c = {c3[0], c3[1], c3[2]};
            b.lights.push_back(l);
        }
        return true;
    };
    if (!map_get(top, nullptr, "lights", true, nullptr, lights_conv, e)) return false;

    // materials (:942-981)
    std::map<std::string, int32_t> material_ids, shape_ids;
    auto materials_conv = [&](RawValue& m, Err& ie) -> bool {
        if (m.kind != R_MAP) { ie = err_noloc(std::string("Cannot get Map, found ") + raw_kind_name(m)); return false; }
        for (auto& kv : m.map.map) {
            RawValue& v = *kv.second;
            if (v.kind != R_TYPED) { ie = err_noloc(std::string("Cannot get Material, found ") + raw_kind_name(v)); return false; }
            auto tex = [&](const char* key, bool scalar, Tex* t, Err& te) { return map_get(v.map, &v.used_keys, key, true, nullptr, [&](RawValue& x, Err& xe) { return as_texture(x, scalar, t, xe); }, te); };
            Tex a, c, d;
            int32_t id;
            if (v.text == "Matte") { if (!tex("reflectance", false, &a, ie) || !tex("sigma", true, &c, ie)) return false; id = new_matte(b, a, c); }
            else if (v.text == "Glass") {
                double eta = 0;
                if (!tex("reflectance", false, &a, ie) || !tex("transmittance", false, &c, ie) ||
                    !map_get(v.map, &v.used_keys, "eta", true, nullptr, [&](RawValue& x, Err& xe) { return as_number(x, &eta, xe); }, ie)) return false;
                id = new_glass(b, a, c, eta);
            }
            else if (v.text == "Plastic") { if (!tex("diffuse", false, &a, ie) || !tex("specular", false, &c, ie) || !tex("roughness", true, &d, ie)) return false; id = new_plastic(b, a, c, d); }
            else if (v.text == "Metal") { if (!tex("eta", false, &a, ie) || !tex("k", false, &c, ie)) return false; id = new_metal(b, a, c); }
            else { ie = err_at("Unknown material type: " + v.text, v.map.loc); return false; }
            material_ids[kv.first] = id;
        }
        return true;
    };
    if (!map_get(top, nullptr, "materials", true, nullptr, materials_conv, e)) return false;

    // shapes (:984-1019): kept as descriptions; a shape may be referenced by several primitives
    struct ShapeDef { int kind; cray_sphere_desc s; cray_disk_desc d; cray_triangle t; };
    std::vector<ShapeDef> shape_defs;
    auto shapes_conv = [&](RawValue& m, Err& ie) -> bool {
        if (m.kind != R_MAP) { ie = err_noloc(std::string("Cannot get Map, found ") + raw_kind_name(m)); return false; }
        for (auto& kv : m.map.map) {
            RawValue& v = *kv.second;
            if (v.kind != R_TYPED) { ie = err_noloc(std::string("Cannot get Shape, found ") + raw_kind_name(v)); return false; }
            ShapeDef sd; memset(&sd, 0, sizeof(sd));
            auto pt = [&](const char* key, double* p, Err& pe) { return map_get(v.map, &v.used_keys, key, true, nullptr, [&](RawValue& x, Err& xe) { return as_triple(x, R_POINT, "Point", p, xe); }, pe); };
            auto num = [&](const char* key, bool req, double* p, Err& pe) { return map_get(v.map, &v.used_keys, key, req, nullptr, [&](RawValue& x, Err& xe) { return as_number(x, p, xe); }, pe); };
            double o[3];
            if (v.text == "Sphere") {
                sd.kind = CRAY_SHAPE_SPHERE;
                if (!pt("origin", o, ie) || !num("radius", true, &sd.s.radius, ie)) return false;
                sd.s.origin = {o[0], o[1], o[2]};
            } else if (v.text == "Triangle") {
                sd.kind = CRAY_SHAPE_TRIANGLE;
                double v0[3], v1[3], v2[3];
                if (!pt("v0", v0, ie) || !pt("v1", v1, ie) || !pt("v2", v2, ie)) return false;
                // Shape::new_triangle (shape.rs:70-95)
                double e1[3], e2[3], n0[3];
                sub3(v1, v0, e1); sub3(v2, v0, e2); cross3(e2, e1, n0);
                double mag = sqrt(dot3(n0, n0));
                if (mag == 0.0) { ie = err_at("Degenerate triangle: " + v.text, v.map.loc); return false; }
                sd.t.v0 = {v0[0], v0[1], v0[2]}; sd.t.e1 = {e1[0], e1[1], e1[2]}; sd.t.e2 = {e2[0], e2[1], e2[2]};
                sd.t.n0 = {n0[0] / mag, n0[1] / mag, n0[2] / mag};
                sd.t.uv0[0] = 0; sd.t.uv0[1] = 0; sd.t.uv01[0] = 1; sd.t.uv01[1] = 0; sd.t.uv02[0] = 1; sd.t.uv02[1] = 1;
            } else if (v.text == "Disk") {
                sd.kind = CRAY_SHAPE_DISK;
                if (!pt("origin", o, ie) || !num("rotate_x", false, &sd.d.rotate_x, ie) || !num("rotate_y", false, &sd.d.rotate_y, ie) ||
                    !num("radius", true, &sd.d.radius, ie) || !num("inner_radius", false, &sd.d.inner_radius, ie)) return false;
                sd.d.origin = {o[0], o[1], o[2]};
            } else { ie = err_noloc("Unknown shape type: " + v.text); return false; }
            shape_ids[kv.first] = (int32_t)shape_defs.size();
            shape_defs.push_back(sd);
        }
        return true;
    };
    if (!map_get(top, nullptr, "shapes", true, nullptr, shapes_conv, e)) return false;

    // primitives (:1025-1076, :1092-1102)
    std::map<int32_t, uint32_t> emitted;  // shape def -> index in its table (shared Arc<Shape>)
    auto prims_conv = [&](RawValue& arr, Err& ie) -> bool {
        if (arr.kind != R_ARRAY) { ie = err_noloc(std::string("Cannot get Array, found ") + raw_kind_name(arr)); return false; }
        for (auto& pv : arr.array)
            if (pv->kind != R_TYPED) { ie = err_noloc(std::string("Cannot get TypedRawValueMap, found ") + raw_kind_name(*pv)); return false; }
        return true;
    };
    if (!map_get(top, nullptr, "primitives", true, nullptr, prims_conv, e)) return false;
    for (auto& pv : top.map["primitives"]->array) {
        RawValue& v = *pv;
        auto str = [&](const char* key, std::string* s, Err& se) { return map_get(v.map, &v.used_keys, key, true, nullptr, [&](RawValue& x, Err& xe) { return as_string(x, s, xe); }, se); };
        if (v.text == "Shape") {
            std::string shape_name, material_name;
            if (!str("shape", &shape_name, e)) return false;
            auto si = shape_ids.find(shape_name);
            if (si == shape_ids.end()) { e = err_at("Cannot find shape named '" + shape_name + "'", v.map.loc); return false; }
            const ShapeDef& sd = shape_defs[si->second];
            cray_prim p; p.shape_kind = sd.kind;
            auto em = emitted.find(si->second);
            if (em != emitted.end()) p.shape = em->second;
            else {
                if (sd.kind == CRAY_SHAPE_SPHERE) { p.shape = (uint32_t)b.spheres.size(); b.spheres.push_back(sd.s); }
                else if (sd.kind == CRAY_SHAPE_DISK) { p.shape = (uint32_t)b.disks.size(); b.disks.push_back(sd.d); }
                else { p.shape = (uint32_t)b.triangles.size(); b.triangles.push_back(sd.t); }
                emitted[si->second] = p.shape;
            }
            v.used_keys.insert("emittance");
            if (v.map.map.count("emittance")) {
                double c3[3];
                if (!map_get(v.map, &v.used_keys, "emittance", true, nullptr, [&](RawValue& x, Err& xe) { return as_triple(x, R_COLOR, "Color", c3, xe); }, e)) return false;
                cray_light l; memset(&l, 0, sizeof(l));
                l.kind = CRAY_LIGHT_AREA; l.prim = (int32_t)b.prims.size(); l.c = {c3[0], c3[1], c3[2]};
                p.material = -1; p.light = (int32_t)b.lights.size();
                b.lights.push_back(l);
            } else {
                if (!str("material", &material_name, e)) return false;
                auto mi = material_ids.find(material_name);
                if (mi == material_ids.end()) { e = err_at("Cannot find material named '" + material_name + "'", v.map.loc); return false; }
                p.material = mi->second; p.light = -1;
            }
            b.prims.push_back(p);
        } else if (v.text == "Mesh") {
            std::string file_name, material_name;
            if (!str("file_name", &file_name, e) || !str("fallback_material", &material_name, e)) return false;
            auto mi = material_ids.find(material_name);
            if (mi == material_ids.end()) { e = err_at("Cannot find material named '" + material_name + "'", v.map.loc); return false; }
            if (!load_obj(b, file_name, mi->second, e)) return false;
        } else { e = err_at("Unknown primitive type: " + v.text, v.map.loc); return false; }
    }
    if (b.lights.empty()) { Loc z; e = err_at("No lights in the scene.", z); return false; }
    return true;
}

static void set_err(cray_parser_error* out, const Err& e) {
    if (!out) return;
    out->has_location = e.has_loc ? 1 : 0;
    out->line = e.loc.line; out->column = e.loc.column;
    snprintf(out->message, sizeof(out->message), "%s", e.message.c_str());
}

}  // namespace

struct cray_owned_scene {
    Builder b;
    cray_scene_desc desc;
};

extern "C" int cray_cry_tokenize(const char* input, cray_token** tokens, size_t* n_tokens, cray_parser_error* err) {
    std::vector<Token> toks;
    Err e;
    if (!input || !tokens || !n_tokens) return -1;
    if (!tokenize(input, toks, e)) { set_err(err, e); return -1; }
    cray_token* out = (cray_token*)calloc(toks.size(), sizeof(cray_token));
    for (size_t i = 0; i < toks.size(); i++) {
        out[i].kind = (int32_t)toks[i].kind; out[i].line = toks[i].loc.line; out[i].column = toks[i].loc.column; out[i].number = toks[i].number;
        out[i].text = (toks[i].kind == T_IDENT || toks[i].kind == T_STRING) ? strdup(toks[i].text.c_str()) : nullptr;
    }
    *tokens = out; *n_tokens = toks.size();
    return 0;
}
extern "C" void cray_cry_free_tokens(cray_token* tokens, size_t n) {
    if (!tokens) return;
    for (size_t i = 0; i < n; i++) free((void*)tokens[i].text);
    free(tokens);
}
extern "C" int cray_cry_parse_value(const char* input, char** dump, cray_parser_error* err) {
    std::vector<Token> toks;
    Err e;
    if (!input || !dump) return -1;
    if (!tokenize(input, toks, e)) { set_err(err, e); return -1; }
    TokenStream ts{toks};
    RawValue v;
    if (!parse_value(ts, v, e)) { set_err(err, e); return -1; }
    std::string s;
    dump_value(v, s);
    *dump = strdup(s.c_str());
    return 0;
}
extern "C" void cray_cry_free_string(char* s) { free(s); }

extern "C" int cray_cry_parse_scene(const char* input, const char* base_dir, cray_image_loader loader, void* loader_user,
                                    const cray_scene_overrides* ov, cray_owned_scene** out, cray_parser_error* err) {
    if (!input || !out) return -1;
    *out = nullptr;
    std::vector<Token> toks;
    Err e;
    if (!tokenize(input, toks, e)) { set_err(err, e); return -1; }
    TokenStream ts{toks};
    RawValue top;
    top.kind = R_MAP;
    if (!parse_map(ts, top.map, e)) { set_err(err, e); return -1; }
    std::unique_ptr<cray_owned_scene> os(new cray_owned_scene());
    Builder& b = os->b;
    b.base_dir = base_dir ? base_dir : "";
    b.loader = loader ? loader : cray_default_image_loader;   // no loader: PNM / JPEG through cray_load_image (cray_io.h)
    b.loader_user = loader_user;
    if (!build_scene(top.map, b, e)) { set_err(err, e); return -1; }
    count_unused(top, b.warnings);
    if (ov) {
        if (ov->width) b.camera.film_width = ov->width;
        if (ov->height) b.camera.film_height = ov->height;
        if (ov->num_samples) b.num_samples = ov->num_samples;
        if (ov->max_depth) b.max_depth = ov->max_depth;
    }
    cray_scene_desc& d = os->desc;
    memset(&d, 0, sizeof(d));
    d.max_depth = b.max_depth; d.num_samples = b.num_samples; d.camera = b.camera;
    d.n_spheres = (uint32_t)b.spheres.size(); d.spheres = b.spheres.data();
    d.n_disks = (uint32_t)b.disks.size(); d.disks = b.disks.data();
    d.n_triangles = (uint32_t)b.triangles.size(); d.triangles = b.triangles.data();
    d.n_prims = (uint32_t)b.prims.size(); d.prims = b.prims.data();
    d.n_lights = (uint32_t)b.lights.size(); d.lights = b.lights.data();
    d.n_materials = (uint32_t)b.materials.size(); d.materials = b.materials.data();
    d.n_bxdfs = (uint32_t)b.bxdfs.size(); d.bxdfs = b.bxdfs.data();
    d.n_textures = (uint32_t)b.textures.size(); d.textures = b.textures.data();
    d.n_images = (uint32_t)b.images.size(); d.images = b.images.data();
    d.image_pool_bytes = b.pool.size(); d.image_pool = b.pool.data();
    *out = os.release();
    return 0;
}
extern "C" const cray_scene_desc* cray_owned_scene_desc(const cray_owned_scene* s) { return s ? &s->desc : nullptr; }
extern "C" uint32_t cray_owned_scene_warnings(const cray_owned_scene* s) { return s ? s->b.warnings : 0; }
extern "C" void cray_owned_scene_free(cray_owned_scene* s) { delete s; }
