"""The three descriptions of the C ABI agree: the headers (as gcc lays them out), bindings/cray_sys.rs (as repr(C) lays it
out) and the ctypes / numpy mirrors the Python host uses.  A binding that lags the header (round 2: a 56-byte
CrayRenderParams against the 80-byte cray_render_params) overruns memory in the caller; this test is the guard.

Seam: reference src/bin/craytracer.rs:224-259 (`render` and its output contract) — the structs below are its arguments.
"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RS = os.path.join(ROOT, 'bindings', 'cray_sys.rs')

PRIM = {'u8': (1, 1), 'u16': (2, 2), 'u32': (4, 4), 'i32': (4, 4), 'u64': (8, 8), 'i64': (8, 8), 'f32': (4, 4), 'f64': (8, 8),
        'usize': (8, 8)}


def parse_rust(text):
    """-> {RustName: (c_name, [(field, c_field, type_string)])} for every `// C: name` + #[repr(C)] pub struct."""
    out = {}
    pat = re.compile(r'// C: (\w+)\n#\[repr\(C\)\]\n(?:#\[derive\([^\n]*\)\]\n)?pub struct (\w+) \{\n(.*?)\n\}', re.S)
    for c_name, rs_name, body in pat.findall(text):
        fields = []
        for line in body.split('\n'):
            line = line.strip()
            if not line or line.startswith('//'):
                continue
            m = re.match(r'pub (\w+): (.+?),(?:\s*// C: (\w+))?(?:\s*//.*)?$', line)
            assert m, 'unparsed field line in %s: %r' % (rs_name, line)
            fields.append((m.group(1), m.group(3) or m.group(1), m.group(2)))
        out[rs_name] = (c_name, fields)
    return out


def layout(ty, structs, cache):
    """(size, align) of a Rust type under repr(C) on x86-64 / the LP64 targets the library supports."""
    ty = ty.strip()
    if ty in PRIM:
        return PRIM[ty]
    if ty.startswith('*const ') or ty.startswith('*mut ') or ty.startswith('Option<'):
        return (8, 8)
    m = re.match(r'\[(.+); (\d+)\]$', ty)
    if m:
        s, a = layout(m.group(1), structs, cache)
        return (s * int(m.group(2)), a)
    if ty in structs:
        return struct_layout(ty, structs, cache)[0:2]
    raise AssertionError('unknown Rust type %r' % ty)


def struct_layout(name, structs, cache):
    if name in cache:
        return cache[name]
    off, align, offsets = 0, 1, {}
    for _, c_field, ty in structs[name][1]:
        s, a = layout(ty, structs, cache)
        off = (off + a - 1) // a * a
        offsets[c_field] = (off, s)
        off += s
        align = max(align, a)
    size = (off + align - 1) // align * align
    cache[name] = (size, align, offsets)
    return cache[name]


@pytest.fixture(scope='module')
def rust():
    with open(RS) as fh:
        structs = parse_rust(fh.read())
    assert len(structs) >= 25, sorted(structs)
    return structs


@pytest.fixture(scope='module')
def c_layout(rust, tmp_path_factory):
    """sizeof / offsetof / field sizes of every C struct the Rust file names, as gcc sees the headers."""
    d = tmp_path_factory.mktemp('abi')
    src = ['#include <stddef.h>', '#include <stdio.h>', '#include "cray.h"', '#include "cray_host.h"', '#include "cray_cry.h"',
           '#include "cray_io.h"', 'int main(void) {']
    for rs_name, (c_name, fields) in rust.items():
        src.append('  printf("S %s %%zu\\n", sizeof(%s));' % (c_name, c_name))
        for _, c_field, _ in fields:
            src.append('  printf("F %s %s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s*)0)->%s));' % (c_name, c_field, c_name, c_field, c_name, c_field))
    src += ['  return 0;', '}']
    cfile = d / 'abi.c'
    cfile.write_text('\n'.join(src))
    exe = d / 'abi'
    subprocess.check_call(['gcc', '-std=c11', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), '-o', str(exe), str(cfile)])
    sizes, fields = {}, {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        p = line.split()
        if p[0] == 'S':
            sizes[p[1]] = int(p[2])
        else:
            fields.setdefault(p[1], {})[p[2]] = (int(p[3]), int(p[4]))
    return sizes, fields


def header_fields(c_name):
    """Field names of a typedef'd struct as written in the headers (so a field the Rust file lacks is noticed even when
    padding hides it from sizeof)."""
    text = ''
    for h in ('cray_scene_desc.h', 'cray.h', 'cray_cry.h', 'cray_io.h', 'cray_host.h'):
        with open(os.path.join(ROOT, 'include', h)) as fh:
            text += fh.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    m = re.search(r'typedef struct(?: \w+)? \{([^{}]*)\} %s;' % c_name, text)
    assert m, c_name
    names = []
    for decl in m.group(1).split(';'):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r'^(const\s+)?(struct\s+)?\w+\s*\**\s*', '', decl, count=1)   # drop the type
        for part in decl.split(','):
            nm = re.match(r'\s*\**\s*(\w+)', part)
            names.append(nm.group(1))
    return names


def test_rust_structs_match_the_headers(rust, c_layout):
    sizes, fields = c_layout
    cache = {}
    for rs_name, (c_name, rs_fields) in rust.items():
        size, _, offsets = struct_layout(rs_name, rust, cache)
        assert size == sizes[c_name], '%s: Rust %d bytes, C %d bytes' % (rs_name, size, sizes[c_name])
        for c_field, (off, fsize) in offsets.items():
            assert (off, fsize) == fields[c_name][c_field], '%s.%s: Rust (offset, size) %s, C %s' % (rs_name, c_field, (off, fsize), fields[c_name][c_field])
        assert [f[1] for f in rs_fields] == header_fields(c_name), '%s: field list differs from %s' % (rs_name, c_name)


def test_every_header_struct_is_bound(rust):
    bound = {c for c, _ in rust.values()}
    text = ''
    for h in ('cray_scene_desc.h', 'cray.h', 'cray_cry.h'):
        with open(os.path.join(ROOT, 'include', h)) as fh:
            text += re.sub(r'/\*.*?\*/', '', fh.read(), flags=re.S)
    declared = set(re.findall(r'typedef struct(?: \w+)? \{[^{}]*\} (\w+);', text))
    assert declared and declared <= bound, sorted(declared - bound)


def test_abi_version_constant(rust):
    with open(RS) as fh:
        rs = fh.read()
    with open(os.path.join(ROOT, 'include', 'cray.h')) as fh:
        h = fh.read()
    assert re.search(r'CRAY_ABI_VERSION: u32 = (\d+)', rs).group(1) == re.search(r'#define CRAY_ABI_VERSION (\d+)', h).group(1)


def test_rust_constants_match_the_headers():
    """Every `pub const CRAY_*` of the Rust file has the value the headers give the same name (enum members and #defines)."""
    with open(RS) as fh:
        rs = fh.read()
    text = ''
    for h in ('cray_scene_desc.h', 'cray.h', 'cray_host.h', 'cray_cry.h', 'cray_io.h'):
        with open(os.path.join(ROOT, 'include', h)) as fh:
            text += re.sub(r'/\*.*?\*/', '', fh.read(), flags=re.S)
    c_vals = {}
    for body in re.findall(r'enum\s*\{(.*?)\}', text, flags=re.S):
        for name, val in re.findall(r'(CRAY_\w+)\s*=\s*(-?\d+)', body):
            c_vals[name] = int(val)
    for name, val in re.findall(r'#define\s+(CRAY_\w+)\s+(-?\d+)\b', text):
        c_vals[name] = int(val)
    consts = re.findall(r'pub const (CRAY_\w+): \w+ = (-?\d+);', rs)
    assert len(consts) >= 40
    for name, val in consts:
        assert name in c_vals, '%s is not a constant of include/' % name
        assert c_vals[name] == int(val), '%s: %s in Rust, %d in C' % (name, val, c_vals[name])


def test_rust_functions_match_the_headers():
    """Every extern fn of the Rust block is declared in a header with the same number of parameters."""
    with open(RS) as fh:
        rs = fh.read()
    text = ''
    for h in ('cray.h', 'cray_host.h', 'cray_cry.h', 'cray_io.h'):
        with open(os.path.join(ROOT, 'include', h)) as fh:
            text += re.sub(r'/\*.*?\*/', '', fh.read(), flags=re.S)
    fns = re.findall(r'pub fn (\w+)\((.*?)\)(?: -> [^;]+)?;', rs, flags=re.S)
    assert len(fns) >= 40
    for name, args in fns:
        m = re.search(r'\b%s\s*\(([^;{]*?)\)\s*;' % name, text, flags=re.S)
        assert m, '%s is not declared in include/' % name
        c_args = m.group(1).strip()
        n_c = 0 if c_args in ('', 'void') else len(c_args.split(','))
        n_rs = 0 if not args.strip() else len([a for a in args.split(',') if a.strip()])
        assert n_c == n_rs, '%s: %d parameters in C, %d in Rust' % (name, n_c, n_rs)


def _ctypes_layout(cls):
    return C.sizeof(cls), {n: (getattr(cls, n).offset, getattr(cls, n).size) for n, _ in cls._fields_}


def test_python_mirrors_match_the_headers(c_layout):
    sizes, fields = c_layout
    from craytracer_amd import backend, scene
    for cls, c_name in ((backend.RenderParams, 'cray_render_params'), (backend.Stats, 'cray_stats'), (backend.FlatScene, 'cray_flat_scene'),
                        (backend.BvhBuildStats, 'cray_bvh_build_stats'), (backend.CommInfo, 'cray_comm_info')):
        size, offs = _ctypes_layout(cls)
        assert size == sizes[c_name], c_name
        assert offs == fields[c_name], c_name
    for dt, c_name in ((backend.RAY_DT, 'cray_ray'), (backend.HIT_DT, 'cray_hit'), (backend.BVH_NODE_DT, 'cray_bvh_node'),
                       (scene.TEXTURE_DT, 'cray_texture'), (scene.IMAGE_DT, 'cray_image'), (scene.BXDF_DT, 'cray_bxdf'),
                       (scene.MATERIAL_DT, 'cray_material'), (scene.SPHERE_DT, 'cray_sphere_desc'), (scene.DISK_DT, 'cray_disk_desc'),
                       (scene.TRIANGLE_DT, 'cray_triangle'), (scene.PRIM_DT, 'cray_prim'), (scene.LIGHT_DT, 'cray_light')):
        assert dt.itemsize == sizes[c_name], c_name
        got = {n: (dt.fields[n][1], dt.fields[n][0].itemsize) for n in dt.names}
        assert got == fields[c_name], c_name


def test_integration_md_quotes_the_binding_file():
    """INTEGRATION.md §1 shows the checked file, not a hand-copied variant of it."""
    with open(os.path.join(ROOT, 'INTEGRATION.md')) as fh:
        md = fh.read()
    with open(RS) as fh:
        rs = fh.read()
    assert 'bindings/cray_sys.rs' in md and 'ABI version 2' in md and 'ABI version 1' not in md
    for block in re.findall(r'```rust\n(.*?)```', md, flags=re.S):
        for m in re.finditer(r'pub struct (\w+) \{(.*?)\n\}', block, flags=re.S):
            assert m.group(0) in rs, 'INTEGRATION.md shows a %s that differs from bindings/cray_sys.rs' % m.group(1)
