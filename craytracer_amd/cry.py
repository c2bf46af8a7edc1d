"""`.cry` scene files and OBJ/MTL meshes: ctypes binding of include/cray_cry.h.

    tokenize(text)                 tokenizer::tokenize            (src/scene_parser.rs:174-253)
    parse_value(text)              RawValue::from_tokens          (:333-402) as a canonical dump string
    parse_scene(text, base_dir)    scene_parser::parse_scene      (:1078-1117) -> ParsedScene

A ParsedScene quacks like craytracer_amd.scene.Scene (desc(), film_bounds(), num_samples, lights),
so it can be handed to backend.HostScene (and to the test oracle).
"""
import ctypes as C
import os

import numpy as np

from . import backend
from .scene import CSceneDesc

TOKEN_NAMES = ['Identifier', 'Number', 'String', 'LeftBrace', 'RightBrace', 'LeftBracket', 'RightBracket',
               'LeftParen', 'RightParen', 'Comma', 'Colon', 'Eof']


class ParserError(Exception):
    """ParserError {message, location: Option<Location>} (src/scene_parser.rs:65-84)."""

    def __init__(self, message, location):
        super().__init__('%s at %s' % (message, location) if location else message)
        self.message, self.location = message, location


class _CErr(C.Structure):
    _fields_ = [('has_location', C.c_int32), ('line', C.c_uint32), ('column', C.c_uint32), ('message', C.c_char * 512)]

    def raise_(self):
        loc = (self.line, self.column) if self.has_location else None
        raise ParserError(self.message.decode('utf-8', 'replace'), loc)


class _CToken(C.Structure):
    _fields_ = [('kind', C.c_int32), ('line', C.c_uint32), ('column', C.c_uint32), ('number', C.c_double), ('text', C.c_char_p)]


class _COverrides(C.Structure):
    _fields_ = [('width', C.c_uint32), ('height', C.c_uint32), ('num_samples', C.c_uint32), ('max_depth', C.c_uint32)]


_LOADER_T = C.CFUNCTYPE(C.c_int, C.c_char_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p))
_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]


def _lib():
    L = backend.lib()
    if not getattr(L, '_cry_ready', False):
        L.cray_cry_tokenize.argtypes = [C.c_char_p, C.POINTER(C.POINTER(_CToken)), C.POINTER(C.c_size_t), C.POINTER(_CErr)]
        L.cray_cry_free_tokens.argtypes = [C.POINTER(_CToken), C.c_size_t]
        L.cray_cry_parse_value.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(_CErr)]
        L.cray_cry_free_string.argtypes = [C.c_void_p]
        L.cray_cry_parse_scene.argtypes = [C.c_char_p, C.c_char_p, _LOADER_T, C.c_void_p, C.POINTER(_COverrides),
                                           C.POINTER(C.c_void_p), C.POINTER(_CErr)]
        L.cray_owned_scene_desc.restype = C.POINTER(CSceneDesc)
        L.cray_owned_scene_desc.argtypes = [C.c_void_p]
        L.cray_owned_scene_warnings.restype = C.c_uint32
        L.cray_owned_scene_warnings.argtypes = [C.c_void_p]
        L.cray_owned_scene_free.argtypes = [C.c_void_p]
        L._cry_ready = True
    return L


def tokenize(text):
    """-> list of (kind name, value, (line, column)); raises ParserError."""
    L = _lib()
    toks, n, err = C.POINTER(_CToken)(), C.c_size_t(), _CErr()
    if L.cray_cry_tokenize(text.encode('utf-8'), C.byref(toks), C.byref(n), C.byref(err)) != 0:
        err.raise_()
    out = []
    for i in range(n.value):
        t = toks[i]
        name = TOKEN_NAMES[t.kind]
        value = t.number if name == 'Number' else (t.text.decode('utf-8') if t.text is not None else None)
        out.append((name, value, (t.line, t.column)))
    L.cray_cry_free_tokens(toks, n)
    return out


def parse_value(text):
    """RawValue::from_tokens over tokenize(text); returns the canonical dump (see cray_cry.h)."""
    L = _lib()
    s, err = C.c_void_p(), _CErr()
    if L.cray_cry_parse_value(text.encode('utf-8'), C.byref(s), C.byref(err)) != 0:
        err.raise_()
    out = C.string_at(s).decode('utf-8')
    L.cray_cry_free_string(s)
    return out


def _default_image_loader(path):
    from PIL import Image   # `image::io::Reader::open(path).decode().to_rgb8()` (obj.rs:22-23, texture.rs:57-58)
    return np.asarray(Image.open(path).convert('RGB'), dtype=np.uint8)


class ParsedScene:
    def __init__(self, handle, loader_ref):
        self._h, self._loader_ref = handle, loader_ref
        self._desc = _lib().cray_owned_scene_desc(handle).contents
        self.num_samples, self.max_depth = self._desc.num_samples, self._desc.max_depth
        self.lights = [None] * self._desc.n_lights
        self.warnings = _lib().cray_owned_scene_warnings(handle)

    def desc(self):
        return self._desc

    def film_bounds(self):
        return self._desc.camera.film_width, self._desc.camera.film_height

    @property
    def n_prims(self):
        return self._desc.n_prims

    def close(self):
        if self._h:
            _lib().cray_owned_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def parse_scene(text, base_dir=None, width=0, height=0, num_samples=0, max_depth=0, image_loader=_default_image_loader):
    """scene_parser::parse_scene; the keyword overrides are this build's CLI additions."""
    L = _lib()

    def load(path, _user, w, h, px):
        try:
            img = np.ascontiguousarray(image_loader(path.decode('utf-8')), dtype=np.uint8)
            buf = _libc.malloc(img.size)
            C.memmove(buf, img.ctypes.data, img.size)
            w[0], h[0], px[0] = img.shape[1], img.shape[0], buf
            return 0
        except Exception:
            return 1

    # image_loader=None: no callback, the library decodes PNM / JPEG itself (cray_load_image, include/cray_io.h)
    cb = _LOADER_T(load) if image_loader is not None else C.cast(None, _LOADER_T)
    ov = _COverrides(width, height, num_samples, max_depth)
    h, err = C.c_void_p(), _CErr()
    rc = L.cray_cry_parse_scene(text.encode('utf-8'), base_dir.encode('utf-8') if base_dir else None, cb, None,
                                C.byref(ov), C.byref(h), C.byref(err))
    if rc != 0:
        err.raise_()
    return ParsedScene(h, cb)


def load_scene_file(path, **kw):
    with open(path) as f:
        text = f.read()
    return parse_scene(text, base_dir=kw.pop('base_dir', None) or os.getcwd(), **kw)
