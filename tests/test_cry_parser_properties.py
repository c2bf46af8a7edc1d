"""Property tests of the .cry reader (include/cray_cry.h): whatever text comes in, the tokenizer / value parser /
scene builder either return a result or a ParserError — they never crash, hang or report an out-of-range
location — and generated well-formed values are accepted."""
import string

import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

from craytracer_amd import cry

ALPHABET = string.ascii_letters + string.digits + " \n\t{}[]():,.'-_#/\"\\é"
SETTINGS = dict(max_examples=300, deadline=None, suppress_health_check=[HealthCheck.too_slow])


def location_in_range(err, text):
    if err.location is None:
        return True
    line, col = err.location
    n_lines = text.count('\n') + 1
    return 1 <= line <= n_lines + 1 and col >= 1


@settings(**SETTINGS)
@given(st.text(alphabet=ALPHABET, max_size=200))
def test_tokenizer_and_parsers_are_total(text):
    for fn in (cry.tokenize, cry.parse_value, cry.parse_scene):
        try:
            fn(text)
        except cry.ParserError as e:
            assert e.message
            assert location_in_range(e, text), (fn.__name__, e.location, text)


def values(depth=3):
    num = st.one_of(st.integers(-10**6, 10**6).map(str),
                    st.floats(-1e6, 1e6, allow_nan=False, allow_infinity=False).map(lambda x: ('%.6f' % x)))
    ident = st.text(alphabet=string.ascii_lowercase + '_', min_size=1, max_size=8)
    s = st.text(alphabet=string.ascii_letters + ' ./_', max_size=10).map(lambda t: "'%s'" % t)
    leaf = st.one_of(num, s)
    if depth == 0:
        return leaf
    sub = values(depth - 1)
    arr = st.lists(sub, max_size=3).map(lambda xs: '[' + ', '.join(xs) + ']')
    mp = st.lists(st.tuples(ident, sub), max_size=3).map(lambda kv: '{' + ', '.join('%s: %s' % p for p in kv) + '}')
    typed = st.tuples(ident.map(str.capitalize), mp).map(lambda p: '%s %s' % p)
    vec = st.tuples(ident.map(str.capitalize), st.lists(num, min_size=1, max_size=3)).map(lambda p: '%s(%s)' % (p[0], ', '.join(p[1])))
    return st.one_of(leaf, arr, mp, typed, vec)


@settings(**SETTINGS)
@given(values())
def test_generated_values_parse_to_a_dump(text):
    """Values built from the grammar (numbers, strings, arrays, maps, typed maps, vectors; nested) either parse to a
    non-empty canonical dump or are rejected with a ParserError (duplicate keys), deterministically."""
    def run():
        try:
            return ('ok', cry.parse_value(text))
        except cry.ParserError as e:
            return ('err', e.message, e.location)
    a, b = run(), run()
    assert a == b
    if a[0] == 'ok':
        assert isinstance(a[1], str) and a[1]
