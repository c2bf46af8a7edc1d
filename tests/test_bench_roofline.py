"""bench.py's `roofline.traffic` comes from a committed counter profile; it must be REFUSED unless that profile is of the very
build of the kernels the process runs (round 2 reported the counters of another build next to HEAD's timings)."""
import json

import bench


def test_traffic_is_refused_for_another_build(tmp_path):
    p = tmp_path / 'hbm_traffic.json'
    p.write_text(json.dumps({'dragon': {'source': 'profiles/x/pmc_fetch_size.csv', 'git': 'abc1234', 'source_hash': 'a' * 64,
                                        'trace_bytes_per_frame': 900e9, 'shade_bytes_per_frame': 70e9}}))
    t, prov, ent = bench.committed_traffic('dragon', 'a' * 64, 9.0, path=str(p), environ={})
    assert t == 100e9 and 'refused' not in prov and prov['profiled_source_hash'] == 'a' * 16
    t, prov, ent = bench.committed_traffic('dragon', 'b' * 64, 9.0, path=str(p))
    assert t is None and 'refused' in prov and prov['library_source_hash'] == 'b' * 16
    # a profile without a hash (the round-2 format) is never trusted; other workloads, N > 1 and the fast mode have no figure
    p.write_text(json.dumps({'dragon': {'source': 'x', 'git': 'abc', 'trace_bytes_per_frame': 900e9}}))
    t, prov, _ = bench.committed_traffic('dragon', 'a' * 64, 9.0, path=str(p))
    assert t is None and 'refused' in prov
    assert bench.committed_traffic('cornell', 'a' * 64, 9.0, path=str(p))[0] is None
    p.write_text(json.dumps({'dragon': {'source_hash': 'a' * 64, 'trace_bytes_per_frame': 900e9}}))
    t, prov, _ = bench.committed_traffic('dragon', 'a' * 64, 9.0, world=2, path=str(p))
    assert t is None and 'N=1 profile only' in prov['refused']
    # since round 4 the key is the hash of the device-side sources; it takes precedence over the older all-sources hash
    p.write_text(json.dumps({'dragon': {'source_hash': 'c' * 64, 'kernel_hash': 'a' * 64, 'trace_bytes_per_frame': 900e9}}))
    assert bench.committed_traffic('dragon', 'a' * 64, 9.0, path=str(p), environ={})[0] == 100e9
    assert bench.committed_traffic('dragon', 'c' * 64, 9.0, path=str(p), environ={})[0] is None
    assert bench.committed_traffic('dragon', 'a' * 64, 9.0, precision='f32', path=str(p))[0] is None


def test_traffic_is_refused_for_other_instantiations_than_the_profiled_ones(tmp_path):
    """Which kernels run is decided at run time: the records a scene's frames read, CRAY_LIB, the instantiation switches.  The counter
    passes were pinned to one choice (tools/profile_round.sh); a run that reads other records, loads another library or flips a switch
    gets no traffic figure."""
    p = tmp_path / 'hbm_traffic.json'
    rec = {'bounce0': 'certified f32 culling', 'other_launches': 'certified f32 culling'}
    p.write_text(json.dumps({'dragon': {'kernel_hash': 'a' * 64, 'trace_bytes_per_frame': 900e9, 'records': rec}}))
    ok = lambda **kw: bench.committed_traffic('dragon', 'a' * 64, 9.0, path=str(p), **kw)
    assert ok(records=rec, environ={})[0] == 100e9
    t, prov, _ = ok(records={'bounce0': 'certified f32 culling', 'other_launches': 'f64'}, environ={})
    assert t is None and 'other traversal records' in prov['refused']
    for var in bench.INSTANTIATION_ENV:
        t, prov, _ = ok(records=rec, environ={var: '1'})
        assert t is None and var in prov['refused'], var
    assert ok(records=rec, environ={'CRAY_RECORDS_B0': '1', 'CRAY_RECORDS_REST': '1'})[0] == 100e9   # a pin is checked through the records themselves
    # an entry written before round 5 names no records: nothing to compare (the hash rule still applies)
    p.write_text(json.dumps({'dragon': {'kernel_hash': 'a' * 64, 'trace_bytes_per_frame': 900e9}}))
    assert ok(records=rec, environ={})[0] == 100e9


def test_the_committed_profile_is_of_the_committed_kernels():
    """profiles/hbm_traffic.json[dragon] must carry the source hash of the kernels in the tree: a kernel change without a new
    counter pass would make the driver's bench line report `frac: null`."""
    from craytracer_amd import build
    import os
    path = os.path.join(bench.ROOT, 'profiles', 'hbm_traffic.json')
    for wl in ('dragon', 'cornell', 'staircase'):   # every GPU config of BASELINE.json has its counter profile (round 4)
        ent = json.load(open(path))[wl]
        assert ent.get('kernel_hash') == build.kernel_hash(), '%s: re-run tools/profile_round.sh + tools/adopt_profile.sh for the current kernels' % wl
        t, prov, _ = bench.committed_traffic(wl, build.kernel_hash(), 9.0, environ={})
        assert t and 'refused' not in prov


def test_an_edit_to_a_host_side_source_does_not_orphan_the_profile():
    """The profile key covers what the GPU code is compiled from, not the parsers and decoders of the same library."""
    from craytracer_amd import build
    assert 'cray_image.cpp' not in build.KERNEL_SOURCES and 'cray_cry.cpp' not in build.KERNEL_SOURCES and 'cray_host.cpp' not in build.KERNEL_SOURCES
    assert {'cray_hip.hip', 'cray_kernels.h', 'cray_shading.h', 'cray_math.h', 'cray_device.h', 'cray_bvh_build.h'} <= set(build.KERNEL_SOURCES)
    assert set(build.KERNEL_SOURCES) <= set(build.SOURCES + build.HEADERS)
