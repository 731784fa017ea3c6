"""CPU tests: the C-ABI library loads and exports every symbol the header
declares, fails loudly without a device, and the host-side mirror of the
reference interface behaves like the reference (no compute without a GPU)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = os.path.join(ROOT, "include", "audiomatch.h")


def header_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(am_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree(amlib):
    assert header_symbols() == amlib.declared_symbols()


def test_library_exports_every_declared_symbol(amlib):
    lib = C.CDLL(amlib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", amlib.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (am_[a-z0-9_]+)\b", out))
    assert set(header_symbols()) <= exported
    assert amlib.lib().am_abi_version() == 3


def test_header_cites_reference_for_every_entry_point():
    txt = open(HEADER).read()
    assert txt.count("audio_matcher.rs") >= 12 and "mp3_reader.rs" in txt and "matcher/mod.rs" in txt


def test_pure_host_entry_points(amlib):
    n = C.c_size_t(0)
    L = amlib.lib()
    assert L.am_correlate_len(20, 3, int(amlib.Mode.Valid), C.byref(n)) == 0 and n.value == 18
    assert L.am_correlate_len(20, 3, int(amlib.Mode.Full), C.byref(n)) == 0 and n.value == 22
    assert L.am_correlate_len(20, 3, int(amlib.Mode.Same), C.byref(n)) == 0 and n.value == 20
    assert L.am_correlate_len(3, 7, int(amlib.Mode.Valid), C.byref(n)) == 0 and n.value == 1   # saturating_sub + 1
    assert L.am_correlate_len(0, 3, 2, C.byref(n)) == amlib.AM_ERR_INVALID_ARG
    assert L.am_set_option(b"log_n", 99) == amlib.AM_ERR_INVALID_ARG
    assert L.am_set_option(b"nonsense", 1) == amlib.AM_ERR_INVALID_ARG
    assert b"unknown" in L.am_last_error_string()
    assert L.am_set_option(b"log_n", 0) == 0
    assert amlib.get_option("pairs_per_group") >= 1
    # every documented option reads back what was set (and its default afterwards); values are clamped to their range
    for key, default, other in (("tail_block", 1, 0), ("host_pick_wait", 1, 0), ("profile_every", 1, 5), ("dense_scores", 0, 1),
                                ("device_redo", 1, 0), ("batch_overlap", 1, 0), ("k3_group", 1, 0), ("pick_group", 1, 0),
                                ("peak_filter_order", 0, 1), ("distance_rule", 0, 3), ("tail_window", 0, 1), ("surrounding_from", 0, 1),
                                ("debug_no_realloc", 0, 1), ("debug_redo_arm_at", -2, 3)):
        assert amlib.get_option(key) == default, key
        amlib.set_option(key, other)
        assert amlib.get_option(key) == other, key
        amlib.set_option(key, default)
        assert amlib.get_option(key) == default, key
    amlib.set_option("profile_every", 0)
    assert amlib.get_option("profile_every") == 1


def test_no_device_fails_loudly(amlib):
    """Without a GPU every compute entry point must return an error: no CPU fallback."""
    if amlib.device_count() > 0:
        pytest.skip("a GPU is visible; covered by the gpu tests")
    with pytest.raises(amlib.AudioMatchError) as ei:
        amlib.HipConvolve([1.0, 2.0, 3.0])
    assert ei.value.code == amlib.AM_ERR_NO_DEVICE
    with pytest.raises(amlib.AudioMatchError):
        amlib.find_peaks([0.0, 1.0, 0.0], 0.0)
    with pytest.raises(amlib.AudioMatchError):
        amlib.pcm_s16_stereo_to_mono(np.zeros(8, np.int16))


def test_product_never_touches_the_oracle():
    """The shipped sources must not reference oracle/ (checker only)."""
    for sub in ("audio-matcher_amd/csrc", "audio-matcher_amd/python", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".hip", ".h", ".hpp", ".py", ".cpp")):
                    txt = open(os.path.join(dirpath, f)).read()
                    assert "pyoracle" not in txt and "liboracle" not in txt and "orc_" not in txt, f


def test_config_mirrors_reference_defaults(amlib):
    fx = __import__("json").load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))["defaults"]
    cfg = amlib.Config(overlap_length_s=10.0)
    p = cfg.params(44100, amlib.Scale.LIB)
    assert abs(p.min_prominence - fx["prominence"]) < 1e-7
    assert p.chunk == fx["chunk_s"] * 44100 and p.overlap == 441000
    assert p.min_distance == fx["min_distance_samples_at_44100"]
    assert p.overshadow_distance_s == fx["distance_s"] and p.scale == 1
    # (secs * sr).round(): half away from zero (audio_matcher.rs:99-100)
    assert amlib.Config(chunk_size_s=0.5).params(3, 0).chunk == 2
    # distance.as_secs() truncates to whole seconds before * sr (audio_matcher.rs:228)
    assert amlib.Config(distance_s=2.9).params(100, 0).min_distance == 200


def test_cpp_host_mirror_compiles():
    """include/audiomatch.hpp (C++ mirror of CorrelateAlgo / calc_chunks) is valid C++17."""
    hpp = os.path.join(ROOT, "include", "audiomatch.hpp")
    src = '#include "audiomatch.hpp"\nint main() { audiomatch::Config c; (void)c; return 0; }\n'
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c++", "-"],
                   input=src, text=True, check=True)
    assert os.path.exists(hpp)


def test_host_rows_selftest():
    """parse_duration doctest vectors (src/args.rs:66-78), Arguments defaults, print_offsets and
    timelabel_from_peaks formatting of the C++ host code (no GPU needed)."""
    import build as am_build
    exe = am_build.build_selftest()
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "selftest ok" in out.stdout


def test_bench_and_smoke_refuse_to_run_without_a_gpu(amlib):
    """No CPU fallback anywhere on the product path: without a HIP device bench.py and
    smoke() stop with an explicit error instead of producing a number."""
    import subprocess
    if amlib.device_count() > 0:
        pytest.skip("a HIP device is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0",
                        "--ramp-steps", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs a HIP device" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs a HIP device" in r.stderr


def test_shard_plan_is_a_partition(amlib):
    """am_shard_plan (haystack k -> shard k mod n) needs no device: every item
    belongs to exactly one shard, shards differ in size by at most one."""
    for n_items in (0, 1, 7, 1000):
        for n_shards in (1, 2, 3, 8):
            seen = []
            counts = []
            for sh in range(n_shards):
                first, stride, count = amlib.shard_plan(n_items, n_shards, sh)
                assert (first, stride) == (sh, n_shards)
                seen += [first + i * stride for i in range(count)]
                counts.append(count)
            assert sorted(seen) == list(range(n_items)) and max(counts) - min(counts) <= 1
    with pytest.raises(amlib.AudioMatchError):
        amlib.shard_plan(10, 2, 2)
    with pytest.raises(amlib.AudioMatchError):
        amlib.shard_plan(10, 0, 0)
    from audiomatch_amd import sharding
    assert sharding.shard_indices(10, 1, 4) == [1, 5, 9]


def test_pool_needs_a_device(amlib):
    if amlib.device_count() > 0:
        pytest.skip("a HIP device is present; covered by the gpu tests")
    with pytest.raises(amlib.AudioMatchError) as ei:
        amlib.Pool([1.0, 2.0, 3.0])
    assert ei.value.code == amlib.AM_ERR_NO_DEVICE
    with pytest.raises(amlib.AudioMatchError):
        amlib.Pool([1.0, 2.0, 3.0], [0, 1])


def test_bench_gpus_flag_spawns_one_rank_per_gpu():
    """`bench.py --gpus 2 --dry-shard`: the parent starts two fresh rank processes (it makes
    no HIP call itself), every rank prints its shard of the 1000-haystack batch (k mod N) and
    -- on a box without a GPU -- stops with an explicit error instead of a number."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-shard"],
                       capture_output=True, text=True, timeout=300)
    plans = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert sorted(p["rank"] for p in plans) == [0, 1] and all(p["world"] == 2 for p in plans)
    assert [p["count"] for p in sorted(plans, key=lambda p: p["rank"])] == [500, 500]
    assert sorted(plans, key=lambda p: p["rank"])[1]["first"] == [1, 3, 5]
    sys.path.insert(0, os.path.join(ROOT, "audio-matcher_amd", "python"))
    import audiomatch_amd
    if audiomatch_amd.device_count() < 1:
        assert r.returncode != 0 and r.stderr.count("needs a HIP device") == 2
    # a rank count that disagrees with the launcher's world size is refused
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-shard"],
                       env=dict(os.environ, WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr


@pytest.mark.parametrize("config", [3, 4])
def test_bench_config_legs_shard_the_same_way(config):
    """`bench.py --config 3 / 4 --gpus 3 --dry-shard`: BASELINE configs[3] (32 needles) and
    configs[4] (48 kHz i16 stereo, f16) shard their haystacks k mod N like the headline."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(config), "--gpus", "3", "--dry-shard",
                        "--total-haystacks", "1000"], capture_output=True, text=True, timeout=300)
    plans = sorted((json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")), key=lambda p: p["rank"])
    assert [p["config"] for p in plans] == [config] * 3
    assert [p["count"] for p in plans] == [334, 333, 333] and [p["first"][0] for p in plans] == [0, 1, 2]
    assert [p["last"] for p in plans] == [999, 997, 998]


def test_bench_workloads_state_their_design_bytes():
    """The per-kernel algorithmic bytes bench.py prices each config's roofline on."""
    sys.path.insert(0, ROOT)
    import bench
    g = bench.plan_geometry(441000, 158760000)
    assert g == {"log_n": 22, "n_fft": 1 << 22, "hop": 3752960, "out_count": 158319001, "nblocks": 43, "npairs": 22, "tail_n_fft": 0}
    g4 = bench.plan_geometry(480000, 172800000)
    assert g4["npairs"] == 24 and g4["hop"] == 3714048 and g4["nblocks"] == 47
    # the odd last block on the smaller plan (option tail_block): 21 / 23 main pairs and one pair of the 2^21 plan
    gt = bench.plan_geometry(441000, 158760000, tail_block=1)
    assert (gt["npairs"], gt["tail_n_fft"], gt["nblocks"]) == (21, 1 << 21, 43)
    gt4 = bench.plan_geometry(480000, 172800000, tail_block=1)
    assert (gt4["npairs"], gt4["tail_n_fft"]) == (23, 1 << 21)
    assert bench.plan_geometry(441000, 158760000 + 3752960, tail_block=1)["tail_n_fft"] == 0          # 44 blocks: nothing odd
    assert bench.plan_geometry(441000, 42 * 3752960 + 441000 + 3400000, tail_block=1)["tail_n_fft"] == 0   # more than one 2^21 pair holds
    assert bench.plan_geometry(441000, 158760000, 22, tail_block=1)["tail_n_fft"] == 0                   # a forced plan is taken as it is


def test_isa_has_no_store_data_hazard(tmp_path):
    """The gfx950 store-data hazard of DESIGN.md section 3: no >64-bit store in the compiled
    kernels may have one of its data VGPRs overwritten after fewer than two wait states
    (tools/check_store_hazard.py on the ISA hipcc emits for every HIP source)."""
    sys.path.insert(0, os.path.join(ROOT, "audio-matcher_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build as am_build
    import check_store_hazard as chk
    from concurrent.futures import ThreadPoolExecutor

    def emit(src):
        out = str(tmp_path / (src + ".s"))
        subprocess.check_call([am_build._hipcc(), *[f for f in am_build.FLAGS if f != "-fPIC"], "--cuda-device-only", "-S",
                               os.path.join(am_build.CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
        return out
    with ThreadPoolExecutor(max_workers=len(am_build.SOURCES)) as ex:
        files = list(ex.map(emit, am_build.SOURCES))
    found = [hit for f in files for hit in chk.check(f)]
    assert found == [], found[:3]
    # the checker itself: every wide-store family is recognised, a safe distance is accepted
    bad = {
        "buffer_store_dwordx4 v[2:5], v1, s[0:3], s30 offen\n\tv_sub_f32 v4, v6, v7": [4],
        "global_store_dwordx4 v[8:9], v[2:5], off nt\n\tv_mov_b32 v2, 0": [2],
        "buffer_store_dwordx3 v[2:4], v1, s[0:3], 0 offen\n\ts_nop 0\n\tv_add_f32 v3, v3, v3": [3],
        "tbuffer_store_format_xyzw v[10:13], v1, s[0:3], s4 format:77 offen\n\tv_mov_b32 v13, v0": [13],
        "flat_store_dwordx4 v[0:1], v[20:23]\n\tv_fma_f32 v21, v1, v2, v3": [21],
        "global_atomic_cmpswap_x2 v[0:1], v[4:7], off\n\tv_mov_b32 v5, 1": [5],
    }
    ok = ["buffer_store_dwordx4 v[2:5], v1, s[0:3], s30 offen\n\ts_nop 1\n\tv_sub_f32 v4, v6, v7",
          "buffer_store_dwordx2 v[2:3], v1, s[0:3], s30 offen\n\tv_sub_f32 v2, v6, v7",
          "global_store_dwordx4 v[8:9], v[2:5], off\n\tv_mov_b32 v8, 0"]
    for i, (asm, regs_hit) in enumerate(bad.items()):
        f = tmp_path / f"bad{i}.s"
        f.write_text("kern:\n\t" + asm + "\n\ts_endpgm\n")
        hits = chk.check(str(f))
        assert len(hits) == 1 and hits[0][4] == regs_hit, (asm, hits)
    for i, asm in enumerate(ok):
        f = tmp_path / f"ok{i}.s"
        f.write_text("kern:\n\t" + asm + "\n\ts_endpgm\n")
        assert chk.check(str(f)) == [], asm


def test_long_plan_partitions_the_windows(amlib):
    """am_long_plan (one long haystack over several devices; audio_matcher.rs:104-131 fans the windows
    of ONE haystack out) needs no device: the parts own contiguous window ranges that cover every window with a valid
    lag exactly once, a part's samples reach to the end of its last window (chunk + overlap: the S - 1 halo and more),
    part sizes differ by one window at most; also with full-length windows only (option tail_window = 1)."""
    import random
    rnd = random.Random(7)
    for trial in range(300):
        chunk = rnd.randint(1, 50)
        overlap = rnd.randint(0, 60)
        s = rnd.randint(1, 70)
        n = rnd.randint(0, 700)
        n_parts = rnd.randint(1, 6)
        p = amlib.AmMatchParams(sr=1, chunk=chunk, overlap=overlap, min_prominence=0.1, min_distance=0,
                                overshadow_distance_s=1.0, scale=1)
        for tail in (0, 1):
            amlib.set_option("tail_window", tail)
            try:
                window = chunk + overlap
                valid = [i for i in range(0, (n + chunk - 1) // chunk)
                         if min(window, n - i * chunk) >= s and (not tail or n - i * chunk >= window)]
                assert valid == list(range(len(valid)))           # a prefix: only the windows at the end shrink
                plans = [amlib.long_plan(n, s, p, n_parts, k) for k in range(n_parts)]
                seen = []
                for w0, nw, a, cnt in plans:
                    seen += list(range(w0, w0 + nw))
                    if nw:
                        assert a == w0 * chunk and a + cnt == min(n, (w0 + nw - 1) * chunk + window)
                    else:
                        assert cnt == 0
                assert seen == valid, (n, s, chunk, overlap, n_parts, tail)
                sizes = [pl[1] for pl in plans]
                assert max(sizes) - min(sizes) <= 1
            finally:
                amlib.set_option("tail_window", 0)
    p = amlib.AmMatchParams(sr=1, chunk=10, overlap=2, min_prominence=0.1, min_distance=0, overshadow_distance_s=1.0, scale=1)
    with pytest.raises(amlib.AudioMatchError):
        amlib.long_plan(100, 5, p, 2, 2)
    with pytest.raises(amlib.AudioMatchError):
        amlib.long_plan(100, 0, p, 2, 0)


def test_merge_peaks_on_the_host_equals_the_checker(amlib, oracle):
    """am_merge_peaks = sort by start + filter_surrounding(!is_overshadowed) (audio_matcher.rs:132-160) is host code
    (no device needed): on random peak lists it keeps exactly what the checker's rule keeps, under both neighbour
    policies; the reference's truth table (K3) included."""
    import random
    rnd = random.Random(11)
    sr = 7
    for trial in range(200):
        n = rnd.randint(0, 12)
        peaks = [amlib.Peak(rnd.randint(0, 200), 0, rnd.random(), rnd.choice([0.1, 0.2, 0.2, 0.5, rnd.random()])) for _ in range(n)]
        for q in peaks:
            q.end = q.start + 1
        dist = rnd.choice([0.5, 1.0, 3.0, 10.0])
        p = amlib.AmMatchParams(sr=sr, chunk=10, overlap=2, min_prominence=0.1, min_distance=0, overshadow_distance_s=dist, scale=1)
        srt = sorted(peaks, key=lambda q: q.start)          # (stable: equal starts keep their order)
        tup = [(q.start, q.end, q.height, q.prominence) for q in srt]
        for from_filtered in (0, 1):
            keep, last = [], None
            for i, e in enumerate(tup):
                before = (last if from_filtered else (tup[i - 1] if i else None))
                after = tup[i + 1] if i + 1 < len(tup) else None
                if oracle.is_overshadowed(e, before, sr, dist) or oracle.is_overshadowed(e, after, sr, dist):
                    continue
                keep.append(e)
                last = e
            amlib.set_option("surrounding_from", from_filtered)
            try:
                got = amlib.merge_peaks(p, peaks)
            finally:
                amlib.set_option("surrounding_from", 0)
            assert [(g.start, g.end) for g in got] == [(e[0], e[1]) for e in keep], (trial, from_filtered)
