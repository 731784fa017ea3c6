"""End-to-end test of the C++ CLI (matcher::run, src/matcher/mod.rs:17-104) on WAV files:
stereo i16 ingest (GPU down-mix, mp3_reader.rs:28-37), match, `Offset i: ...` lines and the
Audacity label file (archive/data.rs:87-107)."""
import os
import subprocess
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR = 8000


def write_wav_stereo(path, lr):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(SR)
        w.writeframes(np.ascontiguousarray(lr, dtype="<i2").tobytes())


def make_case(oracle, tmp_path):
    rng = np.random.default_rng(5)
    s, h = 2 * SR, 70 * SR
    needle_lr = rng.integers(-8000, 8000, size=2 * s).astype(np.int16)
    hay_lr = rng.integers(-8000, 8000, size=2 * h).astype(np.int32)
    for t in (5.0, 31.0, 55.5):
        off = int(t * SR)
        hay_lr[2 * off:2 * (off + s)] += needle_lr
    hay_lr = np.clip(hay_lr, -32768, 32767).astype(np.int16)
    write_wav_stereo(tmp_path / "needle.wav", needle_lr)
    write_wav_stereo(tmp_path / "hay.wav", hay_lr)
    needle = oracle.pcm_s16_stereo_to_mono(needle_lr)
    hay = oracle.pcm_s16_stereo_to_mono(hay_lr)
    return needle, hay


def test_cli_end_to_end(gpu, oracle, tmp_path):
    import build as am_build
    cli = am_build.build_cli()
    needle, hay = make_case(oracle, tmp_path)
    exp = oracle.calc_chunks(SR, hay, needle, 20 * SR, needle.size, 0.13, 10 * SR, 10.0)
    assert [e[0] for e in exp] == [int(5.0 * SR), int(31.0 * SR), int(55.5 * SR)]
    out = subprocess.run([cli, str(tmp_path / "hay.wav"), "--snippet", str(tmp_path / "needle.wav"),
                          "--chunk-size", "20", "--distance", "10s", "-n"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("Offset")]
    assert len(lines) == 3
    for i, (line, e) in enumerate(zip(lines, exp), 1):
        secs = e[0] // SR
        assert line.startswith(f"Offset {i}: {secs // 3600:02d}:{secs // 60 % 60:02d}:{secs % 60:02d} with prominence ")
        assert abs(float(line.rsplit(" ", 1)[1]) - e[3]) < 1e-4
    label = (tmp_path / "hay.txt").read_text().splitlines()
    assert len(label) == 2
    for i, row in enumerate(label):
        a, b, name = row.split("\t")
        assert name == f"Segment {i + 1}"
        assert abs(float(a) - (exp[i][0] / SR + 7.0)) < 1e-5 and abs(float(b) - exp[i + 1][0] / SR) < 1e-5
    # --skip-existing leaves the file alone and produces no offsets
    out2 = subprocess.run([cli, str(tmp_path / "hay.wav"), "--snippet", str(tmp_path / "needle.wav"),
                           "--chunk-size", "20", "--distance", "10s", "--skip-existing"], capture_output=True, text=True)
    assert out2.returncode == 0 and "Offset" not in out2.stdout
    # --no-out --dry-run: offsets only
    os.remove(tmp_path / "hay.txt")
    out3 = subprocess.run([cli, str(tmp_path / "hay.wav"), "--snippet", str(tmp_path / "needle.wav"),
                           "--chunk-size", "20", "--distance", "10s", "--no-out", "--dry-run", "-n"],
                          capture_output=True, text=True)
    assert out3.returncode == 0 and out3.stdout.count("Offset") == 3 and not (tmp_path / "hay.txt").exists()


def test_cli_errors(gpu, tmp_path):
    import build as am_build
    cli = am_build.build_cli()
    r = subprocess.run([cli, "x.wav"], capture_output=True, text=True)
    assert r.returncode == 2 and "--snippet" in r.stderr
    r = subprocess.run([cli, "x.wav", "--snippet", "missing.wav", "-n"], capture_output=True, text=True)
    assert r.returncode == 1 and "couldn't find file" in r.stderr


def test_progress_callback(gpu, oracle):
    """f1/f2 of audio_matcher.rs:102-117,129, per haystack."""
    sr = 8000
    needle = oracle.synth_uniform(1, 0, 0, sr)
    hay = oracle.synth_uniform(1, 1, 0, 30 * sr)
    events = []
    gpu.set_progress_callback(lambda k, stage, n: events.append((k, stage, n)))
    try:
        cfg = gpu.Config(chunk_size_s=10.0, overlap_length_s=1.0)
        gpu.HipConvolve(needle).match(hay, cfg.params(sr, gpu.Scale.LIB))
    finally:
        gpu.set_progress_callback(None)
    assert events == [(0, 0, 3), (0, 1, 3)]
