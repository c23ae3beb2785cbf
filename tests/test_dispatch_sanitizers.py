"""Race / memory-error detection on the HOST side of rtpose.bin (SURVEY section 5 "race detection / sanitizers").

rtpose_main.cpp (producer + decode pool, ONE shared queue, N workers, re-orderer, JSON writers, JPEG encoders) and the host-only sources
of the library (host_util.cpp, preprocess.cpp, codecs.cpp) are compiled with g++ -fsanitize=thread and -fsanitize=address,undefined
against tests/helpers/engine_stub.cpp (link-time stand-ins for the device entry points: no HIP runtime in the process) and driven in
--dry_engine mode — the workers cost the host what an engine costs it and complete at a fixed rate — through every thread structure the
CLI has: 8 workers, frame drops (rtpose.cpp:1112-1124 and the re-orderer's skip, :1227-1231), --json_writers, --write_frames (encoder
pool), --producer_threads, --image_dir (decode pool: JPEG + PNG), --host_preprocess.  Zero sanitizer reports is the assertion.

The reference's own acknowledged races stay out by construction: rtpose.cpp:104 (unsynchronised quit flags: std::atomic here), :319 /
:441 (UI state read by the producer without a lock: no UI thread here), :1450 (global counters written by the display thread and read
elsewhere: atomics)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "caffe_rtpose_amd", "csrc")
SOURCES = [os.path.join(CSRC, f) for f in ("rtpose_main.cpp", "host_util.cpp", "preprocess.cpp", "codecs.cpp")] + [os.path.join(ROOT, "tests", "helpers", "engine_stub.cpp")]
REPORT = re.compile(r"ThreadSanitizer|AddressSanitizer|LeakSanitizer|runtime error:|UndefinedBehaviorSanitizer")


def _build(tmp, name, flags):
    exe = os.path.join(tmp, name)
    p = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer"] + flags + ["-o", exe] + SOURCES + ["-lpthread", "-lz"],
                       capture_output=True, text=True, timeout=600)
    if p.returncode != 0 and ("cannot find" in p.stderr or "unrecognized" in p.stderr):
        pytest.skip(f"this toolchain has no {flags}: {p.stderr[-200:]}")
    assert p.returncode == 0, p.stderr[-3000:]
    return exe


def _images(d, n=24):
    from PIL import Image
    os.makedirs(d, exist_ok=True)
    rs = np.random.RandomState(1)
    for i in range(n):
        a = np.kron(rs.randint(0, 256, (12, 20, 3)), np.ones((8, 8, 1))).astype(np.uint8)
        Image.fromarray(a).save(os.path.join(d, f"im{i:03d}." + ("jpg" if i % 2 else "png")))
    return n


def _runs(tmp):
    """(name, extra flags, frames expected to be written or None when drops are allowed)"""
    img = os.path.join(tmp, "imgs")
    n_img = _images(img)
    syn = ["--video", "synthetic:320x180:240", "--resolution", "320x180"]
    return [
        ("eight_workers_json_writers", syn + ["--num_gpu", "8", "--dry_engine", "60", "--json_writers", "4", "--producer_threads", "4", "--no_frame_drops", "--dry_people", "5"], 240),
        ("frame_drops_one_slow_worker", syn + ["--num_gpu", "2", "--dry_engine", "15", "--json_writers", "2"], None),
        ("write_frames_encoders", syn + ["--num_gpu", "4", "--dry_engine", "60", "--no_frame_drops", "--write_frames", os.path.join(tmp, "jpg")], 240),
        ("single_writer_thread", syn + ["--num_gpu", "3", "--dry_engine", "80", "--json_writers", "0", "--no_frame_drops", "--frames_in_flight", "2"], 240),
        ("image_dir_decode_pool", ["--image_dir", img, "--resolution", "160x96", "--num_gpu", "4", "--dry_engine", "60", "--no_frame_drops", "--producer_threads", "3"], n_img),
        ("host_preprocess_3_scales", syn + ["--net_resolution", "160x96", "--num_scales", "3", "--scale_gap", "0.15", "--start_scale", "0.8", "--host_preprocess", "--num_gpu", "2",
                                            "--dry_engine", "40", "--no_frame_drops"], 240),
    ]


@pytest.mark.parametrize("kind,flags,env", [
    ("tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=0 second_deadlock_stack=1"}),
    ("asan_ubsan", ["-fsanitize=address,undefined"], {"ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "print_stacktrace=1"}),
])
def test_host_dispatcher_is_clean_under_sanitizers(kind, flags, env, tmp_path):
    tmp = str(tmp_path)
    exe = _build(tmp, "rtpose_" + kind, flags)
    for name, extra, want in _runs(tmp):
        out = os.path.join(tmp, kind + "_" + name)
        p = subprocess.run([exe, "--model", "coco", "--write_json", out, "--no_display"] + extra, capture_output=True, text=True, timeout=600, env={**os.environ, **env})
        reports = [ln for ln in p.stderr.splitlines() if REPORT.search(ln)]
        assert not reports, f"{kind} / {name}:\n" + "\n".join(p.stderr.splitlines()[-60:])
        assert p.returncode == 0, f"{kind} / {name}: rc {p.returncode}\n{p.stderr[-1500:]}"
        m = re.search(r"frames produced (\d+), written (\d+), dropped (\d+)", p.stderr)
        assert m, p.stderr[-500:]
        produced, written, dropped = (int(v) for v in m.groups())
        assert written + dropped == produced and len(os.listdir(out)) == written
        if want is not None:
            assert written == want and dropped == 0, (name, produced, written, dropped)
        if name == "write_frames_encoders":
            assert len(os.listdir(os.path.join(tmp, "jpg"))) == written


def test_the_harness_sees_a_planted_race(tmp_path):
    """Negative control: the round-4 race (every worker writing a plain int that the JSON writers read, rtpose_main.cpp `G.num_parts`) put
    back into a copy of the source IS reported — an always-green sanitizer job proves nothing."""
    tmp = str(tmp_path)
    src = open(os.path.join(CSRC, "rtpose_main.cpp")).read()
    racy = src.replace("std::atomic<int> num_parts{18};", "int num_parts = 18;").replace("G.num_parts.store(num_parts);", "G.num_parts = num_parts;") \
              .replace("G.num_parts.load()", "G.num_parts").replace('"../../include/rtpose_mi355x.h"', '"' + os.path.join(ROOT, "include", "rtpose_mi355x.h") + '"')
    assert racy.count("int num_parts = 18;") == 1 and "G.num_parts = num_parts;" in racy
    path = os.path.join(tmp, "main_racy.cpp")
    open(path, "w").write(racy)
    exe = os.path.join(tmp, "racy")
    p = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-o", exe, path] + SOURCES[1:] + ["-lpthread", "-lz"], capture_output=True, text=True, timeout=600)
    if p.returncode != 0 and "cannot find" in p.stderr:
        pytest.skip("no libtsan")
    assert p.returncode == 0, p.stderr[-2000:]
    seen = False
    for _ in range(3):   # a race report needs the two accesses to actually overlap
        q = subprocess.run([exe, "--video", "synthetic:320x180:200", "--resolution", "320x180", "--model", "coco", "--write_json", os.path.join(tmp, "j"), "--no_display",
                            "--num_gpu", "8", "--dry_engine", "60", "--json_writers", "4"], capture_output=True, text=True, timeout=300,
                           env={**os.environ, "TSAN_OPTIONS": "halt_on_error=0"})
        if "ThreadSanitizer: data race" in q.stderr and "::G'" in q.stderr:   # "Location is global '(anonymous namespace)::G'"
            seen = True
            break
    assert seen
