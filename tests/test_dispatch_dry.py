"""Host side of --num_gpu N without GPUs: `rtpose.bin --dry_engine RATE` replaces every engine by a
stand-in that costs the host what rtp_submit_frame / rtp_collect cost it (the staging copy, the joints copy) and completes RATE
frames/s, so producer -> shared queue -> 8 workers -> re-orderer (window 4, rtpose.cpp:90,1214-1273) -> JSON writers can be driven
at the rate 8 GPUs deliver.  Here (8 container cores) the check is functional + a modest rate; profiles/r03_dry_scaling.txt holds the
same command on the GPU box's 128 host cores at 8 x the measured per-GPU rates."""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "caffe_rtpose_amd", "rtpose.bin")


def _run(tmp, *flags, frames=1200):
    out = os.path.join(tmp, "json")
    p = subprocess.run([BIN, "--video", f"synthetic:640x360:{frames}", "--resolution", "640x360", "--model", "coco", "--write_json", out, "--no_display",
                        "--frames_in_flight", "8"] + list(flags), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    m = re.search(r"frames produced (\d+), written (\d+), dropped (\d+) \(([\d.]+) FPS incl. init, ([\d.]+) FPS first frame", p.stderr)
    assert m, p.stderr[-500:]
    per_worker = [int(v) for v in re.findall(r"worker \d+ \(GPU \d+\) processed (\d+) frames", p.stderr)]
    return out, int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(5)), per_worker


def test_eight_dry_workers_share_one_queue_and_every_frame_is_written(tmp_path):
    out, produced, written, dropped, fps, per_worker = _run(str(tmp_path), "--num_gpu", "8", "--dry_engine", "300", "--no_frame_drops",
                                                            "--producer_threads", "4", "--dry_people", "5")
    assert produced == written == 1200 and dropped == 0
    files = sorted(os.listdir(out))
    assert files == [f"frame{i:06d}.json" for i in range(1200)]           # named by video frame number (rtpose.cpp:1388)
    body = json.load(open(os.path.join(out, files[777])))
    assert body["version"] == 0.1 and len(body["bodies"]) == 5 and len(body["bodies"][0]["joints"]) == 54
    assert len(per_worker) == 8 and sum(per_worker) == 1200
    assert min(per_worker) > 0.5 * 150 and max(per_worker) < 1.5 * 150      # dynamic pull keeps the workers level
    assert fps > 800, f"8 workers x 300 frames/s delivered only {fps} frames/s through the host pipeline"


def test_dry_workers_drop_aged_frames_like_processFrame(tmp_path):
    # one slow worker, no --no_frame_drops: frames older than 0.1 s at fetch are dropped and the re-orderer skips them (rtpose.cpp:1112-1124, 1227-1231)
    out, produced, written, dropped, fps, _ = _run(str(tmp_path), "--num_gpu", "1", "--dry_engine", "50", frames=300)
    assert produced == 300 and dropped > 0 and written + dropped == 300
    assert len(os.listdir(out)) == written
