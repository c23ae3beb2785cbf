"""Imported first by the tools that use experiment knobs (RTP_* environment variables, probes, ablation variants): they run against
the EXPERIMENTS build of the library (caffe_rtpose_amd/librtpose_mi355x_exp.so, -DRTP_EXPERIMENTS) unless the caller chose one with
RTP_LIB.  The production library does not know the knobs (tests/test_host_cpu.py::test_production_library_has_no_experiment_knobs)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RTP_LIB", os.path.join(ROOT, "caffe_rtpose_amd", "librtpose_mi355x_exp.so"))
