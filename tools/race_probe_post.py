import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _exp  # noqa: E401,E402,F401  (experiments build of the library)
import caffe_rtpose_amd as r
kw = dict(net_w=320, net_h=176, num_scales=2, scale_gap=0.25, disp_w=640, disp_h=360)
A = r.Engine(r.Config(frames_in_flight=1, **kw))
B = r.Engine(r.Config(frames_in_flight=3, **kw))
img = r.synth_frame(800, 600, 0, seed=3)
x = r.preprocess_frame(img, 640, 360, 320, 176, 2, 1.0, 0.25)[0]
d = A.forward_debug(x)
low = d["lowres"]
pk0, j0, n0 = A.post_from_lowres(low)
stop = False
def load():
    pend = 0
    while not stop:
        B.submit(x, tag=0); pend += 1
        if pend == 3:
            B.collect(); pend -= 1
    while pend:
        B.collect(); pend -= 1
mode = sys.argv[1] if len(sys.argv) > 1 else "load"
t = threading.Thread(target=load)
if mode == "load":
    t.start()
bad = 0
for it in range(int(os.environ.get("PROBE_ITERS", "300"))):
    pk, j, n = A.post_from_lowres(low)
    if not np.array_equal(pk, pk0):
        bad += 1
        if bad <= 5:
            dd = np.argwhere(pk != pk0)
            print("iter", it, "peak diffs", len(dd), dd[:6].tolist(), [(float(pk[tuple(b)]), float(pk0[tuple(b)])) for b in dd[:3]])
stop = True
if mode == "load":
    t.join()
print(mode, "bad taps:", bad, "of", os.environ.get("PROBE_ITERS", "300"))
