import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ctu_common as cc, flatapi, kvazaar_amd
lib = kvazaar_amd.load_library()
w, h = 3840, 2160
model = cc.hip_cost_model(lib, 22)
frames = cc.yuv_frames(w, h, 2, 2, "large")
def run(fs):
    b = cc.HipBatch(lib, w, h, len(fs))
    for i, f in enumerate(fs): b.upload(i, f)
    b.run(model); out = [b.download(i) for i in range(len(fs))]; b.close(); return out
wc = (w + 63) // 64
batch = run(frames); again = run(frames); alone = run(frames[1:])
for name, x, y in (("again0", batch[0], again[0]), ("again1", batch[1], again[1]), ("alone", batch[1], alone[0])):
    d = np.nonzero(x["cost"] != y["cost"])[0]
    print(name, cc.compare(x, y), "n diff ctus", len(d), "first", [(int(i % wc), int(i // wc)) for i in d[:12]])
