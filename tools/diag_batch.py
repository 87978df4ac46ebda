import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ctu_common as cc, flatapi, kvazaar_amd
lib = kvazaar_amd.load_library()
w, h = int(sys.argv[1]), int(sys.argv[2])
model = cc.hip_cost_model(lib, 22)
frames = cc.yuv_frames(w, h, 2, 2, "large")
def run(fs):
    b = cc.HipBatch(lib, w, h, len(fs))
    for i, f in enumerate(fs): b.upload(i, f)
    b.run(model); out = [b.download(i) for i in range(len(fs))]; b.close(); return out
batch = run(frames); a0 = run(frames[:1]); a1 = run(frames[1:])
wc = (w + 63) // 64
for name, x, y in (("f0", batch[0], a0[0]), ("f1", batch[1], a1[0])):
    d = np.nonzero(x["cost"] != y["cost"])[0]
    print(name, cc.compare(x, y), "n diff ctus", len(d), "first", [(int(i % wc), int(i // wc)) for i in d[:8]])
if len(sys.argv) > 3:
    o = flatapi.load_oracle()
    for i in range(2):
        want = cc.run_oracle(o, model, w, h, frames[i])
        for nm, got in (("batch", batch[i]), ("alone", (a0, a1)[i][0])):
            d = np.nonzero(want["cost"] != got["cost"])[0]
            print("oracle vs", nm, i, cc.compare(want, got), len(d), [(int(k % wc), int(k // wc)) for k in d[:8]])
