#!/usr/bin/env python3
"""Developer tool (GPU): one large picture WITHOUT WPP -- the longest serial chain of CTUs and so the longest waits the ticket
schedule sees (kvz_batch.hpp scales its wait bound with the picture for this mode) -- against the oracle."""
import sys, time; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import ctu_common as cc, flatapi, kvazaar_amd
lib=kvazaar_amd.load_library(); orc=flatapi.load_oracle()
for (w,h) in [(1920,1080),(3840,2160)]:
    m=cc.hip_cost_model(lib,22,cc.coeff_weights(22)); m.no_wpp=1
    f=cc.yuv_frames(w,h,1,3,"large")[0]
    b=cc.HipBatch(lib,w,h,1); b.upload(0,f); t=time.time(); b.run(m); dt=time.time()-t; got=b.download(0); b.close()
    want=cc.run_oracle(orc,m,w,h,f)
    print(w,h,"single picture, no WPP:", round(dt,2),"s on the GPU; differs in", cc.compare(want,got))
