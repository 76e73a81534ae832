import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tadataka_amd import _lib, ops, synthetic
import bench
B,H,W=256,480,640
cam=synthetic.camera_for(W,H)
bt=ops.DvoBatch(B,H,W,n_levels=3,ratio=1.5)
SEED=int(os.environ.get("SEED","0")); bt.fill_synthetic(cam, bench.true_poses(B,SEED), seed0=SEED, noise=0.02)
bt.build_pyramid()
ident=np.tile(ops.pose12(np.eye(3),np.zeros(3)),(B,1))
for lvl in (2,1,0):
    f0=bt.tukey_fallbacks()
    t0=time.perf_counter()
    for _ in range(5): ev=bt.evaluate(lvl,cam,cam,ident,ops.W_TUKEY)
    dt=(time.perf_counter()-t0)/5
    print("level",lvl,"eval %.3f ms"%(dt*1e3),"fallbacks per eval",(bt.tukey_fallbacks()-f0)/5)
f0=bt.tukey_fallbacks()
t0=time.perf_counter()
for _ in range(3): P,px=bt.estimate(cam,cam,ident,ops.W_TUKEY,20)
dt=(time.perf_counter()-t0)/3
print("estimate %.3f ms"%(dt*1e3),"fallbacks per estimate",(bt.tukey_fallbacks()-f0)/3, "evals px", px/(H*W*B))
