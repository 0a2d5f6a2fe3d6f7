import sys, time; sys.path.insert(0,'/root/repo')
import torch
from touch_gs_amd import analytic_scene as A
t=time.time()
print(A.write_raw_capture('/tmp/capt', n_views=6, device='cuda', verbose=True), time.time()-t, flush=True)
t=time.time(); print(A.prepare_capture('/tmp/capt', 0.5), time.time()-t)
