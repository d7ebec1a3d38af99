import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nu_scaler_amd as nsc
from nu_scaler_amd import hostmem
hostmem.route_tensor_cpu_through_pinned_staging()  # device -> pinned staging -> numpy (nu_scaler_amd/hostmem.py)
import oracle
w,h,ow,oh = 3840,2160,1920,1080
n = int(sys.argv[1]) if len(sys.argv)>1 else 1
frames = np.stack([oracle.gen_noise(w,h,300+k) for k in range(n)])
u = nsc.PyWgpuUpscaler("quality","lanczos3"); u.initialize(w,h,ow,oh)
print(u.kernel_variant)
d_in = torch.from_numpy(frames).cuda(); d_out = torch.zeros((n,oh,ow,4),dtype=torch.uint8,device="cuda")
for rep in range(3):
    d_out.zero_()
    u.upscale_device(d_in.data_ptr(), d_out.data_ptr(), n, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    want = oracle.resize(frames[0], ow, oh, 0, threads=16)
    d = np.abs(got[0].astype(int)-want.astype(int)).max(axis=2)
    bad = np.argwhere(d>1)
    print("rep",rep,"bad px",len(bad), "rows", np.unique(bad[:,0])[:40], "cols min/max", (bad[:,1].min(), bad[:,1].max()) if len(bad) else None)
    if len(bad):
        rows,counts = np.unique(bad[:,0], return_counts=True)
        print(" per-row counts", dict(zip(rows[:20].tolist(), counts[:20].tolist())))
        r0=bad[0][0]; cs=bad[bad[:,0]==r0][:,1]; print(" first bad row", r0, "cols", cs[:30], "...", cs[-5:])
