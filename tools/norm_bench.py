import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from fdoct_amd import Config, Reconstructor, synth, DTYPE_U16, VARIANT_SIM
W,H,N,D=2048,1000,2048,1024
nf=262
frames=np.tile(synth.make_frames(0,2,W,H),(nf//2,1,1))
d_in=torch.from_numpy(frames.view(np.int16)).cuda()
d_out=torch.empty((nf,H,D),dtype=torch.float32,device='cuda')
for name,kw in (("main donotnormalize=1",dict(donotnormalize=1)),("main donotnormalize=0",dict(donotnormalize=0)),("sim",dict(variant=VARIANT_SIM)),("rowwisenormalize",dict(rowwisenormalize=1))):
    r=Reconstructor(Config(width=W,height=H,numfftpoints=N,numdisplaypoints=D,**kw))
    r.set_background(synth.make_background(W))
    st=torch.cuda.Stream(); torch.cuda.synchronize(); r.set_stream(st.cuda_stream)
    for i in range(300): r.process_device(d_in.data_ptr(),DTYPE_U16,nf,W*2,None,d_out.data_ptr())
    r.synchronize(); t0=time.perf_counter()
    for i in range(200): r.process_device(d_in.data_ptr(),DTYPE_U16,nf,W*2,None,d_out.data_ptr())
    r.synchronize(); dt=(time.perf_counter()-t0)/200
    print("%-24s %.3f ms  %.1f M A-scans/s"%(name,dt*1e3,nf*H/dt/1e6)); r.close()
