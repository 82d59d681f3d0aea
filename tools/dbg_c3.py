import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import helpers
from fdoct_amd import Config, Reconstructor, synth
W,H,N,D=2048,1000,2048,1024
cfg=Config(width=W,height=H,numfftpoints=N,numdisplaypoints=D)
frames=synth.make_frames(40,1,W,H); yb=synth.make_background(W)
win,ph=synth.hann_window(W),synth.dispersion_phase(N)
for phase in (ph,None):
    r=Reconstructor(cfg); r.set_background(yb); r.set_window(win)
    if phase is not None: r.set_dispersion_phase(phase)
    b,_=r.process(frames)
    r.set_plan(-1,True); bg,_=r.process(frames)
    bf,_=r.process(frames.astype(np.float32))
    r.close()
    mo,_,_=helpers.oracle_reference(cfg,frames,yb,window=win,phase=phase)
    rm=mo.max(axis=-1,keepdims=True)
    for name,x in (("lean",b),("general",bg),("general f32 in",bf)):
        e=np.abs(x-mo)/(1e-4*np.abs(mo)+1e-6*rm)
        rows=np.nonzero(e.max(axis=-1)[0]>1)[0]
        print("phase" if phase is not None else "real", name, "worst", e.max(), "bad rows", len(rows), rows[:10], "argmax bin", np.unravel_index(e.argmax(), e.shape))
