import sys, numpy as np, torch, ctypes
sys.path.insert(0,'.')
from tiler_amd import synth, stages
from tiler_amd.encoder import TilingEncoder
from tests.oracle_binding import Oracle
o = Oracle('oracle/libtm_oracle.so')
fr = synth.video(6, 100, 52, cut=4)
tiles, flags, lab = stages.load(torch.from_numpy(fr.view(np.int32)).cuda(), 13, 7)
lab = lab.cpu().numpy().reshape(6, -1)
enc = TilingEncoder(); enc.LoadDefaultSettings(); enc.SetVideo(100,52,24.0,6)
for f in range(6): enc.PushFrame(f, fr[f])
enc.Run(0)
got = enc.FrameCorrelations()
for f in range(1,6):
    x, y = lab[f-1], lab[f]
    exp = o.pearson(x, y)
    # numpy float32 sequential
    mx = np.float32(np.sum(x.astype(np.float64))/x.size); my = np.float32(np.sum(y.astype(np.float64))/y.size)
    num=np.float32(0); dx2=np.float32(0); dy2=np.float32(0)
    for i in range(x.size):
        dx=np.float32(x[i]-mx); dy=np.float32(y[i]-my)
        num=np.float32(num+np.float32(dx*dy)); dx2=np.float32(dx2+np.float32(dx*dx)); dy2=np.float32(dy2+np.float32(dy*dy))
    ref = np.float32(num/np.float32(np.sqrt(dx2)*np.sqrt(dy2)))
    print(f, got[f].view(np.uint32) if hasattr(got[f],'view') else got[f], np.float32(exp).view(np.uint32), ref.view(np.uint32), float(got[f]), exp)
