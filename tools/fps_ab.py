import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, stm_amd
from stm_amd import device_api as dev, synth
H,W,D,zd=1080,1920,64,32
sbs,_=synth.sbs_frame(H,W,D,zd); p=dev.FrameParams(num_disp=D,zero_disp=zd)
d_sbs=torch.from_numpy(sbs).cuda(); dl=torch.zeros(H,W,dtype=torch.float32,device='cuda'); dr=torch.zeros_like(dl); out=torch.zeros(H,W,3,dtype=torch.uint8,device='cuda')
lib=stm_amd.lib(); ref=None
for rnd in range(3):
    for v in [int(x) for x in sys.argv[1:]]:
        lib.stm_set_agg_variant(v)
        for _ in range(3): dev.d_adcensus_stm(d_sbs,dl,dr,out,p,stages=3)
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(20): dev.d_adcensus_stm(d_sbs,dl,dr,out,p,stages=3)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
        o=out.cpu().numpy().copy()
        if ref is None: ref=o
        assert np.array_equal(o,ref)
        print('variant',v,'ms/frame %.3f fps %.1f'%(dt*1e3,1/dt))
