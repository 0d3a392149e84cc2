import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle_lib as O
from orb_slam2_amd import capi, synth
img = synth.textured_image(1241,376,synth.BASE_SEED+3)
g=capi.ORBextractor(2000,1.2,8,20,7); o=O.OracleExtractor(2000,1.2,8,20,7)
g(img); o.extract(img)
for l in range(8):
    cg,co=g.debug_stage(l,2),o.candidates(l)
    same = cg.shape==co.shape and np.array_equal(cg,co)
    print(l,cg.shape,co.shape,same)
    if not same:
        sg=set(map(tuple,cg.astype(int))); so=set(map(tuple,co.astype(int)))
        print(' only gpu',sorted(sg-so)[:8],len(sg-so),' only oracle',sorted(so-sg)[:8],len(so-sg))
        if cg.shape==co.shape:
            d=np.nonzero((cg!=co).any(1))[0]; print(' first diff rows',d[:5], cg[d[:3]], co[d[:3]])
        break
