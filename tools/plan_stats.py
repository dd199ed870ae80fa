import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, gnn_uds_amd as U, time, sys
from gnn_uds_amd import _lib
N,E=int(sys.argv[1]),int(sys.argv[2])
g=U.DrainageGraph.from_edges(U.synthetic_drainage_network(N,E,0))
t=int(sys.argv[3]) if len(sys.argv)>3 else 128
t0=time.time(); hdr,pool,caps=_lib.tile_plan(g,t,t,int(sys.argv[4]) if len(sys.argv)>4 else 128,int(sys.argv[5]) if len(sys.argv)>5 else 176); dt=time.time()-t0
print(caps, hdr.shape, 'plan time %.2f'%dt)
for side in (0,1):
    h=hdr[hdr[:,6]==side]
    print('side',side,'tiles',len(h),'own mean %.1f min %d max %d'%(h[:,0].mean(),h[:,0].min(),h[:,0].max()),'prim mean %.1f max %d'%(h[:,1].mean(),h[:,1].max()),'sec mean %.1f max %d'%(h[:,2].mean(),h[:,2].max()))
    print('  blocks: prim %.2f sec %.2f'%(np.ceil(h[:,1]/16).mean(), np.ceil(h[:,2]/16).mean()), 'halo frac %.3f'%((h[:,1].sum()-h[:,0].sum())/h[:,0].sum()), 'sec/own %.3f'%(h[:,2].sum()/h[:,0].sum()))
    print('  own hist', np.histogram(h[:,0],bins=[0,20,40,60,80,100,129])[0])
