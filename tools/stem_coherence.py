"""Would spatially coherent row tiles let a dense-over-offsets stem kernel skip empty K-chunks?  (VERDICT round 3, item 4 ii.)
For one synthetic 16 000-point plot: the 7^3 kernel map, and for tiles of 16 / 32 / 64 consecutive rows under four row orders
(input = z-major level order, Morton, xy-column-major) the fraction of (tile, offset) and (tile, chunk of 7 / 10 offsets)
combinations with at least one present pair — the work a tile-skipping kernel still has to do.  CPU only (numpy).
Result (profiles/r04_stem_coherence.txt): coherent tiles see MORE offsets, not fewer (a compact blob of rows reaches in every
direction): 0.81 -> 0.88 at 64 rows, 0.67 -> 0.68 at 16; chunk skipping saves 8-18 % at best.  The sparsity of the stem map
has no tile structure to exploit: csrc/stem.hip works on pairs instead."""
import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from dpcr_agb_amd import synthetic
b = synthetic.make_sparse_batch([0], n_points=16000)
c = b.coords.numpy().astype(np.int64)  # [N,3] x,y,z
c -= c.min(0)
N=len(c); print('voxels',N, 'extent', c.max(0)+1)
X,Y,Z = c.max(0)+1+6
grid = -np.ones((Z,Y,X),dtype=np.int64)
grid[c[:,2]+3,c[:,1]+3,c[:,0]+3]=np.arange(N)
offs=[(dx,dy,dz) for dz in range(-3,4) for dy in range(-3,4) for dx in range(-3,4)]
nbr=np.stack([grid[c[:,2]+3+dz,c[:,1]+3+dy,c[:,0]+3+dx] for dx,dy,dz in offs])  # [343,N]
pres = nbr>=0
print('pairs',pres.sum(), 'density',pres.mean())
def morton(c):
    def part(v):
        v=v.astype(np.uint64); r=np.zeros_like(v)
        for i in range(8): r |= ((v>>np.uint64(i))&np.uint64(1))<<np.uint64(3*i)
        return r
    return part(c[:,0])|(part(c[:,1])<<np.uint64(1))|(part(c[:,2])<<np.uint64(2))
orders={'input':np.arange(N),'zmajor':np.lexsort((c[:,0],c[:,1],c[:,2])),'morton':np.argsort(morton(c),kind='stable'),
 'xy-col-major(z fastest)':np.lexsort((c[:,2],c[:,0],c[:,1]))}
for name,o in orders.items():
    p=pres[:,o]
    for T in (16,32,64):
        nt=N//T
        pt=p[:,:nt*T].reshape(343,nt,T).any(2)   # [343, tiles] offset present in tile
        # chunk of 10 offsets
        for OPC in (1,7,10):
            nch=(343+OPC-1)//OPC
            pad=np.zeros((nch*OPC,nt),bool); pad[:343]=pt
            ch=pad.reshape(nch,OPC,nt).any(1)
            print(f'{name:24s} T={T:3d} OPC={OPC:2d}: nonempty chunk frac {ch.mean():.3f}; offsets-present-in-tile {pt.mean():.3f}')
