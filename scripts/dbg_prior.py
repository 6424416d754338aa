import numpy as np, sys
sys.path.insert(0,'.')
from mpsfm_amd import capi
from oracle import prior_oracle
z=np.load('tests/golden/reference_priorutils.npz')
tag='a'; kps=z[f'pu_{tag}_kps'][11:12]; s=z[f'pu_{tag}_s']; valid=z[f'pu_{tag}_valid']
kw=dict(depth_maps=[valid.astype(float)], valid_maps=[valid], sx=[s[0]], sy=[s[1]], cam_quat=[[0,0,0,1.0]], cam_t=[[0,0,5.0]], obs_img=np.zeros(1,np.int32), obs_xy=kps, obs_var=np.full(1,0.01), obs_pt=np.zeros(1,np.int32), pts=np.zeros((1,3)))
h=capi.depth_blocks(**kw); r=prior_oracle.depth_blocks(**kw)
print("hip depth", repr(h['depth'][0]), "flags", h['flags'], "ref", repr(r['depth'][0]), r['flags'])
# x-only ramp map to read back the x coordinate weights: map[y][x] = x
H,W=valid.shape
ramp=np.tile(np.arange(W,dtype=float),(H,1))
kw['depth_maps']=[ramp]; print("x sample hip", repr(capi.depth_blocks(**kw)['depth'][0]), "ref", repr(prior_oracle.depth_blocks(**kw)['depth'][0]))
