import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from test_gpu_triangulator import *
sc, cg, prob, truth = empty_scene(9, 1200, 75)
tri = MpsfmTriangulator({"colmap_options": dict(OPTS), "lift_low_parallax": False, "new_retry_nbatch": None}, sc, cg)
register_all(sc, tri)
rng = np.random.default_rng(0)
long_pts = [pid for pid, p in sc.points3D.items() if p.track.length() >= 4]
errs=[]; removed=[]
for pid in rng.choice(long_pts, 150, replace=False):
    p=sc.points3D[int(pid)]; e = p.track.elements[-1]
    im=sc.images[e.image_id]; Xc=im.cam_from_world*p.xyz[None]; K=sc.rec.cameras[im.camera_id].params
    uv=np.array([K[0]*Xc[0,0]/Xc[0,2]+K[2], K[1]*Xc[0,1]/Xc[0,2]+K[3]])
    errs.append(np.linalg.norm(uv-im.kps[e.point2D_idx])); removed.append((int(pid),e.image_id,e.point2D_idx))
    sc.obs.delete_observation(e.image_id, e.point2D_idx)
errs=np.array(errs); print("reproj err of removed: median %.2f, <4px: %d of 150"%(np.median(errs),(errs<4).sum()))
n_c = tri.complete_all_tracks()
back=sum(1 for (pid,i,k) in removed if sc.images[i].kp_point3D[k]==pid)
other=sum(1 for (pid,i,k) in removed if sc.images[i].kp_point3D[k]!=pid and sc.images[i].kp_point3D[k]!=INVALID_POINT3D)
print("completed", n_c, "back to own point", back, "to another point", other)
ops=tri._triangulator.last_ops; print("ops", len(ops["type"]), np.bincount(ops["type"]))
