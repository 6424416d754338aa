"""integration_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy/SciPy restatement of the CPU branch of the reference's depth-from-normals integration
(bilateral normal integration with depth priors), the place where normal priors enter MP-SfM:

  reference mpsfm/sfm/scene/image/integration.py
      _integrate                :383-520   IRLS loop, preconditioned CG, energy tests
      calc_energy               :139-165
      calc_Amat                 :167-234   normal equations as explicit CSR
      process_depth_prior       :236-263,  process_normals_prior :265-283,
      process_sparse_depth      :285-293,  load_depth_checkpoint :295-306
      init_Nz / init_int_vars   :308-356,  update_W :358-364,  calc_Wpm :366-377
      generate_dx_dy            :631-680   forward/backward difference operators
  reference mpsfm/sfm/scene/camera.py:36-99       CameraIntData masks and index tables
  reference mpsfm/utils/integration.py:32-56      move_left/right/top/bottom, sigmoid
  defaults: reference mpsfm/sfm/scene/image/base.py:30-55

The reference's module itself cannot be imported here (it needs cv2 and cholespy at import time),
but its CPU branch is plain NumPy/SciPy (utils/integration.py:17-24) and two of its pieces ARE
importable: CameraIntData and utils.integration.  tests/golden/make_golden_integration.py imports those
from /root/reference to pin the masks, index tables and the sigmoid of this restatement; the solver
itself is scipy.sparse.linalg.cg, as in the reference.

Only tests/ and bench legs may import this module.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.linalg import cg

DEFAULT_CONF = dict(
    large_number=1e6, max_iter=10, tol=5e-2, step_size=1, cg_max_iter=5000, cg_tol=1e-3, lambda1=1, lambda2=1, k=1,
    depth_magnitude_multiplier=1, normals_magnitude_multiplier=1, scale_filter=True, scale_filter_factor=1.5,
)


# ---- utils/integration.py:32-56 -------------------------------------------------------------
def move_left(mask):
    return np.pad(mask, ((0, 0), (0, 1)), "constant", constant_values=0)[:, 1:]


def move_right(mask):
    return np.pad(mask, ((0, 0), (1, 0)), "constant", constant_values=0)[:, :-1]


def move_top(mask):
    return np.pad(mask, ((0, 1), (0, 0)), "constant", constant_values=0)[1:, :]


def move_bottom(mask):
    return np.pad(mask, ((1, 0), (0, 0)), "constant", constant_values=0)[:-1, :]


def sigmoid(x, k=1):
    cc = np.clip(-k * x, -709, 709)
    return 1 / (1 + np.exp(cc))


# ---- camera.py:36-99 --------------------------------------------------------------------------
class CamData:
    def __init__(self, H, W):
        self.nshape = (H, W)
        self.num_normals = H * W
        m = np.ones((H, W), bool)
        self.has_left_mask = np.logical_and(move_right(m), m)
        self.has_left_mask_left = move_left(self.has_left_mask)
        self.has_right_mask = np.logical_and(move_left(m), m)
        self.has_right_mask_right = move_right(self.has_right_mask)
        self.has_bottom_mask = np.logical_and(move_top(m), m)
        self.has_bottom_mask_bottom = move_bottom(self.has_bottom_mask)
        self.has_top_mask = np.logical_and(move_bottom(m), m)
        self.has_top_mask_top = move_top(self.has_top_mask)
        for nm in ("left", "right", "bottom", "top"):
            setattr(self, f"has_{nm}_mask_flat", getattr(self, f"has_{nm}_mask")[m])
            setattr(self, f"has_{nm}_mask_{nm}_flat", getattr(self, f"has_{nm}_mask_{nm}")[m])
        self.pixel_idx = np.arange(self.num_normals).reshape(H, W)
        self.pixel_idx_flat = np.arange(self.num_normals)
        self.pixel_idx_flat_indptr = np.arange(self.num_normals + 1)
        self.pixel_idx_left_center = self.pixel_idx[self.has_left_mask]
        self.pixel_idx_right_right = self.pixel_idx[self.has_right_mask_right]
        self.pixel_idx_top_center = self.pixel_idx[self.has_top_mask]
        self.pixel_idx_bottom_bottom = self.pixel_idx[self.has_bottom_mask_bottom]
        self.pixel_idx_left_left_indptr = np.concatenate([np.array([0]), np.cumsum(self.has_left_mask_left_flat)])
        self.pixel_idx_right_center_indptr = np.concatenate([np.array([0]), np.cumsum(self.has_right_mask_flat)])
        self.pixel_idx_top_top_indptr = np.concatenate([np.array([0]), np.cumsum(self.has_top_mask_top_flat)])
        self.pixel_idx_bottom_center_indptr = np.concatenate([np.array([0]), np.cumsum(self.has_bottom_mask_flat)])


# ---- integration.py:631-680 ---------------------------------------------------------------------
def generate_dx_dy(cam: CamData, nz_horizontal, nz_vertical, step_size=1):
    n = cam.num_normals
    pix = cam.pixel_idx

    def op(values, idx_a, idx_b, row_mask):
        data = np.stack([-values / step_size, values / step_size], -1).flatten()
        indices = np.stack((idx_a, idx_b), -1).flatten()
        indptr = np.concatenate([np.array([0]), np.cumsum(row_mask.flatten().astype(int) * 2)])
        return csr_matrix((data, indices, indptr), shape=(n, n))

    nz_left = nz_horizontal[cam.has_left_mask.flatten()]
    nz_right = nz_horizontal[cam.has_right_mask.flatten()]
    nz_top = nz_vertical[cam.has_top_mask.flatten()]
    nz_bottom = nz_vertical[cam.has_bottom_mask.flatten()]
    D_h_neg = op(nz_left, pix[move_left(cam.has_left_mask)], pix[cam.has_left_mask], cam.has_left_mask)
    D_h_pos = op(nz_right, pix[cam.has_right_mask], pix[move_right(cam.has_right_mask)], cam.has_right_mask)
    D_v_pos = op(nz_top, pix[cam.has_top_mask], pix[move_top(cam.has_top_mask)], cam.has_top_mask)
    D_v_neg = op(nz_bottom, pix[move_bottom(cam.has_bottom_mask)], pix[cam.has_bottom_mask], cam.has_bottom_mask)
    return D_h_pos, D_h_neg, D_v_pos, D_v_neg


@dataclass
class IntState:
    """What the reference caches on the image between calls (IntVars, integration.py:18-29)."""

    integrated: bool = False
    energy_old: float | None = None
    wu: np.ndarray | None = None
    wv: np.ndarray | None = None
    A: tuple | None = None  # (A1, A2, A3, A4)


@dataclass
class IntInputs:
    depth_prior: np.ndarray          # [H,W]
    depth_uncertainty: np.ndarray    # [H,W] variance of the prior depth
    valid: np.ndarray                # [H,W] bool
    normals: np.ndarray              # [H,W,3]
    normals_uncertainty: np.ndarray  # [H,W,3,3]
    depth_init: np.ndarray           # [H,W] current (integrated) depth map = checkpoint
    K: tuple                         # (K[1,1]*sy, K[0,0]*sx, K[1,2]*sy, K[0,2]*sx)  (integration.py:118-124)
    kps: np.ndarray                  # [n,2] integer pixel (x, y) of the sparse 3-D points
    depth3d: np.ndarray              # [n]
    zvars3d: np.ndarray              # [n]
    conf: dict = field(default_factory=lambda: dict(DEFAULT_CONF))


def prepare(inp: IntInputs):
    """process_depth_prior / process_normals_prior / load_depth_checkpoint / process_sparse_depth
    (+ the sparse scale filter of _integrate :392-398)."""
    c = inp.conf
    H, W = inp.depth_prior.shape
    depth_prior = np.asarray(inp.depth_prior, dtype=np.float64)
    depth_precision = c["depth_magnitude_multiplier"] * np.asarray(1 / (inp.depth_uncertainty + 1e-6), dtype=np.float64)
    depth_precision = (depth_precision * depth_prior**2).flatten()
    z_prior = np.log(depth_prior).flatten()
    normal_map = np.asarray(inp.normals, dtype=np.float64)
    nx, ny, nz = normal_map[..., 1].flatten(), normal_map[..., 0].flatten(), -normal_map[..., 2].flatten()
    nu = np.array(inp.normals_uncertainty, dtype=np.float64)
    nu[~np.asarray(inp.valid, bool)] = c["large_number"]
    m = 1 / c["normals_magnitude_multiplier"]
    Vnx, Vny, Vnz = m * nu[..., 1, 1].flatten(), m * nu[..., 0, 0].flatten(), m * nu[..., 2, 2].flatten()
    z = np.log(np.asarray(inp.depth_init, dtype=np.float64).flatten())
    kps = np.asarray(inp.kps, dtype=np.int64).reshape(-1, 2)
    if len(kps):
        sparse_ids = np.ravel_multi_index((kps[:, 1], kps[:, 0]), (H, W))
        depth3d = np.asarray(inp.depth3d, dtype=np.float64)
        sparse_precision = (np.asarray(1 / inp.zvars3d) * depth3d**2).flatten()
        sparse_depth = np.log(depth3d).flatten()
    else:
        sparse_ids, sparse_precision, sparse_depth = np.zeros(0, np.int64), np.zeros(0), np.zeros(0)
    if c["scale_filter"] and len(sparse_ids):
        div = np.exp(sparse_depth) / np.exp(z_prior[sparse_ids])
        valid = (div < c["scale_filter_factor"]) * (div > (1 / c["scale_filter_factor"]))
        sparse_ids, sparse_precision, sparse_depth = sparse_ids[valid], sparse_precision[valid], sparse_depth[valid]
    return dict(depth_precision=depth_precision, z_prior=z_prior, nx=nx, ny=ny, nz=nz, Vnx=Vnx, Vny=Vny, Vnz=Vnz, z=z,
                sparse_ids=sparse_ids, sparse_precision=sparse_precision, sparse_depth=sparse_depth, shape=(H, W))


def init_int_vars(cam: CamData, z, K, nx, ny, nz, Vnx, Vny, Vnz, conf, state: IntState, init=True):
    fx, fy, cx, cy = K
    yy, xx = np.meshgrid(np.arange(cam.nshape[1]), np.arange(cam.nshape[0]))
    xx = np.flip(xx, axis=0)
    uu = xx.flatten() - cx
    vv = yy.flatten() - cy
    nz_u = uu * nx + vv * ny + fx * nz
    nz_v = uu * nx + vv * ny + fy * nz
    if init and state.integrated:
        A1, A2, A3, A4 = state.A
    else:
        A3, A4, A1, A2 = generate_dx_dy(cam, nz_horizontal=nz_v, nz_vertical=nz_u, step_size=conf["step_size"])
    state.A = (A1, A2, A3, A4)
    Nz = dict(left_square=nz_v[cam.has_left_mask_flat] ** 2, right_square=nz_v[cam.has_right_mask_flat] ** 2,
              top_square=nz_u[cam.has_top_mask_flat] ** 2, bottom_square=nz_u[cam.has_bottom_mask_flat] ** 2)
    if not (init and state.integrated):
        state.wu, state.wv = update_W(z, state.A, conf["k"])
    Duz = -nx / nz_u
    Dvz = -ny / nz_v
    one = np.ones_like(z)
    Nu_precision = 1 / (Vnx * ((uu * Duz + one) ** 2) + Vny * (vv * Duz) ** 2 + fx**2 * Vnz * Duz**2)
    Nv_precision = 1 / (Vnx * (uu * Dvz) ** 2 + Vny * (vv * Dvz + one) ** 2 + fy**2 * Vnz * Dvz**2)
    return Nz, Nu_precision, Nv_precision, nz_u, nz_v


def update_W(z, A, k):
    A1, A2, A3, A4 = A
    wu = sigmoid((A2.dot(z)) ** 2 - (A1.dot(z)) ** 2, k)
    wv = sigmoid((A4.dot(z)) ** 2 - (A3.dot(z)) ** 2, k)
    return wu, wv


def calc_Wpm(Nu_precision, Nv_precision, wu, wv):
    return wu * Nu_precision, (1 - wu) * Nu_precision, wv * Nv_precision, (1 - wv) * Nv_precision


def calc_energy(A, W4, z, nx, ny, depth_precision, z_prior, sparse_precision, sparse_depth, sparse_ids, conf):
    A1, A2, A3, A4 = A
    wu_plus, wu_minus, wv_plus, wv_minus = W4
    e = np.sum((wu_plus * (A1.dot(z) + nx) ** 2) + (wu_minus * (A2.dot(z) + nx) ** 2) + (wv_plus * (A3.dot(z) + ny) ** 2)
               + (wv_minus * (A4.dot(z) + ny) ** 2))
    e += np.sum(conf["lambda1"] * depth_precision * (z_prior - z) ** 2)
    if len(sparse_ids) > 0:
        e += np.sum(conf["lambda2"] * sparse_precision * (sparse_depth - z[sparse_ids]) ** 2)
    return e


def calc_Amat(cam: CamData, Nz, W4, depth_precision, sparse_precision, sparse_ids, conf, sparse_depth=True):
    wu_plus, wu_minus, wv_plus, wv_minus = W4
    t_top = wu_plus[cam.has_top_mask_flat] * Nz["top_square"]
    t_bottom = wu_minus[cam.has_bottom_mask_flat] * Nz["bottom_square"]
    t_left = wv_minus[cam.has_left_mask_flat] * Nz["left_square"]
    t_right = wv_plus[cam.has_right_mask_flat] * Nz["right_square"]
    d = np.zeros(cam.num_normals)
    d[cam.has_left_mask_flat] += t_left
    d[cam.has_left_mask_left_flat] += t_left
    d[cam.has_right_mask_flat] += t_right
    d[cam.has_right_mask_right_flat] += t_right
    d[cam.has_top_mask_flat] += t_top
    d[cam.has_top_mask_top_flat] += t_top
    d[cam.has_bottom_mask_flat] += t_bottom
    d[cam.has_bottom_mask_bottom_flat] += t_bottom
    d += conf["lambda1"] * depth_precision
    if sparse_depth and len(sparse_ids) > 0:
        d[sparse_ids] += conf["lambda2"] * sparse_precision  # NumPy semantics: duplicates do not accumulate
    n = cam.num_normals
    A_d = csr_matrix((d, cam.pixel_idx_flat, cam.pixel_idx_flat_indptr), shape=(n, n))
    A_l = csr_matrix((-t_left, cam.pixel_idx_left_center, cam.pixel_idx_left_left_indptr), shape=(n, n))
    A_r = csr_matrix((-t_right, cam.pixel_idx_right_right, cam.pixel_idx_right_center_indptr), shape=(n, n))
    A_t = csr_matrix((-t_top, cam.pixel_idx_top_center, cam.pixel_idx_top_top_indptr), shape=(n, n))
    A_b = csr_matrix((-t_bottom, cam.pixel_idx_bottom_bottom, cam.pixel_idx_bottom_center_indptr), shape=(n, n))
    odu = A_t + A_b + A_r + A_l
    return A_d + odu + odu.T, d


def rhs(A, W4, nx, ny, depth_precision, z_prior, sparse_precision, sparse_depth, sparse_ids, conf):
    A1, A2, A3, A4 = A
    wu_plus, wu_minus, wv_plus, wv_minus = W4
    b = A1.T @ (wu_plus * (-nx)) + A2.T @ (wu_minus * (-nx)) + A3.T @ (wv_plus * (-ny)) + A4.T @ (wv_minus * (-ny))
    b += conf["lambda1"] * depth_precision * z_prior
    if len(sparse_ids) > 0:
        b[sparse_ids] += conf["lambda2"] * sparse_precision * sparse_depth  # duplicates do not accumulate
    return b


def integrate(inp: IntInputs, state: IntState | None = None, init=True, record=None):
    """_integrate (:383-520).  Returns (depth map [H,W] or None when nothing changed, changed, state, info)."""
    state = state if state is not None else IntState()
    c = inp.conf
    P = prepare(inp)
    H, W = P["shape"]
    cam = CamData(H, W)
    z = P["z"]
    sp_ids, sp_prec, sp_depth = P["sparse_ids"], P["sparse_precision"], P["sparse_depth"]
    Nz, Nu_p, Nv_p, nz_u, nz_v = init_int_vars(cam, z, inp.K, P["nx"], P["ny"], P["nz"], P["Vnx"], P["Vny"], P["Vnz"], c, state, init=init)
    W4 = calc_Wpm(Nu_p, Nv_p, state.wu, state.wv)
    en = lambda zz, ww: calc_energy(state.A, ww, zz, P["nx"], P["ny"], P["depth_precision"], P["z_prior"], sp_prec, sp_depth, sp_ids, c)
    energy = en(z, W4)
    info = dict(energies=[float(energy)], cg_iters=[], nz_u=nz_u, nz_v=nz_v, Nu_precision=Nu_p, Nv_precision=Nv_p)
    if state.integrated and not (abs(energy - state.energy_old) / state.energy_old > c["tol"]):
        return None, False, state, info
    energy_0 = min_energy = energy
    for _ in range(c["max_iter"]):
        A_mat, diag = calc_Amat(cam, Nz, W4, P["depth_precision"], sp_prec, sp_ids, c)
        b = rhs(state.A, W4, P["nx"], P["ny"], P["depth_precision"], P["z_prior"], sp_prec, sp_depth, sp_ids, c)
        n = cam.num_normals
        D = csr_matrix((1 / np.clip(diag, 1e-5, None), cam.pixel_idx_flat, cam.pixel_idx_flat_indptr), shape=(n, n))
        its = [0]

        def cb(_x):
            its[0] += 1

        if record is not None:
            record.append(dict(A_diag=diag.copy(), b=b.copy(), z_in=z.copy(), W4=[w.copy() for w in W4]))
        z, _ = cg(A_mat, b, x0=z, M=D, maxiter=c["cg_max_iter"], rtol=c["cg_tol"], callback=cb)
        info["cg_iters"].append(its[0])
        state.wu, state.wv = update_W(z, state.A, c["k"])
        W4 = calc_Wpm(Nu_p, Nv_p, state.wu, state.wv)
        energy_old = energy
        min_energy = min(energy, min_energy)
        energy = en(z, W4)
        info["energies"].append(float(energy))
        rel = abs(energy - energy_old) / energy_old
        rel_min = abs(energy - min_energy) / min_energy
        if ((rel < c["tol"] and (energy_old - energy) > 0) or (rel_min < c["tol"] and (min_energy - energy) > 0)) and energy < energy_0:
            break
        if energy > energy_0:
            state.integrated = True
            state.energy_old = energy_0
            return None, False, state, info
    state.integrated = True
    state.energy_old = energy
    return np.exp(z.reshape(H, W)).astype(np.float64), True, state, info


# ---- row f4: uncertainty propagation (reference integration.py:51-79, 522-616) ----------------------------

def calculate_hessian(inp: IntInputs, ignore_depths=True):
    """calculate_hessian (:522-574) for the map in `inp` (already downscaled by the caller when the
    reference would): calc_Amat at the checkpoint `depth_init`, weights recomputed from it (init=False),
    no scale filter, sparse term only when not ignore_depths."""
    conf = dict(inp.conf)
    conf["scale_filter"] = False  # :542 calls process_sparse_depth only
    P = prepare(IntInputs(**{**inp.__dict__, "conf": conf}))
    H, W = P["shape"]
    cam = CamData(H, W)
    state = IntState()
    Nz, Nu_p, Nv_p, _, _ = init_int_vars(cam, P["z"], inp.K, P["nx"], P["ny"], P["nz"], P["Vnx"], P["Vny"], P["Vnz"], conf, state,
                                         init=False)
    W4 = calc_Wpm(Nu_p, Nv_p, state.wu, state.wv)
    A_mat, _ = calc_Amat(cam, Nz, W4, P["depth_precision"], P["sparse_precision"], P["sparse_ids"], conf,
                         sparse_depth=not ignore_depths)
    return A_mat


def uncertainty_solve(hessian, xy, imshape, chunk_size=128):
    """IntegrationUncertainty.solve (:62-78), literally: one unit right-hand side per query pixel, the
    result is the SUM of the solution column (`x.sum(0)`, :77).  The reference factors with cholespy in
    float32; this restatement uses SciPy's sparse LU in float64."""
    from scipy.sparse import csc_matrix
    from scipy.sparse.linalg import splu

    lu = splu(csc_matrix(hessian))
    n = hessian.shape[0]
    xy = np.round(np.asarray(xy, dtype=np.float64)).astype(int).reshape(-1, 2)
    out = []
    for i in range(0, len(xy), chunk_size):
        q = xy[i:i + chunk_size]
        indices = np.ravel_multi_index(q.T[::-1], imshape[:2])
        tgt = np.zeros((n, len(indices)))
        tgt[(indices, np.arange(len(indices)))] = 1
        out.append(lu.solve(tgt).sum(0))
    return np.concatenate(out) if out else np.zeros(0)


def resize_linear(img, dsize):
    """cv2.resize(img, (w, h)) with the default INTER_LINEAR for a single-channel float64 image, restated
    from OpenCV's documented sampling rule (pixel centres: src = (dst + 0.5) * scale - 0.5, border
    replicated, interpolation coefficients held in float32).  cv2 is not importable here: unpinned."""
    img = np.asarray(img, dtype=np.float64)
    h, w = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])

    def taps(n_src, n_dst):
        scale = 1.0 / (n_dst / n_src)
        f = (np.arange(n_dst) + 0.5) * scale - 0.5
        i0 = np.floor(f).astype(int)
        a = (f - i0).astype(np.float32)
        lo = i0 < 0
        i0[lo], a[lo] = 0, 0
        hi = i0 >= n_src - 1
        i0[hi], a[hi] = n_src - 1, 0
        return i0, np.minimum(i0 + 1, n_src - 1), a.astype(np.float64)

    x0, x1, ax = taps(w, dw)
    y0, y1, ay = taps(h, dh)
    rows = img[:, x0] * (1 - ax) + img[:, x1] * ax
    return rows[y0] * (1 - ay)[:, None] + rows[y1] * ay[:, None]
