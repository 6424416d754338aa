"""NumPy stand-in for the reference's ``Image`` as far as ``Integration`` needs it
(mpsfm/sfm/scene/image/base.py:26-80): an image of a NumpyReconstruction with prior depth / normal
maps, exposing ``mpsfm_rec, imid, image, camera, depth, normals`` and the Integration mixin."""

from __future__ import annotations

import numpy as np

from mpsfm_amd.sfm.scene.integration import Integration


class NumpyNormals:
    def __init__(self, data, uncertainty, data_downscaled=None, uncertainty_downscaled=None):
        self.data = np.asarray(data, dtype=np.float64)                # [H,W,3]
        self.uncertainty = np.asarray(uncertainty, dtype=np.float64)  # [H,W,3,3]
        # the reference's Normals object carries half-resolution maps from the normal estimator
        # (scene/image/normals.py:196-233); tests pass them in
        self.data_downscaled = None if data_downscaled is None else np.asarray(data_downscaled, dtype=np.float64)
        self.uncertainty_downscaled = None if uncertainty_downscaled is None else np.asarray(uncertainty_downscaled, dtype=np.float64)


class NumpyIntegrableImage(Integration):
    def __init__(self, mpsfm_rec, imid, normals: NumpyNormals):
        Integration.__init__(self)
        self.mpsfm_rec, self.imid = mpsfm_rec, imid
        self.image = mpsfm_rec.images[imid]
        self.camera = mpsfm_rec.rec.cameras[self.image.camera_id]
        self.depth = self.image.depth
        self.normals = normals
