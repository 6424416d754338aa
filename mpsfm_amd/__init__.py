"""mpsfm_amd — MI355X-native bundle adjustment / triangulation numerics for MP-SfM.

Host-side mirror of the reference's ``mpsfm.sfm.mapper`` Optimizer / MpsfmTriangulator API over
the C ABI of ``libmpsfm_hip.so`` (include/mpsfm_hip.h).
"""

__version__ = "0.1.0"
