"""Host camera with the API of sutil::Camera (SDK/sutil/Camera.h:38-76, incl. the fork's focal scale
`setFocalLength`, Camera.h:64,75).  The frame itself is computed by libmipt's host code."""
import numpy as np

from . import api


class Camera:
    def __init__(self, eye=(1.0, 1.0, 1.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fovY=35.0, aspectRatio=1.0):
        self._eye = np.asarray(eye, np.float32)
        self._lookat = np.asarray(lookat, np.float32)
        self._up = np.asarray(up, np.float32)
        self._fovY = float(fovY)
        self._aspect = float(aspectRatio)
        self._fod = 1.0

    def eye(self): return self._eye
    def setEye(self, v): self._eye = np.asarray(v, np.float32)
    def lookat(self): return self._lookat
    def setLookat(self, v): self._lookat = np.asarray(v, np.float32)
    def up(self): return self._up
    def setUp(self, v): self._up = np.asarray(v, np.float32)
    def fovY(self): return self._fovY
    def setFovY(self, v): self._fovY = float(v)
    def aspectRatio(self): return self._aspect
    def setAspectRatio(self, v): self._aspect = float(np.float32(v))
    def setFocalLength(self, v): self._fod = float(v)

    def UVWFrame(self):
        """UVW forms an orthogonal, but not orthonormal basis (Camera.h:66)."""
        return api.camera_uvw(self._eye, self._lookat, self._up, self._fovY, self._aspect, self._fod)
