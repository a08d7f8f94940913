"""Host-side stencil tap definitions (drop-in for nsol/kernels.py).

The taps only parameterise the HIP kernels (inverse spacings for the finite
differences, 1-D factors or dense taps for the blur); they are tiny arrays
built once in float64 with the reference's formulas:
  finite differences   kernels.py:102-112, 160-190, 240-286
  Gaussian             kernels.py:80-100, 120-158, 198-238
Arrays are indexed [(z,) y, x]; spacing[0] belongs to x, the LAST array axis.
"""
import numpy as np


class Kernels(object):

    def __init__(self, dimension, spacing):
        spacing = np.atleast_1d(spacing).astype(float)
        if spacing.size != dimension:
            raise ValueError("dimension of spacing and space must be the same")
        self._dimension = dimension
        self._spacing = spacing

    def get_dimension(self):
        return self._dimension

    def get_spacing(self):
        return self._spacing

    # one generic builder serves dx/dy/dz in every dimension: direction a
    # (0 = x, 1 = y, 2 = z) lives on array axis dimension-1-a.
    def _difference(self, direction, backward):
        if direction >= self._dimension:
            raise AttributeError(
                "a %dD kernel set has no derivative along direction %d" %
                (self._dimension, direction))
        taps = np.array([0., 1., -1.]) if backward else np.array([1., -1.])
        if self._dimension == 1:
            return taps / self._spacing
        shape = [1] * self._dimension
        shape[self._dimension - 1 - direction] = taps.size
        return taps.reshape(shape) / self._spacing[direction]

    def get_dx_forward_difference(self):
        return self._difference(0, False)

    def get_dx_backward_difference(self):
        return self._difference(0, True)

    def get_gaussian(self, cov, alpha_cut=3):
        d = self._dimension
        if d == 1:
            var = np.asarray(cov, dtype=float)
            half = np.ceil(np.sqrt(var) * alpha_cut / self._spacing)
            half = float(np.atleast_1d(half)[0])
            t = np.arange(-half, half + 1, 1)
            w = np.exp(-0.5 * (t * (self._spacing ** 2 / var) * t))
            return w / np.sum(w)
        cov = np.asarray(cov)
        if cov.shape != (d, d):
            raise ValueError("Numpy array 'cov' must be of shape (%d,%d)" %
                             (d, d))
        half = np.ceil(np.sqrt(cov.diagonal()) * alpha_cut / self._spacing)
        axes = [np.arange(-h, h + 1, 1) for h in half]      # x, y(, z) ranges
        mesh = np.meshgrid(*axes, indexing='ij')
        # coordinates are stacked (z,) y, x while the array is laid out
        # (x, y(, z)): this reproduces the reference's axis convention.
        pts = np.array([m.flatten() for m in mesh[::-1]])
        S = np.diag(self._spacing)
        prec = S.dot(np.linalg.inv(cov)).dot(S)
        w = np.exp(-0.5 * np.sum(pts * prec.dot(pts), 0))
        w = w / np.sum(w)
        return w.reshape([a.size for a in axes])


class Kernels1D(Kernels):

    def __init__(self, spacing=1):
        Kernels.__init__(self, dimension=1, spacing=spacing)


class Kernels2D(Kernels):

    def __init__(self, spacing=np.ones(2)):
        Kernels.__init__(self, dimension=2, spacing=spacing)

    def get_dy_forward_difference(self):
        return self._difference(1, False)

    def get_dy_backward_difference(self):
        return self._difference(1, True)


class Kernels3D(Kernels2D):

    def __init__(self, spacing=np.ones(3)):
        Kernels.__init__(self, dimension=3, spacing=spacing)

    def get_dz_forward_difference(self):
        return self._difference(2, False)

    def get_dz_backward_difference(self):
        return self._difference(2, True)
