"""Evaluation measures on the GPU (drop-in for the
reduction-type measures of nsol/similarity_measures.py:26-120): SSD, SAD, MAE,
MSE, RMSE, PSNR, NCC.  x, x_ref: NumPy arrays or torch HIP tensors of equal
shape.  One fused pass (nsol_pair_stats_*) yields every sum; NCC takes a
second, mean-centred pass for accuracy.  SSIM, MI, NMI and Dice of the
reference are evaluation-only extras and are not provided.
"""
import numpy as np

from . import ops
from .device import is_device_tensor, to_device


def _pair(x, x_ref):
    shape_x = tuple(x.shape)
    if shape_x != tuple(x_ref.shape):
        raise ValueError("Input data shapes do not match")
    if is_device_tensor(x) and is_device_tensor(x_ref):
        a = x.contiguous().view(-1)
        return a, x_ref.to(a.dtype).contiguous().view(-1)
    dx = x if is_device_tensor(x) else to_device(
        np.asarray(x, dtype=np.float64).reshape(-1), np.float64)
    dr = x_ref if is_device_tensor(x_ref) else to_device(
        np.asarray(x_ref, dtype=np.float64).reshape(-1), np.float64)
    dx = dx.contiguous().view(-1)
    return dx, dr.to(dx.dtype).contiguous().view(-1)


class SimilarityMeasures(object):

    @staticmethod
    def sum_of_absolute_differences(x, x_ref):
        a, b = _pair(x, x_ref)
        return float(ops.pair_stats(a, b)[3])

    @staticmethod
    def mean_absolute_error(x, x_ref):
        a, b = _pair(x, x_ref)
        return float(ops.pair_stats(a, b)[3]) / float(a.numel())

    @staticmethod
    def sum_of_squared_differences(x, x_ref):
        a, b = _pair(x, x_ref)
        return float(ops.pair_stats(a, b)[4])

    @staticmethod
    def mean_squared_error(x, x_ref):
        a, b = _pair(x, x_ref)
        return float(ops.pair_stats(a, b)[4]) / float(a.numel())

    @staticmethod
    def root_mean_square_error(x, x_ref):
        return float(np.sqrt(SimilarityMeasures.mean_squared_error(x, x_ref)))

    @staticmethod
    def peak_signal_to_noise_ratio(x, x_ref):
        a, b = _pair(x, x_ref)
        st = ops.pair_stats(a, b)
        mse = st[4] / float(a.numel())
        return float(10 * np.log10(st[5] ** 2 / mse))

    @staticmethod
    def normalized_cross_correlation(x, x_ref):
        a, b = _pair(x, x_ref)
        n = float(a.numel())
        st = ops.pair_stats(a, b)
        st = ops.pair_stats(a, b, st[6] / n, st[7] / n)   # centred pass
        sx = np.sqrt(st[1] / (n - 1.0))
        sy = np.sqrt(st[2] / (n - 1.0))
        return float(st[0] / (n * sx * sy))

    SSD = sum_of_squared_differences
    SAD = sum_of_absolute_differences
    MAE = mean_absolute_error
    MSE = mean_squared_error
    RMSE = root_mean_square_error
    PSNR = peak_signal_to_noise_ratio
    NCC = normalized_cross_correlation


SimilarityMeasures.similarity_measures = {
    "SSD": SimilarityMeasures.sum_of_squared_differences,
    "SAD": SimilarityMeasures.sum_of_absolute_differences,
    "MAE": SimilarityMeasures.mean_absolute_error,
    "MSE": SimilarityMeasures.mean_squared_error,
    "RMSE": SimilarityMeasures.root_mean_square_error,
    "PSNR": SimilarityMeasures.peak_signal_to_noise_ratio,
    "NCC": SimilarityMeasures.normalized_cross_correlation,
}
