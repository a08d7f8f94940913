"""Synthetic benchmark volumes (build-owned inputs, not
part of the reference): checkerboard blocks of period n/4 (amplitude 100) plus
a centred ball of radius n/3 (height 50); 'gauss' adds 5 % Gaussian noise, 'sp'
sets 5 % of the voxels to 0 and 5 % to 150."""
import numpy as np


def synth_volume(n, seed=0, kind="gauss", dtype=np.float64):
    q = max(n // 4, 1)
    i = np.arange(n)
    blk = i // q
    v = 100.0 * ((blk[:, None, None] + blk[None, :, None] +
                  blk[None, None, :]) % 2).astype(np.float64)
    r2 = (i - n / 2.0) ** 2
    v += 50.0 * ((r2[:, None, None] + r2[None, :, None] + r2[None, None, :])
                 < (n / 3.0) ** 2)
    rng = np.random.default_rng(seed)
    if kind == "gauss":
        v = v + 0.05 * v.max() * rng.standard_normal(v.shape)
    elif kind == "sp":
        u = rng.random(v.shape)
        v = np.where(u < 0.05, 0.0, np.where(u > 0.95, 150.0, v))
    elif kind != "clean":
        raise ValueError(kind)
    return v.astype(dtype)
