"""Image / volume reader (drop-in for nsol/data_reader.py:21-66 without
SimpleITK): png (PIL), mat (scipy.io), nii / nii.gz (nsol_amd.nifti), npy."""
import os

import numpy as np

from . import nifti


class ImageInfo(object):
    """Stands in for the SimpleITK image the reference hands around: carries the
    voxel spacing and the header needed to write a result next to its input."""

    def __init__(self, spacing, header=None):
        self._spacing = tuple(spacing)
        self.header = header

    def GetSpacing(self):
        return self._spacing


class DataReader(object):

    def __init__(self, path_to_file):
        self._path_to_file = path_to_file
        self._file_type = os.path.basename(path_to_file).split(".")[1]
        self._nda = None
        self._image_info = None

    def read_data(self):
        if not os.path.isfile(self._path_to_file):
            raise IOError("Filename '%s' not found" % (self._path_to_file))
        reader = {"png": self._read_png, "mat": self._read_mat,
                  "nii": self._read_nii, "npy": self._read_npy}
        if self._file_type not in reader:
            raise IOError("File type '%s' is not supported" % self._file_type)
        reader[self._file_type]()

    def get_data(self):
        return np.array(self._nda, dtype=np.float64)

    def get_image_sitk(self):
        return self._image_info

    def _read_png(self):
        from PIL import Image
        img = Image.open(self._path_to_file)
        if img.mode not in ("L", "I", "F", "I;16"):
            img = img.convert("L")
        self._nda = np.array(img)

    def _read_mat(self):
        import scipy.io
        dic = scipy.io.loadmat(self._path_to_file)
        ndas = [dic[k] for k in dic.keys() if isinstance(dic[k], np.ndarray)]
        if len(ndas) > 1:
            raise IOError("MAT file '%s' must include one array only" %
                          (self._path_to_file))
        self._nda = ndas[0]

    def _read_nii(self):
        arr, spacing, header = nifti.read(self._path_to_file)
        self._nda = arr
        self._image_info = ImageInfo(spacing, header)

    def _read_npy(self):
        self._nda = np.load(self._path_to_file)
