"""Result writer (drop-in for nsol/data_writer.py:21-71 without SimpleITK)."""
import os

import numpy as np

from . import nifti


class DataWriter(object):

    def __init__(self, nda, path_to_file, image_sitk=None):
        self._nda = nda
        self._path_to_file = path_to_file
        self._image_info = image_sitk
        self._file_type = os.path.basename(path_to_file).split(".")[1]

    def write_data(self):
        d = os.path.dirname(self._path_to_file)
        if d:
            os.makedirs(d, exist_ok=True)
        writer = {"txt": self._write_txt, "png": self._write_png,
                  "mat": self._write_mat, "nii": self._write_nii,
                  "npy": self._write_npy}
        if self._file_type not in writer:
            raise IOError("File type '%s' is not supported" % self._file_type)
        writer[self._file_type]()

    def _write_png(self):
        from PIL import Image
        nda = np.round(np.array(self._nda)).astype(np.uint8)
        Image.fromarray(nda).save(self._path_to_file)

    def _write_txt(self):
        np.savetxt(self._path_to_file, np.atleast_2d(self._nda))

    def _write_mat(self):
        import scipy.io
        scipy.io.savemat(self._path_to_file, {"nda": self._nda})

    def _write_nii(self):
        spacing = header = None
        if self._image_info is not None:
            spacing = self._image_info.GetSpacing()
            header = self._image_info.header
        nifti.write(self._path_to_file, np.asarray(self._nda), spacing, header)

    def _write_npy(self):
        np.save(self._path_to_file, np.asarray(self._nda))
