"""Minimal NIfTI-1 single-file (.nii / .nii.gz) reader and writer, enough for
the volumes NSoL's command-line tools exchange (the reference goes through
SimpleITK, nsol/data_reader.py:62-66).  Arrays are returned indexed [z, y, x];
spacing is (hx, hy, hz) = pixdim[1:4]."""
import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32,
           64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32}
_CODES = {np.dtype(v).name: k for k, v in _DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") \
        else open(path, mode)


def read(path):
    """-> (array [z,y,x] (or [y,x]), spacing tuple, raw 348-byte header)."""
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 352:
        raise IOError("'%s' is not a NIfTI-1 file" % path)
    endian = "<"
    if struct.unpack("<i", raw[0:4])[0] != 348:
        endian = ">"
        if struct.unpack(">i", raw[0:4])[0] != 348:
            raise IOError("'%s' is not a NIfTI-1 file" % path)
    dim = struct.unpack(endian + "8h", raw[40:56])
    datatype = struct.unpack(endian + "h", raw[70:72])[0]
    pixdim = struct.unpack(endian + "8f", raw[76:108])
    vox_offset = int(struct.unpack(endian + "f", raw[108:112])[0])
    slope, inter = struct.unpack(endian + "2f", raw[112:120])
    if datatype not in _DTYPES:
        raise IOError("NIfTI datatype %d is not supported" % datatype)
    nd = dim[0]
    shape = [d for d in dim[1:1 + nd]]
    while len(shape) > 1 and shape[-1] == 1:
        shape.pop()
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(endian)
    count = int(np.prod(shape))
    data = np.frombuffer(raw, dtype=dt, count=count, offset=vox_offset)
    arr = data.reshape(shape[::-1]).astype(np.float64)
    if slope not in (0.0, 1.0) or inter != 0.0:
        if slope != 0.0:
            arr = arr * slope + inter
    spacing = tuple(float(p) if p > 0 else 1.0
                    for p in pixdim[1:1 + len(shape)])
    return arr, spacing, raw[:348]


def write(path, array, spacing=None, header=None):
    """Write `array` ([z,y,x]) as float32/float64 (other dtypes as given)."""
    arr = np.ascontiguousarray(array)
    if arr.dtype.name not in _CODES:
        arr = arr.astype(np.float32)
    shape = arr.shape[::-1]
    hdr = bytearray(header) if header is not None and len(header) == 348 \
        else bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [len(shape)] + list(shape) + [1] * (7 - len(shape))
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<h", hdr, 70, _CODES[arr.dtype.name])
    struct.pack_into("<h", hdr, 72, arr.dtype.itemsize * 8)
    if spacing is not None or header is None:
        sp = list(spacing) if spacing is not None else [1.0] * len(shape)
        pix = [1.0] + [float(v) for v in sp] + [1.0] * (7 - len(sp))
        struct.pack_into("<8f", hdr, 76, *pix)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<2f", hdr, 112, 1.0, 0.0)
    hdr[344:348] = b"n+1\x00"
    with _open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(b"\x00\x00\x00\x00")
        f.write(arr.astype(arr.dtype.newbyteorder("<")).tobytes())
