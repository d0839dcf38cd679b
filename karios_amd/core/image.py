"""numpy-backed stand-in for `karios.core.image.GdalRasterImage` + `shift_image`.

GDAL I/O is out of scope (KARIOS keeps it); the matcher only touches the accessor
surface listed in SURVEY.md 8(a18): `.x_size .y_size .read() .array .no_data_value
.clear_cache() .file_name .filepath`.  A real `GdalRasterImage` can be passed instead.
"""
from __future__ import annotations

import numpy as np

from .. import ops


class NumpyRasterImage:
    """Duck type of GdalRasterImage (reference core/image.py:255-459) over an in-memory array."""

    def __init__(self, array: np.ndarray, no_data_value=None, filepath: str = "memory.tif"):
        a = np.asarray(array)
        if a.ndim != 2:
            raise ValueError("NumpyRasterImage expects a single-band 2-D array")
        self._array = a
        self.no_data_value = no_data_value
        self.filepath = filepath

    @property
    def file_name(self) -> str:
        return self.filepath.rsplit("/", 1)[-1]

    @property
    def x_size(self) -> int:
        return int(self._array.shape[1])

    @property
    def y_size(self) -> int:
        return int(self._array.shape[0])

    @property
    def array(self) -> np.ndarray:
        return self._array

    def read(self, band: int, x_off: int, y_off: int, x_size: int, y_size: int) -> np.ndarray:
        """Same argument order as GdalRasterImage.read (image.py:351-370)."""
        if band != 1:
            raise ValueError("single-band image")
        return self._array[y_off:y_off + y_size, x_off:x_off + x_size]

    _keeps_array_across_clear_cache = True

    def clear_cache(self) -> None:
        """GdalRasterImage drops its cached array here (image.py:445-447); the in-memory image keeps its data - but the call is the
        owner's signal that the pixels may change: resident copies shared between the matcher services are invalidated and the
        array becomes writeable again (`karios_amd.resident.shared_pair`)."""
        from ..resident import invalidate_raster
        invalidate_raster(self)


class DeviceRasterImage:
    """The same accessor surface over a raster that already lives in HBM (a 2-D torch tensor on the GPU): `.read()` hands out
    device views, `ResidentPair.from_window` / `karios_amd.parallel.ResidentUnit.load` then copy device to device and nothing
    crosses PCIe.  `dtype` names the pixel type when the tensor only stores its bit pattern (torch has no uint16: the
    synthetic generator returns int16 storage)."""

    def __init__(self, tensor, dtype=None, no_data_value=None, filepath: str = "device.tif"):
        if tensor.dim() != 2 or not tensor.is_cuda:
            raise ValueError("DeviceRasterImage expects a 2-D tensor on the GPU")
        self.tensor = tensor
        self.dtype = np.dtype(dtype) if dtype is not None else np.dtype(str(tensor.dtype).replace("torch.", ""))
        self.no_data_value = no_data_value
        self.filepath = filepath

    file_name = NumpyRasterImage.file_name

    @property
    def x_size(self) -> int:
        return int(self.tensor.shape[1])

    @property
    def y_size(self) -> int:
        return int(self.tensor.shape[0])

    def read(self, band: int, x_off: int, y_off: int, x_size: int, y_size: int):
        if band != 1:
            raise ValueError("single-band image")
        return DeviceWindow(self.tensor[y_off:y_off + y_size, x_off:x_off + x_size], self.dtype)

    def clear_cache(self) -> None:
        pass


class DeviceWindow:
    """A box of a `DeviceRasterImage`: a (possibly strided) device tensor view + the pixel type it stands for."""

    def __init__(self, view, dtype):
        self.view, self.dtype = view, np.dtype(dtype)
        self.shape = tuple(view.shape)


def shift_image(img, y_off=0, x_off=0):
    """karios.core.image.shift_image (image.py:70-101), executed on the GPU."""
    return ops.shift_image(img, y_off=y_off, x_off=x_off)
