"""Host-side mirrors of the `karios.core` names the matching path touches."""
from .configuration import KLTConfiguration, ShiftConfiguration, AccuracyAnalysisConfiguration  # noqa: F401
from .image import DeviceRasterImage, DeviceWindow, NumpyRasterImage, shift_image  # noqa: F401
