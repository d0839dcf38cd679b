"""MI355X drop-in for the `karios.matcher` package surface."""
from .klt import KLT, klt_tracker  # noqa: F401
from .large_offset import LargeOffsetMatcher, detect_large_offset  # noqa: F401
from .mutual_info_service import MutualInfoService  # noqa: F401
from .zncc_service import ZNCCService  # noqa: F401
