"""Configuration objects of the matching path, field-for-field as in the reference
(`karios/core/configuration.py:36-50, 93-104`; defaults from
`karios/configuration/processing_configuration.json:6-21`).  Any object with the same
attributes (the reference's own dataclass, a Mock, a SimpleNamespace) is accepted
everywhere a `conf` is expected.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Literal, Union


@dataclass
class KLTConfiguration:
    # pylint: disable=invalid-name, too-many-instance-attributes
    """KLT config object (names kept compatible with existing KARIOS config files)."""
    minDistance: int = 10
    blocksize: int = 15
    maxCorners: int = 20000
    matching_winsize: int = 25
    qualityLevel: float = 0.1
    xStart: int = 0
    tile_size: int = 20000
    laplacian_kernel_size: Union[int, Dict[str, int], Literal["auto"]] = 7
    outliers_filtering: bool = False
    laplacian_invert_polarity: Union[bool, Literal["auto"]] = False


@dataclass
class AccuracyAnalysisConfiguration:
    """Accuracy analysis module configuration (only the field the scoring step reads)."""
    confidence_threshold: float = 0.4


@dataclass
class ShiftConfiguration:
    """Large shift image preprocessing configuration."""
    bias_correction_min_threshold: int = 2
