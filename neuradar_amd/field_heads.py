"""FieldHeadNames used by the path (mirrors field_components/field_heads.py of the reference)."""
from enum import Enum


class FieldHeadNames(Enum):
    DENSITY = "density"
    SDF = "sdf"
    ALPHA = "alpha"
    FEATURE = "feature"
