"""FieldHeadNames used by the path.  Inside a nerfstudio installation the reference's own enum is re-exported, so that
the output dicts of the drop-in fields are indexed by the very objects `NeuRadarModel` looks them up with
(field_components/field_heads.py:25-38; models/neuradar.py:1011); stand-alone, an enum with the same members."""
from enum import Enum

try:  # drop-in use: the reference's enum
    from nerfstudio.field_components.field_heads import FieldHeadNames  # type: ignore  # noqa: F401
except Exception:  # noqa: BLE001  (stand-alone: nerfstudio is not importable)

    class FieldHeadNames(Enum):
        DENSITY = "density"
        SDF = "sdf"
        ALPHA = "alpha"
        FEATURE = "feature"
