"""neuradar_amd -- MI355X-native (gfx950) volumetric-rendering hot path of NeuRadar.

Host-side mirror of the reference's operator surface for this path (same class names, constructor
arguments and output conventions as nerfstudio's `HashEncoding`, `MLP`, `SHEncoding`,
`NeuRADHashEncoding`, `NeuRADField`, `NeuRADProposalField`, `PowerSampler`, `PDFSampler`,
`ProposalNetworkSampler`, renderers and sensor ray generators) over hand-written HIP kernels behind a
C ABI (`include/neuradar_hip.h`, `neuradar_amd/csrc/`).  `implementation="hip"` is the only
implementation: there is no CPU or eager fallback -- a missing extension is a hard error.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "ops", "encodings", "mlp", "neurad_encoding", "neurad_field", "rays", "ray_samplers",
           "renderers", "sensors", "step", "parallel"]
