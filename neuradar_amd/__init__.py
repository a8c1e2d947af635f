"""neuradar_amd -- MI355X-native (gfx950) volumetric-rendering hot path of NeuRadar.

Host-side mirror of the reference's operator surface for this path (same class names, constructor
arguments and output conventions as nerfstudio's `HashEncoding`, `MLP`, `SHEncoding`,
`NeuRADHashEncoding`, `NeuRADField`, `NeuRADProposalField`, `PowerSampler`, `PDFSampler`,
`ProposalNetworkSampler`, renderers and sensor ray generators) over hand-written HIP kernels behind a
C ABI (`include/neuradar_hip.h`, `neuradar_amd/csrc/`).  `implementation="hip"` is the only
implementation: there is no CPU or eager fallback -- a missing extension is a hard error.
"""
import os as _os

# MIOpen 3.5.0 (ROCm 7.2, the one PyTorch 2.10 bundles): the NHWC implicit-GEMM backward-DATA assembly kernels for gfx950
# (`igemm_bwd_gtcx35_nhwc_{fp16,fp32}_*`, solver ConvAsmImplicitGemmGTCDynamicBwdXdlopsNHWC) access memory behind the end of a
# tensor -- found with the guard allocator of tests/test_gpu_redzone.py on the RGB decoder's own first layer, Conv2d(48, 32, 1) on a
# 2 x 8 x 8 channels-last batch: 128 pixels under a 256-row tile.  MIOpen's benchmark search runs every
# applicable solver, so any channels-last convolution backward that goes through the library (the fp32 decoder, torch.autocast
# reference legs) executes them; the overrun is silent until the tensor is the last block of an allocator segment, then it is a
# GPU page fault = SIGABRT from the ROCr runtime (round 4: two aborted suite runs).  The solver is switched off for every process
# that imports this package (MIOpen reads the variable when it first enumerates solvers; an explicit setting by the user wins).
# The product path of the 16-bit decoder never enters MIOpen (nr_conv7_* / nr_pw_* / nr_bn_act_*).
_os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC", "0")

from . import _lib  # noqa: E402,F401

__all__ = ["_lib", "ops", "encodings", "mlp", "neurad_encoding", "neurad_field", "rays", "ray_samplers",
           "renderers", "sensors", "step", "parallel"]
