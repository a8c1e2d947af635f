"""neuradar_amd -- MI355X-native (gfx950) volumetric-rendering hot path of NeuRadar.

Host-side mirror of the reference's operator surface for this path (same class names, constructor
arguments and output conventions as nerfstudio's `HashEncoding`, `MLP`, `SHEncoding`,
`NeuRADHashEncoding`, `NeuRADField`, `NeuRADProposalField`, `PowerSampler`, `PDFSampler`,
`ProposalNetworkSampler`, renderers and sensor ray generators) over hand-written HIP kernels behind a
C ABI (`include/neuradar_hip.h`, `neuradar_amd/csrc/`).  `implementation="hip"` is the only
implementation: there is no CPU or eager fallback -- a missing extension is a hard error.
"""
import os as _os

_MIOPEN_KNOB = "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC"


def apply_miopen_workaround(force: bool = False) -> bool:
    """Switch off MIOpen's NHWC implicit-GEMM backward-DATA assembly solver for this process -- EXPLICITLY (until round 5 the
    package did it for every importing process as an import side effect).

    MIOpen 3.5.0 (ROCm 7.2, the one PyTorch 2.10 bundles): the `igemm_bwd_gtcx35_nhwc_{fp16,fp32}_*` kernels (solver
    ConvAsmImplicitGemmGTCDynamicBwdXdlopsNHWC) access memory behind the end of a tensor -- found with the guard allocator of
    tests/test_gpu_redzone.py on the RGB decoder's own first layer, Conv2d(48, 32, 1) on a 2 x 8 x 8 channels-last batch
    (DESIGN.md section 12).  MIOpen's benchmark search runs every applicable solver, so any channels-last convolution backward
    that goes through the library executes them; the overrun is silent until the tensor ends an allocator segment, then it is
    a GPU page fault = SIGABRT from the ROCr runtime.

    Who needs it: processes that run channels-last convolution BACKWARDS through MIOpen -- the fp32 decoder
    (`mlp_dtype="float32"`: DecoderLossHead calls this itself), `torch.autocast` reference legs of the tests, bench.py's fp32
    runs.  The 16-bit decoder of the product path never enters MIOpen (nr_conv7_* / nr_pw_* / nr_bn_act_*).  Call it BEFORE the
    process's first convolution (MIOpen reads the variable when it first enumerates solvers).  Version-gated: applied for the
    MIOpen builds known to be affected (3.5.x) unless force=True; a value the user has set is never overwritten.  Returns True
    when the variable is (now) "0".  Logged once."""
    if _os.environ.get(_MIOPEN_KNOB) is not None:
        return _os.environ[_MIOPEN_KNOB] == "0"
    if not force:
        try:
            import torch

            ver = int(torch.backends.cudnn.version() or 0)  # MIOpen: major * 1 000 000 + minor * 1 000 + patch
        except Exception:  # noqa: BLE001
            ver = 0
        if ver and not (3005000 <= ver < 3006000):
            return False
    _os.environ[_MIOPEN_KNOB] = "0"
    import logging

    logging.getLogger("neuradar_amd").warning("%s=0 set for this process (MIOpen 3.5.0 NHWC backward-data overrun: "
                                              "neuradar_amd.apply_miopen_workaround)", _MIOPEN_KNOB)
    return True


from . import _lib  # noqa: E402,F401

__all__ = ["apply_miopen_workaround", "_lib", "ops", "encodings", "mlp", "neurad_encoding", "neurad_field", "rays", "ray_samplers",
           "renderers", "sensors", "step", "parallel"]
