"""Red-zone harness for the kernels of the RGB decoder's 16-bit chain ("find and fix the abort"): every kernel that
moves 16-byte vectors -- nr_conv7_pack / nr_conv7_fwd (halo `request`) / nr_conv7_wgrad, nr_pw_fwd / nr_pw_bwd_data /
nr_pw_bwd_weight (load8 / store8, the MFMA transposed convolution), nr_bn_act_fwd / nr_bn_act_bwd, nr_unscale_add_16 -- is called
through the C ABI on RAW device buffers laid out so that an out-of-bounds access cannot hide:

  * every buffer is the LAST bytes of its own hipMalloc'ed region, and the region right behind it in the address space has been
    hipFree'd where the runtime handed out adjacent regions (reported): a read or write past the end is a GPU page fault, i.e. the
    ROCr runtime aborts the process -- the cases run in CHILD processes (helpers.run_child), so that is a named test failure;
  * 256 canary bytes sit in FRONT of every buffer and are compared afterwards (writes before the start);
  * shapes are the odd ones: P * H * W not a multiple of 8 or 32, images smaller than a tile, one pixel.

torch's caching allocator never gives this: its blocks are pieces of 2-MB / 20-MB segments, so an overrun lands in mapped memory
and faults only when the block happens to be the last of a segment -- the state-dependent crash the round-4 suite hit twice in
eight runs.  `test_the_guard_itself_catches_an_overrun` shows that the layout does fault (a launch told that its buffer is longer
than it is must kill the child)."""
import ctypes
import os

import pytest
import torch

from helpers import run_child

pytestmark = pytest.mark.gpu
DEV = "cuda"
REGION = 2 << 20  # one hipMalloc per buffer, a multiple of the runtime's 2-MB granule
CANARY = 0xA5


class Guarded:
    """Raw device buffers flush against the end of their own hipMalloc'ed regions, with a freed region behind them where possible."""

    def __init__(self) -> None:
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self.hip.hipFree.argtypes = [ctypes.c_void_p]
        self.hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        self.live, self.bufs = [], {}
        torch.zeros(1, device=DEV)  # (the runtime is initialised by torch)

    def _malloc(self, size: int) -> int:
        p = ctypes.c_void_p()
        assert self.hip.hipMalloc(ctypes.byref(p), size) == 0, "hipMalloc"
        return int(p.value)

    def _region(self, size: int) -> int:
        """A hipMalloc'ed region of `size` bytes (a multiple of 2 MB).  The runtime spaces such regions 2 MB apart and maps nothing in
        between (tools/probe_redzone.py: six 2-MB regions came back 4 MB apart, 14-MB ones 16 MB apart; an overrun of 8 bytes behind a
        region's end faults) -- test_the_guard_itself_catches_an_overrun keeps that assumption honest."""
        base = self._malloc(size)
        self.live.append(base)
        return base

    def put(self, name: str, t: torch.Tensor) -> int:
        """Device copy of `t` (contiguous, in the memory order the kernel wants) ending at the end of a fresh region (exactly, for sizes
        that are multiples of 16 bytes); returns the device address."""
        flat = t.detach().contiguous().view(-1).view(torch.uint8)
        n = flat.numel()
        size = (n + 512 + REGION - 1) // REGION * REGION
        base = self._region(size)
        addr = (base + size - n) & ~15  # (the entry points want 16-byte aligned starts: up to 15 bytes of slack behind odd sizes)
        assert self.hip.hipMemset(ctypes.c_void_p(base), CANARY, size - n) == 0
        if n:
            host = flat.cpu().numpy()
            assert self.hip.hipMemcpy(ctypes.c_void_p(addr), ctypes.c_void_p(host.ctypes.data), n, 1) == 0  # host to device
        self.bufs[name] = (addr, n, base, size, t.dtype, tuple(t.shape))
        return addr

    def get(self, name: str) -> torch.Tensor:
        addr, n, _, _, dtype, shape = self.bufs[name]
        host = torch.empty(n, dtype=torch.uint8)
        if n:
            assert self.hip.hipMemcpy(ctypes.c_void_p(host.numpy().ctypes.data), ctypes.c_void_p(addr), n, 2) == 0  # device to host
        return host.view(dtype)

    def check(self) -> None:
        """Synchronise (hipMemcpy is synchronous: a fault would have killed the process by now) and compare the canaries."""
        torch.cuda.synchronize()
        for name, (addr, n, base, size, _, _) in self.bufs.items():
            k = min(256, size - n)
            host = torch.empty(k, dtype=torch.uint8)
            assert self.hip.hipMemcpy(ctypes.c_void_p(host.numpy().ctypes.data), ctypes.c_void_p(addr - k), k, 2) == 0
            assert bool((host == CANARY).all()), f"{name}: bytes in FRONT of the buffer were overwritten"


def _p(a: int):
    return ctypes.c_void_p(a)


def _case(kernel: str, dtype: str, P: int, H: int, W: int, lie: int = 0):
    """One kernel on guarded buffers (child process).  lie > 0 (the guard's self-test): the launch is told that the buffer holds
    `lie` more elements than it does."""
    from neuradar_amd import _lib

    lib = _lib.lib()
    _lib.ensure_device(0)
    dt = getattr(torch, dtype)
    code = _lib.NR_DTYPES[dtype]
    g = Guarded()
    gen = torch.Generator().manual_seed(P * 1000 + H * 10 + W)
    rnd = lambda *s: torch.randn(*s, generator=gen)  # noqa: E731
    M = P * H * W
    st = None  # the NULL stream
    rc = 0
    if kernel in ("conv7_fwd", "conv7_wgrad", "conv7_pack"):
        w = (rnd(32, 7, 7, 32) / 40.0).to(dt)  # [O, kh, kw, I]: the parameter's channels-last memory
        b = rnd(32).to(dt)
        flat = torch.cat([w.reshape(-1), b])
        lst = _lib.NrConv7List()
        lst.n, lst.offset[0], lst.bias_offset[0] = 1, 0, w.numel()
        img_bytes = lib.nr_conv7_image_bytes()
        a_flat, a_img = g.put("weights", flat), g.put("images", torch.zeros(img_bytes, dtype=torch.uint8))
        rc = lib.nr_conv7_pack(_p(a_flat), ctypes.byref(lst), code, _p(a_img), st)
        if kernel != "conv7_pack":
            x = rnd(P, H, W, 32).to(dt)
            a_x = g.put("x", x)
            if kernel == "conv7_fwd":
                a_res, a_y = g.put("residual", rnd(P, H, W, 32).to(dt)), g.put("y", torch.zeros(P, H, W, 32, dtype=dt))
                for orient in (0, 1):  # forward image, then the data gradient's
                    rc = rc or lib.nr_conv7_fwd(_p(a_x), _p(a_img + orient * (img_bytes // 2)), _p(a_res) if orient == 0 else None,
                                                1 - orient, _p(a_y), P, H, W, code, st)
            else:
                a_gy = g.put("grad_y", rnd(P, H, W, 32).to(dt))
                a_gw, a_gb = g.put("grad_w", torch.zeros(32 * 49 * 32, dtype=dt)), g.put("grad_b", torch.zeros(32, dtype=dt))
                a_ws = g.put("workspace", torch.zeros(lib.nr_conv7_wgrad_workspace_bytes(), dtype=torch.uint8))
                for acc in (0, 1):
                    rc = rc or lib.nr_conv7_wgrad(_p(a_x), _p(a_gy), _p(a_gw), _p(a_gb), acc, _p(a_ws), P, H, W, code, st)
    elif kernel in ("pw_head", "pw_tail", "pw_convt", "pw_convt_generic"):
        if kernel == "pw_convt_generic":
            _lib.set_tuning("NR_PW_MFMA_OFF", 1)
        K, O, act, tr, x_f32, y_f32 = {"pw_head": (48, 32, 1, 0, 1, 0), "pw_tail": (32, 3, 2, 0, 0, 1), "pw_convt": (32, 32, 0, 1, 0, 0),
                                       "pw_convt_generic": (32, 32, 0, 1, 0, 0)}[kernel]
        N = 9 * O if tr else O
        out_px = 9 * M if tr else M
        x = rnd(M, K) if x_f32 else rnd(M, K).to(dt)
        wt = (rnd(K, 9, O) / 6.0).to(dt) if tr else (rnd(O, K) / 6.0).to(dt)
        a_x, a_w, a_b = g.put("x", x), g.put("w", wt), g.put("b", rnd(O).to(dt))
        ydt = torch.float32 if y_f32 else dt
        a_y, a_gy = g.put("y", torch.zeros(out_px, O, dtype=ydt)), g.put("grad_y", rnd(out_px, O).to(ydt))
        a_gx = g.put("grad_x", torch.zeros(M, K, dtype=torch.float32 if x_f32 else dt))
        a_gw, a_gb = g.put("grad_w", torch.zeros(N * K, dtype=dt)), g.put("grad_b", torch.zeros(O, dtype=dt))
        a_ws = g.put("workspace", torch.zeros(lib.nr_pw_workspace_bytes(), dtype=torch.uint8))
        a_sc = g.put("scale", torch.full((1,), 0.5))
        rc = lib.nr_pw_fwd(_p(a_x), x_f32, _p(a_w), _p(a_b), _p(a_y), y_f32, M, K, O, act, tr, H, W, code, st)
        rc = rc or lib.nr_pw_bwd_data(_p(a_gy), _p(a_y), y_f32, _p(a_w), _p(a_gx), x_f32, M, K, O, act, tr, H, W, _p(a_sc) if x_f32 else None,
                                      code, st)
        for acc in (0, 1):
            rc = rc or lib.nr_pw_bwd_weight(_p(a_x), x_f32, _p(a_gy), _p(a_y), y_f32, _p(a_gw), _p(a_gb), acc, _p(a_ws), M, K, O, act, tr, H, W,
                                            code, st)
    elif kernel == "bn":
        C = 32
        xdt = dt
        a_x, a_res = g.put("x", rnd(M, C).to(xdt)), g.put("residual", rnd(M, C).to(xdt))
        a_y, a_g = g.put("y", torch.zeros(M, C, dtype=xdt)), g.put("grad_y", rnd(M, C).to(xdt))
        a_dx, a_dr = g.put("grad_x", torch.zeros(M, C, dtype=xdt)), g.put("grad_res", torch.zeros(M, C, dtype=xdt))
        a_ga, a_be = g.put("gamma", 1.0 + 0.1 * rnd(C)), g.put("beta", 0.1 * rnd(C))
        a_rm, a_rv = g.put("running_mean", torch.zeros(C)), g.put("running_var", torch.ones(C))
        a_sm, a_sr = g.put("save_mean", torch.zeros(C)), g.put("save_rstd", torch.zeros(C))
        a_gg, a_gb = g.put("grad_gamma", torch.zeros(C)), g.put("grad_beta", torch.zeros(C))
        a_ws = g.put("workspace", torch.zeros(lib.nr_bn_act_workspace_floats(M, C)))
        rc = lib.nr_bn_act_fwd(_p(a_x), _p(a_res), M, C, code, _p(a_ga), _p(a_be), 1e-5, 0.1, _p(a_rm), _p(a_rv), 1, _p(a_y), _p(a_sm), _p(a_sr),
                               _p(a_ws), st)
        rc = rc or lib.nr_bn_act_bwd(_p(a_g), _p(a_y), _p(a_x), M, C, code, _p(a_ga), _p(a_sm), _p(a_sr), 1, _p(a_dx), _p(a_dr), _p(a_gg), _p(a_gb),
                                     _p(a_ws), st)
    elif kernel == "unscale_add_16":
        n = M * 32 + 3  # (not a multiple of 4: the scalar tail)
        a_dst, a_src = g.put("dst", rnd(n)), g.put("src", rnd(n).to(dt))
        a_inv, a_flag = g.put("inv_scale", torch.full((1,), 0.25)), g.put("flag", torch.zeros(1))
        rc = lib.nr_unscale_add_16(_p(a_dst), _p(a_src), n + lie, code, _p(a_inv), _p(a_flag), st)
    else:
        raise KeyError(kernel)
    assert rc == 0, f"{kernel}: rc {rc}"
    g.check()
    finite = all(bool(torch.isfinite(g.get(k).float()).all()) for k in g.bufs if g.bufs[k][4].is_floating_point and k not in ("workspace",))
    return {"buffers": len(g.bufs), "finite": finite}


SHAPES = [(3, 5, 3), (1, 17, 45), (1, 1, 1), (2, 8, 8), (2, 24, 24), (1, 9, 33)]  # (2, 8, 8) -> (2, 24, 24): the golden batch's own


def _cases(kernel: str):
    """Every dtype and shape of one kernel in ONE child (a fault kills it: the case in flight is the last line it printed)."""
    import sys

    n = 0
    for dtype in ("float16", "bfloat16"):
        for P, H, W in (SHAPES[:1] if kernel == "conv7_pack" else SHAPES):
            print(f"[case] {kernel} {dtype} P={P} H={H} W={W}", file=sys.stderr, flush=True)
            r = _case(kernel, dtype, P, H, W)
            assert r["finite"], f"{kernel} {dtype} {P}x{H}x{W}: non-finite values in an output"
            n += r["buffers"]
    return {"buffers": n}


@pytest.mark.parametrize("kernel", ["conv7_pack", "conv7_fwd", "conv7_wgrad", "pw_head", "pw_tail", "pw_convt", "pw_convt_generic", "bn",
                                    "unscale_add_16"])
def test_cnn_kernels_stay_inside_their_buffers(kernel):
    r = run_child(__file__, "_cases", timeout=600, kernel=kernel)
    print(f"{kernel}: {r['buffers']} guarded buffers over 2 operand types x {1 if kernel == 'conv7_pack' else len(SHAPES)} shapes: no fault, canaries intact")


def _environment_assumption(ok: bool, what: str) -> None:
    """Statements about THIS runtime / library build (unmapped space behind a region; MIOpen's defect still present), not about this
    repo's code: where one no longer holds the test is reported as xfailed with the reason -- under the driver's `-x` a hard failure
    here would cut the tests behind it off for something no change in this repo can fix."""
    if not ok:
        pytest.xfail(what)


def _dies(func: str, env=None, **kwargs):
    """Run tests/child_main.py <this file> func in a child that is EXPECTED to die; returns (returncode, stderr tail)."""
    import json
    import subprocess
    import sys
    import tempfile

    here = os.path.dirname(os.path.abspath(__file__))
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [sys.executable, os.path.join(here, "child_main.py"), os.path.abspath(__file__), func, json.dumps(kwargs), os.path.join(tmp, "r.pt")]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(here), env=dict(os.environ, **(env or {})))
    return r.returncode, r.stderr[-20000:]


def test_the_guard_itself_catches_an_overrun():
    """nr_unscale_add_16 told that its buffers hold FOUR more elements than they do (8 bytes of fp16 behind the end): the child must
    die of the GPU fault -- SIGABRT from the ROCr runtime with "Memory access fault by GPU node" on its stderr, the signature of
    the round-4 aborts.  This is what keeps the harness honest: if the runtime ever maps memory behind a region, it fails."""
    rc, err = _dies("_case", kernel="unscale_add_16", dtype="float16", P=1, H=9, W=33, lie=4)
    print(f"overrun child: rc {rc}; stderr tail: {err[-300:]}")
    _environment_assumption(rc != 0 and "Memory access fault" in err,
                            f"an 8-byte overrun behind a hipMalloc'ed region did not fault here (rc {rc}): the red-zone tests cannot see overruns on this runtime")


# ---- the whole camera chain under a guard ALLOCATOR ----------------------------------------------------------------------------
GUARD_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "guard_alloc", "libguard_alloc.so")


def _ensure_guard_so() -> None:
    """The allocator library is built by __graft_entry__.build() and travels with the tree; if it is missing anyway, build it here
    (hipcc is on the GPU box too), and if that is impossible skip rather than fail: a missing test tool is not a product defect."""
    if os.path.exists(GUARD_SO):
        return
    import subprocess

    src = os.path.join(os.path.dirname(GUARD_SO), "guard_alloc.cpp")
    try:
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-fPIC", "-shared", "-std=c++17", "-w", src, "-o", GUARD_SO],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"{GUARD_SO} is missing and could not be built here ({type(e).__name__})")


def _install_guard_allocator():
    """Every torch allocation of this process = the last bytes of its own hipMalloc'ed region (tests/guard_alloc/guard_alloc.cpp):
    an overrun by ANY kernel -- this repo's, MIOpen's, ATen's -- is a GPU fault.  Must run before the first device allocation."""
    assert os.path.exists(GUARD_SO), f"{GUARD_SO} is missing: __graft_entry__.build() compiles it"
    alloc = torch.cuda.memory.CUDAPluggableAllocator(GUARD_SO, "guard_malloc", "guard_free")
    torch.cuda.memory.change_current_allocator(alloc)
    return ctypes.CDLL(GUARD_SO)


def _guarded_cnn_leg(dtype: str, mode: str, overrun: int = 0):
    """One leg of test_gpu_full_step.py::test_cnn_16_bit_working_copies_equal_autocast (the test the round-4 suite aborted in) with
    the guard allocator installed.  overrun > 0: the allocator's self-test -- a kernel is made to run behind two torch tensors."""
    lib = _install_guard_allocator()
    if overrun:  # a launch told that two torch tensors hold `overrun` more elements than they do
        from neuradar_amd import _lib

        dst, src = torch.zeros(4096, device=DEV), torch.ones(4096, device=DEV, dtype=torch.float16)
        rc = _lib.lib().nr_unscale_add_16(_p(dst.data_ptr()), _p(src.data_ptr()), 4096 + overrun, _lib.NR_DTYPES["float16"], None, None, None)
        torch.cuda.synchronize()
        return {"rc": rc}
    import test_gpu_full_step as t

    out = t._cnn_leg(dtype, mode)
    torch.cuda.synchronize()
    n_alloc = int(lib.guard_allocations())
    del t
    import gc

    gc.collect()
    return {"allocations": n_alloc, "canary_failures": int(lib.guard_canary_failures()), "loss": out["loss"]}


def test_the_guard_allocator_catches_an_overrun_of_a_torch_kernel():
    _ensure_guard_so()
    rc, err = _dies("_guarded_cnn_leg", dtype="float16", mode="copies", overrun=8)
    print(f"overrun child: rc {rc}; stderr tail: {err[-300:]}")
    _environment_assumption(rc != 0 and "Memory access fault" in err,
                            f"a 16-byte overrun behind a guard-allocator tensor did not fault here (rc {rc}): the guarded tests cannot see overruns on this runtime")


@pytest.mark.parametrize("mode", ["copies", "autocast", "fp32"])
@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_cnn_legs_under_the_guard_allocator(dtype, mode):
    """The three legs of the test in which the round-4 suite aborted twice (SIGABRT inside the camera chain's backward), each in a
    child whose every tensor ends at the end of its own mapped region: whichever kernel reads or writes behind a tensor -- the
    working copies' kernels of this repo (`copies`), MIOpen's / ATen's under torch.autocast (`autocast`) or in fp32 -- faults here
    deterministically instead of once in four suite runs.  AMD_SERIALIZE_KERNEL=3: the fault arrives while the launching thread
    still sits in the launch, so the child's stack dump (stderr tail in the failure message) names the operator."""
    _ensure_guard_so()
    r = run_child(__file__, "_guarded_cnn_leg", timeout=900, env={"AMD_SERIALIZE_KERNEL": "3"}, dtype=dtype, mode=mode)
    print(f"{dtype} {mode}: {r['allocations']} guarded allocations, loss {r['loss']:.6f}")
    assert r["canary_failures"] == 0, "bytes in front of an allocation were overwritten (see the child's stderr)"


def test_the_round_4_abort_is_miopens_nhwc_backward_data_kernel():
    """ROOT CAUSE of the round-4 aborts, pinned: with MIOpen's solver ConvAsmImplicitGemmGTCDynamicBwdXdlopsNHWC enabled (the
    library's default; neuradar_amd/__init__.py switches it off) the fp32 leg dies under the guard allocator with the ROCr fault
    message, and the last kernel the runtime launched is one of MIOpen's `igemm_bwd_gtcx35_nhwc_*` -- tried by its benchmark
    search for the backward of the decoder's Conv2d(48, 32, 1) on 2 x 8 x 8 pixels.  With the solver off (the test above) the same
    leg is clean.  If a later MIOpen fixes the kernel this test reports xfailed -- then the workaround can go."""
    _ensure_guard_so()
    rc, err = _dies("_guarded_cnn_leg", env={"MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC": "1", "AMD_SERIALIZE_KERNEL": "3",
                                             "AMD_LOG_LEVEL": "3"}, dtype="float16", mode="fp32")
    names = [ln.split("ShaderName :")[1].strip() for ln in err.splitlines() if "ShaderName :" in ln]
    print(f"child rc {rc}; last kernels launched: {names[-3:]}")
    _environment_assumption(rc != 0 and "Memory access fault" in err,
                            f"the fp32 leg survived MIOpen's NHWC backward-data solver (rc {rc}): a fixed MIOpen? then neuradar_amd/__init__.py's workaround can go")
    _environment_assumption(bool(names) and names[-1].startswith("igemm_bwd_gtc"), f"the last kernel before the fault was {names[-1:]}, not igemm_bwd_gtc*")


# ---- whole test files under the guard allocator ----------------------------------------------------------------------------------
def _guarded_pytest(files):
    """pytest, in this (child) process, on `files` with the guard allocator installed: every kernel of every test of those files
    runs on tensors that end at the end of their own mapped region."""
    lib = _install_guard_allocator()
    os.environ["NR_TEST_XDIST"] = "0"
    here = os.path.dirname(os.path.abspath(__file__))
    rc = pytest.main([*(os.path.join(here, f) for f in files), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-p", "no:xdist",
                      "-k", "not test_full_model_workloads and not test_cnn_16_bit"])  # (those two spawn children of their own)
    torch.cuda.synchronize()
    assert int(rc) == 0, f"pytest under the guard allocator: exit code {int(rc)}"
    return {"allocations": int(lib.guard_allocations()), "canary_failures": int(lib.guard_canary_failures())}


SWEEP = [["test_gpu_parity.py"], ["test_gpu_lp.py", "test_gpu_renderers.py"], ["test_gpu_binned.py", "test_gpu_shared_scatter.py"],
         ["test_gpu_conv7.py", "test_gpu_pw.py", "test_gpu_encoder.py"], ["test_gpu_decoders.py", "test_gpu_radar_loss.py"],
         ["test_gpu_full_step.py"], ["test_gpu_fused_actors.py", "test_gpu_batch.py", "test_gpu_render_entry.py"], ["test_gpu_fullsize.py"]]


@pytest.mark.parametrize("files", SWEEP, ids=lambda f: "+".join(x[len("test_gpu_"):-3] for x in f))
def test_kernel_tests_under_the_guard_allocator(files):
    """The golden-parity and per-kernel test files once more, in a child whose allocator places every tensor at the end of its own
    mapped region (NR_TEST_GUARD_SWEEP=0 skips): an out-of-bounds access anywhere on the hot path or in the decoders' kernels -- at
    the goldens' sizes, ragged and empty cases included -- is a fault there, whatever state the caching allocator would be in."""
    if os.environ.get("NR_TEST_GUARD_SWEEP", "1") == "0":
        pytest.skip("NR_TEST_GUARD_SWEEP=0")
    _ensure_guard_so()
    r = run_child(__file__, "_guarded_pytest", timeout=1500, files=files)
    print(f"{files}: {r['allocations']} guarded allocations")
    assert r["canary_failures"] == 0


def _guarded_workload(workload: str, steps: int = 3):
    """bench.py's own step for `workload` at FULL size -- model, optimizers, device-side batch assembly, the pipelined fused step
    with its side streams (and, for the decoder workloads, the decoder segment) -- launched eagerly under the guard allocator."""
    lib = _install_guard_allocator()
    import bench
    from neuradar_amd.parallel import GradAllReducer

    dev = torch.device(DEV)
    wl = bench.WORKLOADS[workload]
    n_rays = wl["rays"]
    mlp_dtype = wl.get("mlp_dtype", "bfloat16")
    model = bench.build_model(wl, dev, mlp_dtype, 8192.0 if mlp_dtype == "float16" else 1.0)
    opts = bench.build_optimizers(model)
    reducer = GradAllReducer(None, buffers=[g for o in opts for g in o.grad_buffers()])
    scene = bench.SyntheticScene(dev, seed=1000, radar=wl.get("radar", "zod"))
    torch.manual_seed(1234)
    targets = (0.1 * torch.randn(n_rays, 32, device=dev), 5.0 + 50.0 * torch.rand(n_rays, 1, device=dev))
    fwd_bwd, _, stepper = bench.make_step(model, scene, opts, reducer, targets, n_rays, fused=True, fuse_optimizer=True,
                                          mixed=wl if "cam_rays" in wl else None)
    losses_ = []
    for _ in range(steps):
        fwd_bwd()
        torch.cuda.synchronize()
        losses_.append(float(stepper.loss.sum()))
    assert all(x == x and abs(x) < 1e30 for x in losses_), losses_
    for n_, p_ in model.named_parameters():
        assert bool(torch.isfinite(p_).all()), f"parameter {n_}"
    return {"allocations": int(lib.guard_allocations()), "canary_failures": int(lib.guard_canary_failures()), "losses": losses_}


@pytest.mark.parametrize("workload", ["mixed16384_neuradar", "cam4096_l16f2_w64", "mixed16384_neuradar_full_fp16", "mixed8192_vod_nll"])
def test_bench_workloads_at_full_size_under_the_guard_allocator(workload):
    """BASELINE configs[2] (the headline), configs[1], configs[4] and configs[3] per-GPU shapes, exactly as bench.py builds them, three
    optimizer steps each with every tensor at the end of its own mapped region: no kernel of the timed path -- batch assembly,
    gathers, field, render, the three scatters (binned, block-shared), Adam, the decoder segment and its losses -- reads or writes
    behind a buffer at the sizes the numbers are quoted on (a size-independent property: a fault kills the child)."""
    if os.environ.get("NR_TEST_GUARD_SWEEP", "1") == "0":
        pytest.skip("NR_TEST_GUARD_SWEEP=0")
    _ensure_guard_so()
    r = run_child(__file__, "_guarded_workload", timeout=1200, workload=workload)
    print(f"{workload}: {r['allocations']} guarded allocations, losses {r['losses']}")
    assert r["canary_failures"] == 0

