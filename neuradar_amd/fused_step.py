"""Autograd-free training step of the hot path: launched back-to-back through the C ABI over buffers
allocated once -- forward (with the fused per-ray launches nr_power_bins_contract, nr_proposal_round,
nr_render_train), hand-chained backward, then the fused Adam.  29 launches per step (vs ~145 on the
modular torch.autograd path), on three HIP streams, replayable as a single hipGraph.

Sequence = models/neuradar.py:495-548 (get_nff_outputs, training branch) + the bench loss
 + backward + optimizer.  Checked against the autograd path and the oracle in
tests/test_gpu_parity.py::test_fused_step_matches_autograd_path.
"""
import os
from ctypes import byref, c_void_p
from typing import Dict, List, Optional

import torch
from torch import Tensor, nn

from . import _lib, losses, ops
from ._lib import NrField, NrFieldGrads, check
from .step import SKY_DISTANCE, NeuRadarHotPath


def flatten_parameters(params: List[nn.Parameter]) -> Dict[str, Tensor]:
    """Re-home `params` (and their .grad) as views of one flat buffer each, so the optimizer and the
    gradient all-reduce touch them with ONE launch.  Parameter objects, names and shapes are unchanged
    (state_dict round-trips as before).  4-D parameters -- the RGB CNN's convolution weights -- are laid out channels-last
    inside the buffer ([O, kh, kw, I] in memory, the same logical [O, I, kh, kw]): with the camera patches arriving as
    [P, h, w, C] rows, MIOpen then runs its NHWC kernels without the layout copies and transposes around every convolution
    (the CNN's share of the full-model step 2.7 -> 1.9 ms); Adam is elementwise, so the order inside the buffer is free."""
    dev, n = params[0].device, sum(p.numel() for p in params)
    n_pad = (n + 3) // 4 * 4
    flat = torch.zeros(n_pad, device=dev, dtype=torch.float32)
    flat_grad = torch.zeros_like(flat)
    off = 0
    for p in params:
        k = p.numel()
        if p.dim() == 4:
            o, i, kh, kw = p.shape
            view = lambda t: t[off:off + k].view(o, kh, kw, i).permute(0, 3, 1, 2)  # noqa: E731
        else:
            view = lambda t: t[off:off + k].view_as(p)  # noqa: E731
        view(flat).copy_(p.data)
        p.data = view(flat)
        p.grad = view(flat_grad)
        off += k
    return {"param": flat, "grad": flat_grad}


class SegmentedStep:
    """A data-parallel training step as hipGraph SEGMENTS cut where the host must act (world > 1).

    torch.distributed collectives cannot sit inside a captured hipGraph portably (gloo is host-driven; capture of RCCL launches
    cannot be verified on a one-GPU box), so a step with a reducer was launched eagerly: fine for the 29-launch headline step,
    host-bound for the 186-launch decoder workloads (5.5 ms of host time per step against 3.6 ms of GPU time).  Here the step
    function runs ONCE under stream capture; every host-side action inside it -- FusedTrainStep._host(fn): the wait for a
    deferred all-gather, the main table's gradient exchange + Adam, the remaining all-reduces + Adam launches -- ends the
    current graph, is RECORDED (not executed: nothing executes during capture) and, unless it is the last one, starts the next
    graph.  replay() then alternates graph launches and the recorded host actions:

        g0 [rays -> sampling rounds]  |host: wait for the previous step's table all-gather|
        g1 [main gather ... render, decoders, field backward, main scatter]  |host: main table exchange + Adam on the comm stream|
        g2 [proposal scatters -- beside that exchange]  |host: proposal table / small parameters / other optimizers|

    Every cut is a point where all forked streams have joined (FusedTrainStep joins them in front of each _host call when a
    recorder is attached).  Reference behaviour: one DDP all-reduce per step overlapped with the backward,
    pipelines/base_pipeline.py:305-307, scripts/train.py:104,145."""

    def __init__(self, stepper: "FusedTrainStep") -> None:
        self.stepper = stepper
        self.parts: List[tuple] = []  # (graph or None, host action or None), in order
        self._graph = None
        self._pool = None
        self.host_ms = None

    # ---- recorder protocol (called by FusedTrainStep._host during capture)
    def _begin(self) -> None:
        self._graph = torch.cuda.CUDAGraph()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        # thread_local: with a process group alive, torch's NCCL / RCCL watchdog thread polls events of earlier collectives while
        # this thread captures; under the default "global" capture mode such a call from ANOTHER thread invalidates the capture
        # and the watchdog takes the process down ("operation not permitted when stream is capturing" -- seen once in four runs
        # of bench.py --one-rank-collectives).  Launches of the decoders' autograd threads onto the capturing streams are
        # captured in either mode.
        self._graph.capture_begin(pool=self._pool, capture_error_mode="thread_local")

    def cut(self, fn, final: bool) -> None:
        self._graph.capture_end()
        self.parts.append((self._graph, fn))
        self._graph = None
        if not final:
            self._begin()

    def capture(self, step_fn) -> "SegmentedStep":
        """step_fn(): one call of the step (FusedTrainStep.forward_backward with optimizers and reducer) on static buffers.  Run
        a few eager steps first (lazy allocations, MIOpen's search).  Nothing is executed here."""
        assert not self.parts, "already captured"
        torch.cuda.synchronize()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        self.stepper._segmenter = self
        try:
            with torch.cuda.stream(cap):
                self._begin()
                try:
                    step_fn()
                except BaseException:
                    # a step that raises mid-capture must not leave the stream capturing (every later launch on it would fail with
                    # "operation not permitted when stream is capturing") nor a half-recorded list of segments behind
                    if self._graph is not None:
                        try:
                            self._graph.capture_end()
                        except Exception:  # noqa: BLE001 -- the capture may already be invalidated; ending it is best effort
                            pass
                        self._graph = None
                    self.parts = []
                    raise
                if self._graph is not None:  # the step ended without a final host action
                    self._graph.capture_end()
                    self.parts.append((self._graph, None))
                    self._graph = None
        finally:
            self.stepper._segmenter = None
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        return self

    def replay(self) -> None:
        for graph, fn in self.parts:
            graph.replay()
            if fn is not None:
                fn()


class FusedTrainStep:
    def __init__(self, model: NeuRadarHotPath, n_rays: int, overlap: bool = True, coherent_rays: Optional[int] = None) -> None:
        """coherent_rays: number of LEADING rays that come in spatially coherent groups (camera patches); their
        per-sample rows are stored sample-major, the remaining (lidar / radar) rays ray-major -- see
        nr_contract_gaussians in include/neuradar_hip.h.  Default: all rays."""
        c = model.config
        self.overlap = overlap and os.environ.get("NR_STEP_OVERLAP", "1") != "0"  # tuning knob
        self._streams = None
        self._segmenter = None  # SegmentedStep while it captures this step (world > 1): see _host
        self._comm = None
        self.timers = None  # dict name -> [(start, end) events]: set by a caller that wants in-step kernel times (eager only)
        assert len(c.num_proposal_samples) == 2
        if not c.field.use_sdf:
            raise NotImplementedError("FusedTrainStep composites sigmoid-SDF alphas (use_sdf=True, the reference's default); "
                                      "the density branch (neuradar.py:1018-1022) runs on the modular path")
        self.model, self.cfg, self.B = model, c, n_rays
        self.sm = n_rays if coherent_rays is None else int(coherent_rays)
        self.early_fork = os.environ.get("NR_EARLY_FORK")  # schedule override for A/B runs, see forward_backward
        # Proposal forward: grid + density head in ONE launch (a thread walks all six levels of its sample) for batches of
        # coherent rows only (+1 ... +1.6 % on the camera-only workloads); with incoherent rows in the batch the level-major
        # launch (all samples of level 0, then level 1, ...: one 4-MB level table at a time in each XCD's 4-MB L2) + a separate
        # density head wins: mixed batch 3.11 -> 3.01 ms per step, same call.  NR_FUSE_PROP_FWD=0/1 overrides.
        self.fuse_prop_fwd = os.environ.get("NR_FUSE_PROP_FWD", "1" if self.sm >= n_rays else "0") != "0"
        self.lib = _lib.lib()
        dev = next(model.parameters()).device
        self.dev = dev
        B = n_rays
        self.S = (*c.num_proposal_samples, c.num_nerf_samples)
        f32 = dict(device=dev, dtype=torch.float32)
        self.prop = model.proposal_fields[-1]  # both rounds use proposal_fields[1] (neuradar.py:302 quirk)
        if model.field.hashgrid.config.layout != "torch" or self.prop.hashgrid.config.layout != "torch":
            raise NotImplementedError("the fused step runs on the torch table layout; tcnn-layout tables (tcnn_compat) use the modular path")
        self.pgrid, self.mgrid = self.prop.hashgrid.static_grid, model.field.hashgrid.static_grid
        self.nears = torch.zeros(B, **f32)
        self.fars = torch.empty(B, **f32)
        self.sp, self.eu, self.x01, self.std, self.feats, self.g_feats = [], [], [], [], [], []
        for lvl, S in enumerate(self.S):
            grid = self.pgrid if lvl < 2 else self.mgrid
            self.sp.append(torch.empty(B, S + 1, **f32))
            self.eu.append(torch.empty(B, S + 1, **f32))
            self.x01.append(torch.empty(B * S, 3, **f32))
            self.std.append(torch.empty(B * S, **f32))
            self.feats.append(torch.empty(grid.num_levels, B * S, grid.features_per_level, **f32))
            self.g_feats.append(torch.empty_like(self.feats[-1]))
        # level-0 bins / contracted samples and the clamped far planes exist twice: `prepare` may fill the other
        # slot for the NEXT step while this step is still running (its scatter reads x01[0] at the very end)
        S0 = self.S[0]
        self._slots = [dict(sp=self.sp[0], eu=self.eu[0], x01=self.x01[0], std=self.std[0], fars=self.fars, fars_is_sky=False),
                       dict(sp=torch.empty(B, S0 + 1, **f32), eu=torch.empty(B, S0 + 1, **f32), x01=torch.empty(B * S0, 3, **f32),
                            std=torch.empty(B * S0, **f32), fars=torch.empty(B, **f32), fars_is_sky=False)]
        self.dens = [torch.empty(B, S, **f32) for S in self.S[:2]]
        self.g_dens = [torch.empty(B, S, **f32) for S in self.S[:2]]
        self.w = [torch.empty(B, S, **f32) for S in self.S]       # proposal weights x2, final weights
        self.g_w = [torch.empty(B, S, **f32) for S in self.S]
        self.prop_depth = [torch.empty(B, **f32) for _ in range(2)]
        Sm, C = self.S[2], c.field.nff_out_dim
        self.C = C
        self.feature = torch.empty(B * Sm, C, **f32)
        self.sdf = torch.empty(B * Sm, **f32)
        self.alpha = torch.empty(B * Sm, **f32)
        self.acc = torch.empty(B, **f32)
        self.features = torch.empty(B, C, **f32)
        self.depth = torch.empty(B, **f32)
        # partial sums; loss value = self.loss.sum().  The kernels add into the first NR_LOSS_SLOTS entries with atomics; the last
        # entry belongs to the decoder segment's plain torch add (it runs beside the proposal chains' loss kernels)
        self.loss = torch.zeros(_lib.NR_LOSS_SLOTS + 1, **f32)
        self.g_features = torch.zeros(B, C, **f32)
        self.g_depth = torch.empty(B, **f32)
        self.g_alpha = torch.empty(B * Sm, **f32)
        self.g_feature = torch.empty(B * Sm, C, **f32)
        # lidar rays of the batch (set_lidar): carving masks on the weights of all three levels; with an appearance embedding
        # and the lidar decoder in the model, the decoder and its two losses run inside the step on the lidar rows
        self.lidar = None
        self.dec = None  # set_decoders
        self.amp = None  # set_grad_scaler
        self.g_features_extra = None
        for p in model.parameters():
            if p.requires_grad and p.grad is None:
                p.grad = torch.zeros_like(p)
        # dynamic actors: per-level assignment buffers; the per-ray candidates / transforms are rebuilt each step
        self.hg_main, self.hg_prop = model.field.hashgrid, self.prop.hashgrid
        self.n_actors = self.hg_main.n_actors
        if self.n_actors:
            i32 = dict(device=dev, dtype=torch.int32)
            self.a_slot = [torch.empty(B * S, **i32) for S in self.S]
            self.a_x01 = [torch.empty(B * S, 3, **f32) for S in self.S]
            self.a_std = [torch.empty(B * S, **f32) for S in self.S]
            self.a_dirs = torch.empty(B * Sm, 3, **f32)
            self.g_w2b = torch.zeros(B, self.hg_main.MAX_CANDIDATES, 3, 4, **f32)
            self.fuse_prop_fwd = False  # actor features are written over the grid's before the density head reads them
            for hg in (self.hg_main, self.hg_prop):
                hg.actor_table_grads()  # raises unless tables and gradients live in their flat buffers
        self._structs()
        # Table scatters go through the two-pass slice-owner kernels where the table allows it (nr_hash_encode_bwd_binned:
        # the proposal grids; a main grid of <= 2^20 floats per level) when the batch holds incoherent rows (lidar rays
        # behind the first self.sm rays) -- then for ALL its rows: mixed batch after 1 500 steps 3.73 ms per step, against
        # 3.81 / 3.95 with the camera / camera + radar rows left to the merging kernel and 4.52 without the binned kernels
        # (freshly initialised: 3.13 / 3.08 / 3.09 / 3.46).  Camera-only batches stay with the merging kernel, which folds
        # neighbouring pixels before anything leaves the wave (16 384 rays: 2.62 against 2.73 ms after 1 500 steps).
        # NR_BINNED=0 / all / lidar and NR_BINNED_FROM=<first binned ray> override (A/B runs).
        # width of the binned scatter's fixed-point tile sums: 64 bits with fp32 MLPs (every addend exact to fp32's resolution),
        # 32 bits where the field MLPs already run on 16-bit operands -- an addend is rounded to 2^-22..2^-21 of the largest
        # contribution of its 512-row tile, far below the operands' own rounding (u = 2^-8 / 2^-11): step 2.55 -> 2.38 ms fresh,
        # 2.83 -> 2.72 ms after 1 500 steps (profiles/r03_ab_runs.txt).  NR_BIN_SUM_BITS overrides.
        self.bin_sum_bits = int(os.environ.get("NR_BIN_SUM_BITS", "64" if model.field.config.mlp_dtype == "float32" else "32"))
        self.binned_ws = [None, None, None]
        self.binned_from = 0 if self.sm < B else B  # first ray whose rows go through the binned kernels
        mode = os.environ.get("NR_BINNED", "")
        if mode in ("all", "lidar"):
            self.binned_from = 0 if mode == "all" else self.sm
        if os.environ.get("NR_BINNED_FROM"):
            self.binned_from = int(os.environ["NR_BINNED_FROM"])
        if os.environ.get("NR_BINNED", "1") != "0" and self.binned_from < B:
            for lvl, S in enumerate(self.S):
                grid = self.pgrid if lvl < 2 else self.mgrid
                need = self.lib.nr_hash_encode_bwd_binned_workspace_bytes(grid.num_levels, grid.features_per_level,
                                                                          grid.log2_hashmap_size, (B - self.binned_from) * S)
                if need > 0:
                    self.binned_ws[lvl] = torch.empty(need, device=dev, dtype=torch.uint8)
        # both proposal rounds scatter into the same table (proposal_fields[1]): ONE bin + ONE apply pass for the two
        # (nr_prop_density_scatter_binned2) when the density head rides inside the scatter for both -- NR_MERGE_PROP_SCATTER=1.
        # Measured SLOWER (same call, graph replay: 2.75 vs 2.55 ms fresh, 3.30 vs 3.00 ms after 1 500 steps,
        # profiles/r03_ab_runs.txt): the saved apply pass is worth less than the two rounds' bin passes sharing the CUs'
        # LDS from two streams, and the heads of both chains then sit in front of the one launch.  Default: a pair per round.
        self.merged_ws = None
        if (self.binned_ws[0] is not None and self.binned_ws[1] is not None and self.binned_from == 0 and not self.n_actors
                and self.pgrid.num_levels <= 8 and os.environ.get("NR_FUSE_DENSITY_BWD", "1") != "0"
                and os.environ.get("NR_MERGE_PROP_SCATTER", "0") == "1"):
            up = lambda v: (v + 511) // 512 * 512  # noqa: E731
            need = self.lib.nr_hash_encode_bwd_binned_workspace_bytes(self.pgrid.num_levels, self.pgrid.features_per_level,
                                                                      self.pgrid.log2_hashmap_size, up(B * self.S[0]) + up(B * self.S[1]))
            if need > 0:
                self.merged_ws = torch.empty(need, device=dev, dtype=torch.uint8)
                self.binned_ws[0] = self.binned_ws[1] = self.merged_ws  # (the per-round workspaces are not needed)
        self.field_ws = torch.empty(self.lib.nr_field_bwd_workspace_floats(byref(self.field_struct), B * Sm), **f32)
        # the main grid's scatter: the block-shared LDS table (nr_hash_encode_bwd_shared) where it applies -- 4-float entries, at
        # most 8 levels, a table too large for the slice-owner kernels, no actor rows, and 32-bit tile sums allowed (16-bit MLP
        # operands: self.bin_sum_bits).  NR_MAIN_SHARED=0 / 1 overrides the last condition (A/B runs).
        mg_ = self.mgrid
        self.main_shared = (mg_.features_per_level == 4 and mg_.num_levels <= 8 and self.binned_ws[2] is None and not self.n_actors
                            and os.environ.get("NR_MAIN_SHARED", "1" if self.bin_sum_bits == 32 else "0") == "1")

    def set_lidar(self, is_lidar: Tensor, did_return: Tensor, lidar_range: Tensor, row0: int, n_lidar: int,
                  target_intensity: Optional[Tensor] = None, sensor_idx: Optional[Tensor] = None, slot: Optional[int] = None,
                  prop_depth_loss: bool = False) -> None:
        """Describe the lidar rays of the batch (they occupy rows [row0, row0 + n_lidar) of every per-ray array; the layout is
        fixed, the values may be rewritten in place between steps): is_lidar / did_return [B] uint8, lidar_range [B] =
        metadata["directions_norm"].  Adds to the step's loss, for those rays,
          * the carving terms of neuradar.py:529-531,537-541,637-638,650 (carving_mult / n_lidar on the final weights,
            prop_lidar_loss_mult * carving_mult / n_lidar on each proposal level), inside nr_render_train and
            nr_interlevel_loss_to_density;
          * when the model has an appearance embedding and a lidar decoder (config.appearance_dim > 0, config.lidar_decoder):
            decoder([rendered features | appearance(time, sensor_idx)]) -> intensity MSE on returning rays + ray-drop BCE
            (neuradar.py:432-452,692-700), with target_intensity [B] and sensor_idx [B] int64.  NOTE: this stand-alone lidar
            segment averages the intensity term over ALL returning rays and has no lidar depth terms; the reference's full
            lidar loss (95 % quantile mask on the depth L1, intensity on mask & returned, non-return targets, per-proposal
            depth losses: neuradar.py:612-650) is what `set_decoders` + prop_depth_loss=True compute.
          * prop_depth_loss=True: depth_loss_i of both proposal levels (neuradar.py:641-648,679-688; prop_lidar_loss_mult *
            depth_mult * mean over the lidar rays, non-returns pulled beyond 150 m), inside nr_interlevel_loss_to_density.
        slot: with pipelined batches (two buffer sets) call once per set; forward_backward(slot=k) uses set k's arrays."""
        from ._lib import NrLidarSup

        c, dev, B = self.cfg, self.dev, self.B
        assert is_lidar.dtype == torch.uint8 and did_return.dtype == torch.uint8 and is_lidar.numel() == B
        prev = self.lidar
        self.lidar = dict(is_lidar=is_lidar, did_return=did_return, range=lidar_range, row0=int(row0), n=int(n_lidar))

        def sup(weight):
            s = NrLidarSup()
            s.is_lidar, s.did_return, s.range = is_lidar.data_ptr(), did_return.data_ptr(), lidar_range.data_ptr()
            s.carving_epsilon, s.non_return_distance, s.weight = c.carving_epsilon, c.non_return_lidar_distance, weight
            s.depth_weight, s.non_return_loss_mult = 0.0, c.non_return_loss_mult
            return s

        self.lidar["main"] = sup(c.carving_mult / max(n_lidar, 1))
        self.lidar["prop"] = sup(c.prop_lidar_loss_mult * c.carving_mult / max(n_lidar, 1))
        if prop_depth_loss:
            self.lidar["prop"].depth_weight = c.prop_lidar_loss_mult * c.depth_mult / max(n_lidar, 1)
        self.lidar["decoder"] = c.appearance_dim > 0 and c.lidar_decoder and target_intensity is not None
        if self.lidar["decoder"]:
            A = c.appearance_dim
            f32 = dict(device=dev, dtype=torch.float32)
            self.lidar.update(target_intensity=target_intensity, sensor_idx=sensor_idx, x=torch.empty(n_lidar, self.C + A, **f32),
                              y=torch.empty(n_lidar, 2, **f32), g_y=torch.empty(n_lidar, 2, **f32),
                              g_x=torch.empty(n_lidar, self.C + A, **f32), inv_ret=torch.ones(1, **f32))
            self.g_features_extra = torch.zeros(B, self.C, **f32)  # rows outside the lidar segment stay zero
            dec = self.model.lidar_decoder
            ws, bs = dec.weights()
            self.lidar["mlp"] = ops._mlp_struct(ws, bs)
            self.lidar["mlp_grads"] = ops._mlp_grads_struct([w.grad for w in ws], [b.grad for b in bs])
        if slot is not None:
            slots = prev if isinstance(prev, list) else [None, None]
            slots[slot] = self.lidar
            self.lidar = slots

    def set_decoders(self, head, batches, sensor_idx: Tensor) -> None:
        """Supervise through the modality decoders (decoder_losses.DecoderLossHead: RGB CNN on the camera patches, lidar MLP +
        the reference's quantile-masked lidar losses, radar transformer + heads + Hungarian-matched radar loss) instead of the
        bench loss on the rendered features / depth: between nr_field_fwd and nr_render_train the step composites, runs the
        head's autograd segment (HIP launches + torch glue, no host read) and hands d loss / d features, d loss / d depth to
        the render backward.  batches[slot]: the dict DecoderLossHead.losses reads (image, did_return, range,
        target_intensity, directions_spher, radar, radar_seg), one per buffer set; sensor_idx [B] int64.  forward_backward
        then needs the rays' `times`."""
        assert self.C == self.cfg.field.nff_out_dim
        self.dec = dict(head=head, batches=list(batches), sensor_idx=sensor_idx)
        if self.amp is not None:
            head.amp = self.amp

    def set_grad_scaler(self, amp, field_groups=(0, 1)) -> None:
        """Dynamic loss scale with found-inf guard (step.GradScalerState = the reference's GradScaler, engine/trainer.py:200,
        572-594) for a step whose MLPs run on fp16 operands: the 16-bit field backward takes its scale from the state and flags
        the optimizers `field_groups` (indices into the `optimizers` the state is attached to: the tables' and the fields') when
        it writes an inf / NaN; the small gradient buffers are checked before their Adam; every flagged optimizer skips its
        update; the scale is updated at the end of the step.  All on the device: the step stays graph-capturable."""
        self.amp, self._amp_field_groups = amp, tuple(field_groups)
        self.field_struct.amp = amp.buf.data_ptr()
        self.field_struct.amp_groups = sum(1 << g for g in field_groups)
        if self.dec is not None:
            self.dec["head"].amp = amp

    def _timed(self, name: str, launch):
        """Run `launch()` (one library call on the current stream); with self.timers set, bracket it with
        HIP events recorded on that stream."""
        if self.timers is None:
            return launch()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = launch()
        b.record()
        self.timers.setdefault(name, []).append((a, b))
        return rc

    def kernel_times(self) -> Dict[str, float]:
        """Mean seconds per timed launch (call after torch.cuda.synchronize())."""
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v) * 1e-3 for k, v in (self.timers or {}).items()}

    def _side_streams(self):
        if self._streams is None:
            pr = int(os.environ.get("NR_SIDE_PRIORITY", "0"))  # (-1: high -- measured: 2.17 -> 2.32 ms, profiles/r04_ab_runs.txt item 19)
            self._streams = [torch.cuda.Stream(device=self.dev, priority=pr), torch.cuda.Stream(device=self.dev, priority=pr)]
        return self._streams

    def _structs(self) -> None:
        """(Re)build the ctypes views of the parameters -- call again if parameters were re-homed."""
        fld = self.model.field
        gw, gb = fld.mlp_geo.weights()
        fw, fb = fld.mlp_feature.weights()
        self.field_struct = NrField()
        self.field_struct.geo, self.field_struct.feat = ops._mlp_struct(gw, gb), ops._mlp_struct(fw, fb)
        self.field_struct.beta = fld.sdf_to_density.beta.data_ptr()
        self.field_struct.dtype = _lib.NR_DTYPES[fld.config.mlp_dtype]
        self.field_struct.grad_scale = float(fld.config.mlp_grad_scale)
        # weight image every block of the field kernels copies into LDS; rebuilt at the start of each step
        self.field_image = torch.empty(self.lib.nr_field_image_floats(byref(self.field_struct)), device=self.dev)
        self.field_struct.packed = self.field_image.data_ptr()
        # activation stash: field_fwd leaves e / hidden activations / sdf there, the backward's feature half reads
        # them instead of recomputing the forward (648 B per sample at width 64)
        if os.environ.get("NR_FIELD_STASH", "1") != "0" and self.field_struct.dtype == 0:
            n_main = self.B * self.S[2]
            self.field_stash = torch.empty(self.lib.nr_field_stash_floats(byref(self.field_struct), n_main), device=self.dev)
            self.field_struct.stash = self.field_stash.data_ptr()
        self.field_grads = NrFieldGrads()
        self.field_grads.geo = ops._mlp_grads_struct([w.grad for w in gw], [b.grad for b in gb])
        self.field_grads.feat = ops._mlp_grads_struct([w.grad for w in fw], [b.grad for b in fb])
        beta = fld.sdf_to_density.beta
        if beta.grad is None:  # learnable_beta=False: the kernels still write d_beta somewhere
            self._beta_grad_sink = torch.zeros_like(beta)
        self.field_grads.beta = (beta.grad if beta.grad is not None else self._beta_grad_sink).data_ptr()
        if self.amp is not None:
            self.set_grad_scaler(self.amp, self._amp_field_groups)

    # -------------------------------------------------------------------------------------------
    def prepare(self, slot: int, origins: Tensor, directions: Tensor, pixel_area: Tensor, fars: Optional[Tensor],
                t_rand: Tensor) -> None:
        """First launch of a step, split off so that a caller can run it EARLY (for the next step, on a side
        stream, while the current step's backward runs): far-plane clamp (neuradar.py:573) + initial bins +
        contracted Gaussians into buffer set `slot`.  forward_backward(..., slot=slot, prepared=True) consumes it."""
        lib, p, c, B = self.lib, ops._p, self.cfg, self.B
        sl = self._slots[slot]
        if fars is None:  # camera rays: fars = 1e6 (cameras.py:948), clamped to the sky distance: a constant
            if not sl["fars_is_sky"]:
                sl["fars"].fill_(SKY_DISTANCE)
                sl["fars_is_sky"] = True
        else:
            torch.clamp(fars.reshape(-1), max=SKY_DISTANCE, out=sl["fars"])
            sl["fars_is_sky"] = False
        check(lib.nr_power_bins_contract(p(self.nears), p(sl["fars"]), p(t_rand), p(origins), p(directions), p(pixel_area), B,
                                         self.S[0], c.power_lambda, c.power_scaling, self.model.field.hashgrid.static_scale,
                                         self.sm, p(sl["sp"]), p(sl["eu"]), p(sl["x01"]), p(sl["std"]), ops._stream()), "power_bins")

    def _adam_stream(self):
        if getattr(self, "_adam_s", None) is None:
            self._adam_s = torch.cuda.Stream(device=self.dev)
        return self._adam_s

    def _host(self, fn, final: bool = False) -> None:
        """A host-side action of the step (a collective, a wait for one, optimizer launches ordered behind one): executed in
        place -- or, while a SegmentedStep captures the step, recorded as the boundary between two graph segments."""
        # (a segmenter cuts only the step the joins in front of the cuts are written for -- reducer AND fused optimizers:
        # forward_backward asserts it)
        if self._segmenter is None:
            fn()
        else:
            self._segmenter.cut(fn, final)

    def forward_backward(self, origins: Tensor, directions: Tensor, pixel_area: Tensor, fars: Tensor,
                         target_features: Tensor, target_depth: Tensor, t_rand: Tensor, jitter1: Tensor,
                         jitter2: Tensor, optimizers=None, reducer=None, after_sampling=None, slot: int = 0,
                         prepared: bool = False, times: Optional[Tensor] = None, flips=None, grads_accumulated: bool = False) -> Tensor:
        """Inputs: origins/directions [B,3], pixel_area [B] (already x9 for camera rays), fars [B],
        targets [B,C] / [B], jitters.  Accumulates into every parameter's .grad; returns the loss as
        NR_LOSS_SLOTS partial sums (call .sum() when the value is needed).

        optimizers = (table_opt, field_opt) (FlatAdam) fuses the optimizer into the step: each table is stepped on
        the stream of its own chain as soon as its scatter (and, with several ranks, its gradient exchange) is
        done; the small parameters follow when every chain has finished.

        slot / prepared: which of the two level-0 buffer sets the step uses, and whether `prepare(slot, ...)` has
        already filled it (then origins/directions/pixel_area must be the tensors given to prepare; fars and
        t_rand are ignored).

        after_sampling: optional callable run on a side stream once the sampling rounds have consumed the
        step's random numbers (t_rand, jitters) -- the caller refills them there for the NEXT step, off the
        critical path.

        times [B] / flips: with dynamic actors, the rays' times and the per-ray random x-flips (+-1 [B] each) of the three
        field evaluations (proposal round 0, round 1, main field; neurad_encoding.py:218-225), None = no flip.  optimizers may
        then carry a third entry, the trajectory optimizer.

        grads_accumulated: the parameters' .grad buffers already hold gradients that this step's optimizer must apply too
        (gradient accumulation, a modular backward before this call).  PRECONDITION otherwise, with fused `optimizers`: the main
        table's .grad is all zero on entry (the fused optimizers clear gradients every step, so steps chained through
        forward_backward(optimizers=...) satisfy it) -- the main table's Adam then trusts the `seen` bytes that THIS step's
        scatter marks and never reads the gradient of an unmarked group: a gradient left there by another writer would neither be
        applied nor cleared until its group is marked by a later step.  True falls back to the Adam launch that reads every gradient.

        reducer (parallel.GradAllReducer, world > 1; needs `optimizers`): data-parallel step.  The main table's
        gradient is exchanged as (row, value) lists right after its scatter (reduce_sparse), the proposal table's
        dense SUM all-reduce is issued on its side stream when both proposal scatters are done and runs over
        RCCL/xGMI beside the main chain, the small-parameter bucket last; Adam applies 1/world (DDP's mean)."""
        lib, p, c, B = self.lib, ops._p, self.cfg, self.B
        st = ops._stream()
        assert self.C <= 32 and (target_features is None or target_features.shape[1] == self.C)
        if c.appearance_dim > 0 and self.dec is None and not any(
                l_ is not None and l_["decoder"] for l_ in (self.lidar if isinstance(self.lidar, list) else [self.lidar])):
            raise NotImplementedError("the model has an appearance embedding but the step has no consumer for it: configure the "
                                      "decoders (set_decoders) or the lidar decoder segment (set_lidar(..., target_intensity=...)); "
                                      "the bench loss on the rendered features would silently ignore the embedding")
        main = torch.cuda.current_stream()
        side = self._side_streams() if self.overlap else [main, main]
        lam, scal = c.power_lambda, c.power_scaling
        o, d, area = p(origins), p(directions), p(pixel_area)
        sl = self._slots[slot]
        self.sp[0], self.eu[0], self.x01[0], self.std[0], self.fars = sl["sp"], sl["eu"], sl["x01"], sl["std"], sl["fars"]
        if not prepared:
            self.prepare(slot, origins, directions, pixel_area, fars, t_rand)
        nears, far = p(self.nears), p(self.fars)
        scale = self.model.field.hashgrid.static_scale
        geom = None
        if self.n_actors:  # per-ray candidates + (autograd) world->box transforms, shared by the three levels
            from types import SimpleNamespace

            assert times is not None, "dynamic actors need the rays' times"
            geom = self.hg_main.actor_geometry(SimpleNamespace(origins=origins, directions=directions, times=times.reshape(B, 1),
                                                               euclid=self.eu[0]), flip=None, draw_flip=False)
            geom["w2b_d"] = geom["w2b"].detach()
            geom["table_ids"] = self.hg_main.actor_table_ids()  # actor -> hash grid (actors.actor_to_id, neurad_encoding.py:183)
            flips = flips if flips is not None else (None, None, None)
            self.field_struct.sample_dirs = self.a_dirs.data_ptr()

        def actor_overwrite(lvl, grid):
            """Samples inside actor boxes: feats rows re-encoded from the actor's grid (after the static gather)."""
            hg = self.hg_prop if lvl < 2 else self.hg_main
            ag, S_, K_ = hg.actor_grids[0], self.S[lvl], hg.MAX_CANDIDATES
            st_ = ops._stream()
            check(lib.nr_actor_assign(o, d, area, p(self.eu[lvl]), B, S_, self.sm, p(geom["cand"]), K_, p(geom["w2b_d"]),
                                      p(geom["centres"]), p(geom["bounds"]), hg.config.actor.actor_scale, p(flips[lvl]),
                                      p(self.a_slot[lvl]), p(self.a_x01[lvl]), p(self.a_std[lvl]),
                                      p(self.a_dirs) if lvl == 2 else None, st_), "actor_assign")
            Fg = grid.features_per_level
            check(lib.nr_actor_encode_fwd(p(self.a_x01[lvl]), p(self.a_std[lvl]), p(self.a_slot[lvl]), p(geom["cand"]), K_, B, S_,
                                          self.sm, p(hg._actor_tables()), p(geom["table_ids"]), p(ag.scalings), ag.num_levels,
                                          ag.features_per_level, ag.log2_hashmap_size, p(self.feats[lvl]), Fg, B * S_ * Fg, grid.num_levels, st_),
                  "actor_encode_fwd")

        def actor_backward(lvl, grid):
            """Before the static scatter of a level: actor tables' gradients, zero rows for the static grid, pose gradients."""
            hg = self.hg_prop if lvl < 2 else self.hg_main
            ag, S_, K_ = hg.actor_grids[0], self.S[lvl], hg.MAX_CANDIDATES
            Fg = grid.features_per_level
            want_pose = lvl == 2 and hg.config.require_actor_grad and geom["w2b"].requires_grad
            check(lib.nr_actor_encode_bwd(p(self.a_x01[lvl]), p(self.a_std[lvl]), p(self.a_slot[lvl]), p(geom["cand"]), K_, B, S_,
                                          self.sm, p(hg._actor_tables()), p(geom["table_ids"]), p(ag.scalings), ag.num_levels,
                                          ag.features_per_level, ag.log2_hashmap_size, p(self.g_feats[lvl]), Fg, B * S_ * Fg, grid.num_levels,
                                          p(hg.actor_table_grads()), o, d, area, p(self.eu[lvl]), p(geom["w2b_d"]),
                                          hg.config.actor.actor_scale, p(flips[lvl]), p(self.g_w2b) if want_pose else None,
                                          ops._stream()), "actor_encode_bwd")
            if want_pose:  # tiny fixed-shape autograd graph: transforms -> trajectories (positions, 6-D rotations)
                torch.autograd.backward([geom["w2b"]], [self.g_w2b])

        # launches of the sampling part: bins+contraction (prepare), then per round hash grid + density -> [weights,
        # depth, resampling, contraction of the new samples].  Per-sample rows (positions, grid features, their
        # gradients) of the first self.sm rays are kept SAMPLE-major, row s*sm+b (include/neuradar_hip.h,
        # nr_contract_gaussians); per-ray arrays stay [B,S].
        # bookkeeping nothing on the sampling rounds depends on runs beside them (forked AFTER the first launch of
        # the critical path: a fork costs it ~10 us): loss slots, the field's weight image, the optimizers' schedule kernels
        if side[0] is not main:
            side[0].wait_stream(main)
        with torch.cuda.stream(side[0]):
            self.loss.zero_()
            if self.n_actors:
                self.g_w2b.zero_()
            check(lib.nr_field_pack(byref(self.field_struct), p(self.field_image), ops._stream()), "field_pack")
            if optimizers is not None:
                for o_ in optimizers:
                    o_.advance()
        # The main table's `seen` bytes set by its scatter (single GPU, fused optimizer, every row through the merging kernel; the
        # optimizer's buffer must be exactly the static table -- the actors' tables live in a buffer of their own):
        # NR_ADAM_MARKED=0 disables
        mark_seen = None
        if (optimizers is not None and reducer is None and self.binned_ws[2] is None and not grads_accumulated
                and os.environ.get("NR_ADAM_MARKED", "1") != "0"):
            t_opt = optimizers[0]
            i_m = t_opt.buffer_of(self.mgrid.hash_table)
            if (t_opt.seen[i_m] is not None and t_opt.buffers[i_m][0].data_ptr() == self.mgrid.hash_table.data_ptr()
                    and t_opt.buffers[i_m][0].numel() == self.mgrid.hash_table.numel() and i_m not in getattr(t_opt, "shards", {})):
                mark_seen = t_opt.seen[i_m]
                if not hasattr(t_opt, "marked"):
                    t_opt.marked = {}
                t_opt.marked[i_m] = True
        if mark_seen is None and optimizers is not None and hasattr(optimizers[0], "marked"):
            optimizers[0].marked.clear()
        # ... and, opt-in (NR_ADAM_SPLIT=1), its Adam in two launches around the scatter (FlatAdam.step_buffer_split).  Bit-identical
        # to the single launch and MEASURED SLOWER (2.15 -> 2.50 ms per step, same box, three interleaved repetitions): the
        # zero-gradient phase streams 0.8 GB beside the forward, whose gathers then take twice as long (field_fwd_gather 152 -> 325 us,
        # render 50 -> 367 us), and is still running when the scatter starts -- DESIGN.md section 13.  Default: one launch behind it.
        # (nr_adam_step_split has no weight-decay term: a table optimizer with weight decay stays on the single launch)
        adam_split = (mark_seen is not None and self.main_shared and self.amp is None and self.overlap
                      and optimizers[0].wd == 0.0 and os.environ.get("NR_ADAM_SPLIT", "0") == "1")
        jit = (jitter1, jitter2)
        pg, w_dec = self.pgrid, self.prop.density_decoder.weight
        for lvl in range(2):
            S, n = self.S[lvl], B * self.S[lvl]
            if self.fuse_prop_fwd:  # grid + density head in one launch
                check(self._timed(f"hash_encode_fwd[prop_s{S}]", lambda: lib.nr_prop_field_fwd(
                    p(self.x01[lvl]), p(self.std[lvl]), p(pg.hash_table), p(pg.scalings), pg.num_levels, pg.features_per_level,
                    pg.log2_hashmap_size, p(w_dec), p(self.feats[lvl]), pg.features_per_level, n * pg.features_per_level, n, S,
                    self.sm, p(self.dens[lvl]), st)), "prop_field_fwd")
            else:
                check(self._timed(f"hash_encode_fwd[prop_s{S}]", lambda: lib.nr_hash_encode_fwd(
                    p(self.x01[lvl]), p(self.std[lvl]), p(pg.hash_table), p(pg.scalings), pg.num_levels, pg.features_per_level,
                    pg.log2_hashmap_size, p(self.feats[lvl]), pg.features_per_level, n * pg.features_per_level, n, 0, st)), "hash_fwd")
                if self.n_actors:
                    actor_overwrite(lvl, pg)
                check(lib.nr_prop_density_fwd(p(self.feats[lvl]), pg.features_per_level, n * pg.features_per_level,
                                              pg.features_per_level, p(w_dec), w_dec.numel(), n, S, self.sm, p(self.dens[lvl]), st),
                      "prop_density")
            check(lib.nr_proposal_round(p(self.dens[lvl]), p(self.eu[lvl]), p(self.sp[lvl]), p(jit[lvl]), nears, far, o, d, area,
                                        B, S, self.S[lvl + 1], lam, scal, SKY_DISTANCE if lvl == 1 else 0.0, scale, self.sm,
                                        p(self.w[lvl]), p(self.prop_depth[lvl]), p(self.sp[lvl + 1]), p(self.eu[lvl + 1]),
                                        p(self.x01[lvl + 1]), p(self.std[lvl + 1]), st), "proposal_round")
        mg, Sm = self.mgrid, self.S[2]
        n = B * Sm
        F = mg.features_per_level
        # the main grid's gather inside the field forward (nr_field_fwd_gather: the per-level features go to the first layer in
        # registers; the [L, n, F] copy the backward recomputes from is written on the way): NeuRadar's grid (8 x 4) into the
        # 32-wide stack on 16-bit operands, no actor rows to patch in between.  NR_FUSE_MAIN_GATHER=0: two launches.
        fuse_gather = (mg.num_levels == 8 and F == 4 and self.field_struct.dtype != 0 and self.model.field.config.geo_hidden_dim == 32
                       and not self.n_actors and os.environ.get("NR_FUSE_MAIN_GATHER", "1") != "0")
        seg = self._segmenter if (reducer is not None and optimizers is not None) else None
        assert self._segmenter is None or seg is not None, "SegmentedStep captures the data-parallel step with fused optimizers (reducer and optimizers given)"
        if reducer is not None:
            # a deferred all-gather of the previous step's sharded table update lands here: first read of the main table
            if seg is not None and side[0] is not main:
                main.wait_stream(side[0])  # (a graph segment ends here: every forked stream joined)
            self._host(reducer.wait_table_sync)
        if not fuse_gather:
            check(self._timed(f"hash_encode_fwd[main_s{Sm}]", lambda: lib.nr_hash_encode_fwd(
                p(self.x01[2]), p(self.std[2]), p(mg.hash_table), p(mg.scalings), mg.num_levels, F, mg.log2_hashmap_size,
                p(self.feats[2]), F, n * F, n, 0, st)), "hash_fwd")
        if self.n_actors:
            actor_overwrite(2, mg)
        if side[0] is not main:
            main.wait_stream(side[0])
        if adam_split:
            # the sampling rounds are done (the main samples' positions are final) and the optimizers' schedule kernels have run:
            # stamp the table entries this step can touch and give every OTHER entry with a history its zero-gradient update now,
            # on a stream of its own beside the field's forward and backward -- the main table's Adam behind the scatter (the end of
            # the step's critical path) then only walks the stamped entries
            if getattr(self, "_stamp", None) is None:
                self._stamp = torch.zeros(mg.hash_table.numel() // 4, device=self.dev, dtype=torch.uint8)
            ev_ = torch.cuda.Event()
            ev_.record(main)
            adam_s = self._adam_stream()
            adam_s.wait_event(ev_)
            with torch.cuda.stream(adam_s):
                check(lib.nr_hash_mark_vertices(p(self.x01[2]), p(mg.scalings), mg.num_levels, mg.log2_hashmap_size, n, p(self._stamp),
                                                c_void_p(optimizers[0].step_t.data_ptr() + 4), ops._stream()), "hash_mark_vertices")
                optimizers[0].step_buffer_split(optimizers[0].buffer_of(mg.hash_table), 1, self._stamp)
        if after_sampling is not None:
            if side[1] is not main:
                side[1].wait_stream(main)
            with torch.cuda.stream(side[1]):
                after_sampling()
        if fuse_gather:
            check(self._timed(f"field_fwd+gather[main_s{Sm}]", lambda: lib.nr_field_fwd_gather(
                byref(self.field_struct), p(self.x01[2]), p(self.std[2]), p(mg.hash_table), p(mg.scalings), mg.num_levels, F,
                mg.log2_hashmap_size, p(self.feats[2]), n * F, d, Sm, self.sm, n, p(self.feature), p(self.sdf), p(self.alpha), st)),
                  "field_fwd_gather")
        else:
            check(self._timed("field_fwd", lambda: lib.nr_field_fwd(byref(self.field_struct), p(self.feats[2]), F, n * F, F, d, Sm, self.sm, n,
                                                                     p(self.feature), p(self.sdf), p(self.alpha), st)), "field_fwd")
        # composite + supervision + distortion + composite backward of the main level: one launch
        lid = self.lidar[slot] if isinstance(self.lidar, list) else self.lidar
        if lid is not None and lid["decoder"]:
            # per-ray lidar decoder on [rendered features | appearance] of the lidar rows, its losses and its backward; the
            # gradient on the rendered features re-enters nr_render_train as grad_features_extra
            emb = self.model.appearance_embedding.weight
            E, r0, nl_ = self.model._num_embeds_per_sensor, lid["row0"], lid["n"]
            assert times is not None, "the appearance embedding needs the rays' times"
            check(lib.nr_composite_fwd(p(self.alpha), p(self.feature), p(self.eu[2]), B, Sm, self.C, p(self.w[2]), p(self.acc),
                                       p(self.features), p(self.depth), st), "composite_fwd")
            check(lib.nr_appearance_concat_fwd(p(self.features), self.C, p(emb), emb.shape[1], p(times), p(lid["sensor_idx"]),
                                               c.duration, E, r0, nl_, p(lid["x"]), st), "appearance_fwd")
            check(lib.nr_mlp_fwd(byref(lid["mlp"]), p(lid["x"]), nl_, p(lid["y"]), st), "lidar_decoder_fwd")
            torch.div(1.0, lid["did_return"][r0:r0 + nl_].sum().clamp(min=1).float(), out=lid["inv_ret"][0])
            check(lib.nr_lidar_head_loss(p(lid["y"]), p(lid["target_intensity"][r0:]), p(lid["did_return"][r0:]), nl_,
                                         p(lid["inv_ret"]), c.intensity_mult, c.ray_drop_loss_mult, p(lid["g_y"]), p(self.loss), st),
                  "lidar_head_loss")
            check(lib.nr_mlp_bwd(byref(lid["mlp"]), p(lid["x"]), p(lid["g_y"]), nl_, p(lid["g_x"]), byref(lid["mlp_grads"]), st),
                  "lidar_decoder_bwd")
            check(lib.nr_appearance_concat_bwd(p(lid["g_x"]), self.C, emb.shape[1], p(times), p(lid["sensor_idx"]), c.duration, E,
                                               r0, nl_, p(self.g_features_extra), p(emb.grad), emb.shape[0], st), "appearance_bwd")
        # ---- backward.  The field's MFMA backward needs 1 wave/SIMD worth of registers and ~127 KB of LDS per
        #      workgroup, so LDS-heavy kernels sharing the CUs with it (and it with them) crawl: forking all three
        #      chains (main grid scatter + Adam / proposal round 1 / proposal round 0) right after the render launch
        #      measured 0.78 ms per step against 0.73 with field_bwd by itself.  The chains meet again in the optimizer. ----
        Fp = pg.features_per_level
        # Schedule (NR_EARLY_FORK selects one for A/B runs; tools/timeline.py shows the result).  With the activation
        # stash the feature half of field_bwd is short enough that the plain order wins or ties on all five bench
        # workloads (graph replay, one box: headline 0.63 / 0.66 / 0.65 ms for 0 / 3 / 4; NeuRadar field and mixed
        # batches: 0 = 3 < 4; only the 64-wide field at 16 384 rays prefers 4, by 1 %):
        #   0  nothing starts before field_bwd (default); 3 = 0 with the field's weight-gradient slabs summed on
        #      round 1's stream (nr_field_grad_reduce) instead of in front of the main scatter
        #   2  round 0's chain (side[1]) starts before field_bwd and shares the chip with it; 4 = 2 with the reduce
        #      moved as in 3 (was the best for the headline batch, by 1.5 %, while the backward still recomputed the forward)
        #   1  both proposal chains before field_bwd (+5...+18 % on one GPU).  Data-parallel steps use it nevertheless:
        #      the proposal table's dense all-reduce (25 MB over xGMI, a few hundred us) can then begin ~150 us earlier
        #      and hide behind field_bwd, the main scatter and the main table's Adam.  That choice is reasoned from
        #      the single-GPU timeline, not measured: no multi-GPU box this round.
        if self.early_fork is not None:
            early = int(self.early_fork)
        else:
            # data parallel: both proposal chains before field_bwd for camera-only batches (the 25-MB proposal table's
            # all-reduce starts earlier); mixed batches run the main grid's scatter and exchange first instead (`order`)
            early = 1 if reducer is not None and self.sm >= B else 0
            # with the decoders in the step, NR_PROP_BESIDE_DECODERS=1: both proposal chains BEFORE the decoder segment.  Their
            # inputs (the three levels' weights, spacings and densities) exist once the forward has composited -- the inter-level
            # loss does not see the decoders -- so the two bin + apply pairs (0.73 ms of kernel time) can run beside the segment's
            # ~300 small dependent launches.  Round 3's default; no longer: a captured step in which MORE THAN THREE branches are
            # in flight at once loses more than the overlap wins (same call, fp16 full-model workload, trained regime: CNN + lidar
            # on one stream, radar, two proposal chains = 4 branches 4.02 ms; the lidar chain on a stream of its own = 5 branches
            # 4.16; the proposal chains behind the segment and CNN | lidar | radar = 3 branches 3.79 -- DESIGN.md section 10)
            if self.dec is not None and os.environ.get("NR_PROP_BESIDE_DECODERS", "0") != "0":
                early = 5
        split_reduce = early in (3, 4)  # 3 / 4: schedule 0 / 2 + the reduce on side[0]
        # Order of the three table scatters.  "concurrent" (single process): all at once -- 2.5 % faster than either serial
        # order on one GPU.  "main_first" (data parallel): the main grid's scatter by itself, the proposal scatters after
        # it -- the main table's gradient exchange (the step's largest: 58 MB of row lists per rank on the mixed batch, or the
        # 537 MB dense all-reduce) can then start ~0.6 ms earlier and run beside the proposal scatters instead of after
        # them.  Reasoned from the single-GPU timeline like the early fork above, not measured (one GPU per call here).
        order = os.environ.get("NR_SCATTER_ORDER", "main_first" if reducer is not None and early == 0 else "concurrent")
        if self._segmenter is not None and reducer is not None and optimizers is not None:
            order = "main_first"  # (graph segments: the main table's exchange is cut in between the main and the proposal scatters)

        # proposal levels whose density-head backward rides inside the binned scatter (nr_prop_density_scatter_binned): all
        # rows binned, no actor rows to patch into the feature gradients -- the [L, n, F] gradient buffer is then never touched
        head_in_scatter = [self.binned_ws[l_] is not None and self.binned_from == 0 and not self.n_actors and pg.num_levels <= 8
                           and os.environ.get("NR_FUSE_DENSITY_BWD", "1") != "0" for l_ in (0, 1)]

        def chain_head(lvl):
            sp_ = ops._stream()
            S, nl = self.S[lvl], B * self.S[lvl]
            check(lib.nr_interlevel_loss_to_density(p(self.sp[2]), Sm + 1, p(self.w[2]), Sm, Sm - 1, p(self.sp[lvl]),
                                                    p(self.w[lvl]), p(self.dens[lvl]), p(self.eu[lvl]), S, B,
                                                    losses.PULSE_WIDTHS[lvl], c.interlevel_loss_mult, p(self.g_dens[lvl]),
                                                    p(self.loss), byref(lid["prop"]) if lid is not None else None,
                                                    sp_), "interlevel_loss")
            if not head_in_scatter[lvl]:
                check(lib.nr_prop_density_bwd(p(self.feats[lvl]), Fp, nl * Fp, Fp, p(w_dec), w_dec.numel(), nl, S, self.sm, p(self.dens[lvl]),
                                              p(self.g_dens[lvl]), p(self.g_feats[lvl]), p(w_dec.grad), sp_), "prop_density_bwd")

        # main table: scatter and Adam pipelined level by level (single GPU, marked Adam, block-shared scatter; NR_MAIN_LEVEL_PIPELINE)
        level_pipeline = int(os.environ.get("NR_MAIN_LEVEL_PIPELINE", "0")) if (mark_seen is not None and self.main_shared and self.amp is None
                                                                                 and not adam_split) else 0
        level_pipeline = min(level_pipeline, mg.num_levels)  # number of level GROUPS (0 = off: one scatter launch, then one Adam launch)
        level_pipeline_buffer = optimizers[0].buffer_of(mg.hash_table) if level_pipeline else None

        def scatter(lvl, grid, tag):
            """grad_table += scatter of g_feats[lvl] on the current stream: the merging kernel on the coherent rows, the
            binned one on the rows behind them (when a workspace was set up for this level)."""
            sp_ = ops._stream()
            S, nl, Fg = self.S[lvl], B * self.S[lvl], grid.features_per_level
            n_coh = nl if self.binned_ws[lvl] is None else self.binned_from * S

            if self.n_actors:
                actor_backward(lvl, grid)

            def launch():
                if lvl < 2 and head_in_scatter[lvl]:
                    return lib.nr_prop_density_scatter_binned_lp(p(self.x01[lvl]), p(self.std[lvl]), p(grid.scalings), grid.num_levels, Fg,
                                                                 grid.log2_hashmap_size, p(self.feats[lvl]), Fg, nl * Fg, p(w_dec),
                                                                 p(self.g_dens[lvl]), S, self.sm, p(grid.hash_table.grad), p(w_dec.grad),
                                                                 nl, p(self.binned_ws[lvl]), self.bin_sum_bits, sp_)
                # batches with incoherent rows: the wide per-wave table of the F = 4 merging kernel (fewer atomics per sample,
                # more of the chip left to the two binned scatters and Adam beside it: step -3 % fresh, -6 % after 1 500 steps)
                cells = 256 if self.sm < B and os.environ.get("NR_WIDE_MERGE", "1") != "0" else 0
                if lvl == 2 and self.main_shared:
                    # NeuRadar's main grid in a step on 16-bit MLP operands: the block-shared vertex-keyed LDS table on 32-bit
                    # integer atomics (grid_shared.hip) for every row -- two waves per SIMD and a third of the merging kernel's
                    # instructions; marks the optimizer's `seen` bytes itself when the marked Adam follows
                    if level_pipeline:
                        # one launch per GROUP of levels, and the group's Adam on a stream of its own as soon as its scatter is done,
                        # beside the next group's scatter: the step's critical path (main scatter 700 us -> main Adam 514 us, nothing
                        # beside the Adam) becomes ~ scatter + one level of Adam
                        T4 = (1 << grid.log2_hashmap_size) * Fg  # floats per level
                        cur_ = torch.cuda.current_stream()
                        adam_s = self._adam_stream()
                        per = -(-grid.num_levels // level_pipeline)  # levels per launch
                        for l_ in range(0, grid.num_levels, per):
                            k_ = min(per, grid.num_levels - l_)
                            rc = lib.nr_hash_encode_bwd_shared(
                                p(self.x01[lvl]), p(self.std[lvl]), c_void_p(grid.scalings.data_ptr() + 4 * l_), k_, Fg, grid.log2_hashmap_size,
                                c_void_p(self.g_feats[lvl].data_ptr() + 4 * l_ * nl * Fg), Fg, nl * Fg,
                                c_void_p(grid.hash_table.grad.data_ptr() + 4 * l_ * T4), nl,
                                c_void_p(mark_seen.data_ptr() + l_ * T4 // 4), sp_)
                            if rc != 0:
                                return rc
                            ev_ = torch.cuda.Event()
                            ev_.record(cur_)
                            adam_s.wait_event(ev_)
                            with torch.cuda.stream(adam_s):
                                optimizers[0].step_buffer(level_pipeline_buffer, 1.0, part=(l_ * T4, (l_ + k_) * T4))
                        return 0
                    # threads per row (ABI v28): this scatter BY ITSELF is 18 % faster with two threads sharing a row's 8 corners
                    # (two waves per SIMD on the one table a CU holds); beside the proposal scatters one thread per row wins
                    # (same-box step times in grid_shared.hip's header)
                    return lib.nr_hash_encode_bwd_shared_split(p(self.x01[lvl]), p(self.std[lvl]), p(grid.scalings), grid.num_levels, Fg,
                                                               grid.log2_hashmap_size, p(self.g_feats[lvl]), Fg, nl * Fg,
                                                               p(grid.hash_table.grad), nl, p(mark_seen), 2 if order == "main_first" else 1, sp_)
                if mark_seen is not None and lvl == 2:
                    # the main table's scatter sets the optimizer's `seen` bytes itself: Adam then skips never-touched groups
                    # on the byte alone instead of reading 4 B of gradient per parameter of the whole table (FlatAdam.marked)
                    rc = lib.nr_hash_encode_bwd_marked(p(self.x01[lvl]), p(self.std[lvl]), p(grid.scalings), grid.num_levels, Fg,
                                                       grid.log2_hashmap_size, p(self.g_feats[lvl]), Fg, nl * Fg,
                                                       p(grid.hash_table.grad), n_coh, 0, cells, p(mark_seen), sp_)
                    return rc
                rc = lib.nr_hash_encode_bwd_tuned(p(self.x01[lvl]), p(self.std[lvl]), p(grid.scalings), grid.num_levels, Fg,
                                                  grid.log2_hashmap_size, p(self.g_feats[lvl]), Fg, nl * Fg, p(grid.hash_table.grad),
                                                  n_coh, 0, cells, sp_) if n_coh > 0 else 0
                if rc == 0 and n_coh < nl:
                    rc = lib.nr_hash_encode_bwd_binned_lp(p(self.x01[lvl][n_coh:]), p(self.std[lvl][n_coh:]), p(grid.scalings),
                                                          grid.num_levels, Fg, grid.log2_hashmap_size,
                                                          p(self.g_feats[lvl][:, n_coh:, :]), Fg, nl * Fg, p(grid.hash_table.grad),
                                                          nl - n_coh, p(self.binned_ws[lvl]), self.bin_sum_bits, sp_)
                return rc

            check(self._timed(f"hash_encode_bwd[{tag}_s{S}]", launch), "hash_bwd")

        def chain_scatter(lvl):
            scatter(lvl, pg, "prop")

        merged = self.merged_ws is not None and all(head_in_scatter)
        if merged:
            split_reduce = False  # (schedules 3 / 4 park the weight-gradient reduce on side[0] in front of round 1's head)

        def merged_scatter():
            """Both proposal rounds' scatters (density-head backward inside) as one bin pass + one apply pass."""
            S0_, S1_ = self.S[0], self.S[1]
            n0_, n1_ = B * S0_, B * S1_
            check(self._timed(f"hash_encode_bwd[prop_s{S0_}+s{S1_}]", lambda: lib.nr_prop_density_scatter_binned2(
                p(self.x01[0]), p(self.std[0]), p(self.feats[0]), p(self.g_dens[0]), S0_, n0_,
                p(self.x01[1]), p(self.std[1]), p(self.feats[1]), p(self.g_dens[1]), S1_, n1_, self.sm,
                p(pg.scalings), pg.num_levels, Fp, pg.log2_hashmap_size, Fp, p(w_dec), p(pg.hash_table.grad), p(w_dec.grad),
                p(self.merged_ws), ops._stream())), "hash_bwd")
        chains = list(zip((1, 0), side))  # (level, stream): side[0] runs round 1 (s64), side[1] round 0 (s128)
        # (both proposal chains on one side stream, or on the main stream in front of the main scatter: +6 % / +8 % per
        # step on the mixed batch -- the three scatters and the main table's Adam do share the chip productively)
        before = {0: (), 1: (0, 1), 2: (1,), 3: (), 4: (1,), 5: (0, 1)}[early]  # side indices whose chain starts before field_bwd
        if merged and before:
            before = (0, 1)  # (a merged scatter needs both heads: the early-fork schedules start both chains early)

        # (experiment knob: both early chains on ONE side stream, one after the other -- one concurrent branch less beside the decoders)
        one_stream = early == 5 and os.environ.get("NR_PROP_ONE_STREAM", "0") == "1" and side[1] is not main

        def start_chains_before():
            """The proposal chains the schedule starts early: before nr_field_bwd, or (5) before the decoder segment."""
            for i_ in before:
                if side[i_] is not main and not (one_stream and i_ == 0):
                    side[i_].wait_stream(main)
            for i_ in before:
                # (merged: both heads and the scatter on ONE side stream -- a capture in which the two side streams wait for each
                # other in turn crashed hipStreamEndCapture on this ROCm build; the heads are ~80 us each)
                with torch.cuda.stream(side[1] if (merged or one_stream) else side[i_]):
                    chain_head(chains[i_][0])
                    if not merged:
                        chain_scatter(chains[i_][0])
            if merged and before:
                with torch.cuda.stream(side[1]):
                    merged_scatter()

        g_f_extra = self.g_features_extra if (lid is not None and lid["decoder"]) else None
        g_d_extra, rgb_mult, depth_mult = None, c.rgb_mult, c.depth_mult
        if self.dec is not None:
            # the decoders' losses replace the direct supervision: composite -> decoder segment (autograd over HIP launches)
            # -> gradients on the rendered features / depth, which re-enter the render backward below
            assert times is not None, "the decoders need the rays' times (appearance embedding)"
            check(lib.nr_composite_fwd(p(self.alpha), p(self.feature), p(self.eu[2]), B, Sm, self.C, p(self.w[2]), p(self.acc),
                                       p(self.features), p(self.depth), st), "composite_fwd")
            if early == 5:
                start_chains_before()
            epoch = optimizers[0].step_t if optimizers is not None else None
            g_f_extra, g_d_extra = self._timed("decoders", lambda: self.dec["head"].backward_into(
                self.features, self.depth, times, self.dec["sensor_idx"], self.dec["batches"][slot], self.loss, seed_epoch=epoch))
            rgb_mult = depth_mult = 0.0
            target_features = target_depth = None
        check(lib.nr_render_train(p(self.alpha), p(self.feature), p(self.eu[2]), p(self.sp[2]), p(target_features),
                                  p(target_depth), B, Sm, self.C, rgb_mult, depth_mult, c.distortion_loss_mult,
                                  p(self.w[2]), p(self.acc), p(self.features), p(self.depth), p(self.g_alpha),
                                  p(self.g_feature), p(self.loss), p(g_f_extra), p(g_d_extra),
                                  byref(lid["main"]) if lid is not None else None, st), "render_train")
        if early != 5:
            start_chains_before()
        check(self._timed("field_bwd", lambda: lib.nr_field_bwd(
            byref(self.field_struct), p(self.feats[2]), F, n * F, F, d, Sm, self.sm, n, p(self.g_feature), p(self.g_alpha), None,
            p(self.g_feats[2]), None if split_reduce else byref(self.field_grads), p(self.field_ws), st)), "field_bwd")
        amp_ev = None
        if self.amp is not None and optimizers is not None:
            # the found-inf flags of the tables' optimizer are final once the field backward has run: the proposal table's Adam
            # (a side stream that may have started before nr_field_bwd) waits for this point, the main table's follows it anyway
            if reducer is not None and seg is None:  # every rank must take the same decision: SUM of the ranks' flags (non-zero = found)
                reducer.start(self.amp.buf[self.amp._F:self.amp._F + 8])
                reducer.wait_all()
            if seg is None:  # (segmented: the flags are summed by the first host action behind the backward, see host_main_table)
                amp_ev = torch.cuda.Event()
                amp_ev.record(main)
        for i_, (lvl, stream) in enumerate(chains):
            late = i_ not in before
            if not (late or (split_reduce and i_ == 0)):
                continue
            if merged and not (split_reduce and i_ == 0):
                stream = side[1]
            if stream is not main:
                stream.wait_stream(main)
            with torch.cuda.stream(stream):
                if split_reduce and i_ == 0:
                    check(lib.nr_field_grad_reduce(byref(self.field_struct), p(self.field_ws), n, byref(self.field_grads),
                                                   ops._stream()), "field_grad_reduce")
                if late:
                    chain_head(lvl)
                    if order != "main_first" and not merged:
                        chain_scatter(lvl)
        if merged and not before and order != "main_first":
            with torch.cuda.stream(side[1]):
                merged_scatter()
        scatter(2, mg, "main")

        def join_all():
            for s_ in side:
                if s_ is not main:
                    main.wait_stream(s_)
            if adam_split:
                main.wait_stream(self._adam_stream())

        table_opt = field_opt = None
        if optimizers is not None:
            table_opt, field_opt = optimizers[:2]
            scale = 1.0 if reducer is None else 1.0 / reducer.world
            i_prop, i_main = table_opt.buffer_of(pg.hash_table), table_opt.buffer_of(mg.hash_table)
            shared = i_prop == i_main  # tiny tables (<= 65536 elements) live in ONE flat buffer: reduce and step it once,
            #                            after both tables' scatters (below, on side[0])

        def amp_flags_sum():
            if self.amp is not None and reducer is not None:  # every rank must take the same decision: SUM of the ranks' flags
                reducer.start(self.amp.buf[self.amp._F:self.amp._F + 8])
                reducer.wait_all()

        def main_table_step():
            """The main table's gradient exchange and Adam, on the current stream."""
            if reducer is not None and reducer.table_mode == "shard" and i_main in getattr(table_opt, "shards", {}):
                # reduce-scatter -> Adam on this rank's 1/world of the rows -> all-gather (parallel.shard_step)
                # (reducer.table_delta / defer_gather: the update deltas in bf16, the all-gather deferred into the next step)
                reducer.shard_step(table_opt, i_main, scale, transport=reducer.table_dtype,
                                   delta_dtype=getattr(reducer, "table_delta", None), defer=getattr(reducer, "defer_gather", False),
                                   row_width=mg.features_per_level)
            else:
                keep = None
                if reducer is not None:
                    if reducer.sparse_tables:  # a step touches ~1 % of the main table's rows: exchange those only
                        # (flag = 2 where a rank's list overflowed: the table's Adam then skips this step and keeps the gradient)
                        keep = reducer.reduce_sparse(table_opt.buffers[i_main][1], mg.features_per_level).get("flag")
                    else:
                        reducer.start(table_opt.buffers[i_main][1])
                        reducer.wait_all()
                if level_pipeline:
                    torch.cuda.current_stream().wait_stream(self._adam_stream())  # (the levels' Adam launches followed their scatters)
                elif adam_split:
                    table_opt.step_buffer_split(i_main, 2, self._stamp)  # the stamped entries; the rest was stepped beside the forward
                else:
                    table_opt.step_buffer(i_main, scale, skip_extra=keep)

        def prop_table_step():
            if reducer is not None:
                reducer.start(table_opt.buffers[i_prop][1])
                reducer.wait_all()  # stream-level wait: the stream continues once RCCL is done
            table_opt.step_buffer(i_prop, scale)

        def small_and_other_steps():
            if reducer is not None:  # small parameters: need field_bwd and both prop_density_bwd
                for _, g_ in field_opt.buffers:
                    reducer.start(g_)
                reducer.wait_all()
            for i in range(len(field_opt.buffers)):
                field_opt.check_buffer(i)  # (loss scaler attached: found-inf of the small parameters' gradients)
                field_opt.step_buffer(i, scale)
            # whatever else the table optimizer holds (per-actor grids) and the trajectory optimizer
            # ... except the tables of proposal_fields[:-1]: never evaluated (the reference's late-binding lambda,
            # neuradar.py:302), so their gradient is identically zero -- neither all-reduced nor walked by Adam
            dead = {p_.data_ptr() for f_ in self.model.proposal_fields[:-1] for p_ in f_.parameters()}
            others = [(table_opt, i) for i in range(len(table_opt.buffers))
                      if i not in (i_prop, i_main) and table_opt.buffers[i][0].data_ptr() not in dead]
            others += [(o_, i) for o_ in optimizers[2:] for i in range(len(o_.buffers))]
            if reducer is not None:
                for o_, i in others:
                    reducer.start(o_.buffers[i][1])
                reducer.wait_all()
            for o_, i in others:
                if o_ is not table_opt:
                    o_.check_buffer(i)
                o_.step_buffer(i, scale)
            if self.amp is not None:
                self.amp.update()

        # ---- graph segments (SegmentedStep, world > 1): the optimizer tail as recorded host actions between graphs
        main_table_cut = False
        if seg is not None:
            scatters_pending = (merged and not before) or any(i_ not in before and not merged for i_ in range(len(chains)))
            if scatters_pending and not shared:
                # cut between the main grid's scatter and the proposal scatters: the main table's exchange (the step's largest
                # collective) runs on the communication stream BESIDE the next segment
                join_all()

                def host_main_table():
                    cur = torch.cuda.current_stream()
                    if self._comm is None:
                        self._comm = torch.cuda.Stream(device=self.dev)
                    amp_flags_sum()
                    self._comm.wait_stream(cur)
                    with torch.cuda.stream(self._comm):
                        main_table_step()

                self._host(host_main_table)
                main_table_cut = True
        if order == "main_first":
            if merged and not before:
                if side[1] is not main:
                    side[1].wait_stream(main)
                with torch.cuda.stream(side[1]):
                    merged_scatter()
            for i_, (lvl, stream) in enumerate(chains):
                if i_ in before or merged:
                    continue
                if stream is not main:
                    stream.wait_stream(main)
                with torch.cuda.stream(stream):
                    chain_scatter(lvl)
        if seg is not None:
            join_all()

            def host_tail():
                cur = torch.cuda.current_stream()
                if main_table_cut:
                    cur.wait_stream(self._comm)
                else:
                    amp_flags_sum()
                    if not shared:
                        main_table_step()
                prop_table_step()
                small_and_other_steps()

            self._host(host_tail, final=True)
            return self.loss
        if optimizers is not None:
            # main table first: its list exchange holds the step's only host read, and issuing it before the
            # proposal table's all-reduce keeps the CPU from parking behind the proposal chains
            if not shared:
                main_table_step()
            if side[0] is not main:  # proposal chains done -> reduce/step the proposal table beside the main chain
                side[0].wait_stream(side[1])
                if shared:
                    side[0].wait_stream(main)
            with torch.cuda.stream(side[0]):
                if reducer is not None:
                    reducer.start(table_opt.buffers[i_prop][1])
                    reducer.wait_all()  # stream-level wait: side[0] continues once RCCL is done
                if amp_ev is not None and side[0] is not main:
                    side[0].wait_event(amp_ev)
                table_opt.step_buffer(i_prop, scale)
        join_all()
        if optimizers is not None:
            small_and_other_steps()
        return self.loss

    def outputs(self) -> Dict[str, Tensor]:
        """Views of the last step's rendered outputs (get_nff_outputs keys)."""
        return {"features": self.features, "depth": self.depth[:, None], "accumulation": self.acc[:, None],
                "weights": self.w[2], "prop_weights_0": self.w[0], "prop_weights_1": self.w[1],
                "prop_depth_0": self.prop_depth[0][:, None], "prop_depth_1": self.prop_depth[1][:, None],
                "final_spacing": self.sp[2], "final_euclid": self.eu[2]}
