"""The training branch behind the rendered features of a mixed batch, on the device: `decode_features`
(models/neuradar.py:410-493) followed by the decoder-dependent terms of `get_metrics_dict` / `get_loss_dict` (:588-704):

  camera rows  -> [patches, C, h, w] -> RGB CNN (training-mode batch norm) -> rgb_mult * MSE(image, rgb)           (:455-461,672-673)
  lidar rows   -> lidar MLP (MFMA kernels) -> quantile-masked depth L1, intensity MSE, ray-drop BCE               (:432-452,612-636)
                  [nr_lidar_depth_quantile / nr_lidar_losses: no boolean-mask indexing, no torch.quantile]
  radar rows   -> sine position embedding of the rendered points, transformer encoder (attention on nr_attention_fwd/bwd),
                  three heads -> radar_output -> Hungarian-matched "nll" | "euclidean" loss                        (:463-491,652-662)
                  [nr_radar_assign / nr_radar_loss: cost matrix, linear sum assignment and loss on the device -- the reference
                   copies the cost matrix to the host and calls scipy every step, radar_utils.py:78]

The segment is a small fixed-shape torch.autograd graph over two leaves (rendered features [B,C], depth [B]) whose heavy
nodes are HIP launches; it contains no host read, so it is captured into the step's hipGraph with everything else.
`FusedTrainStep.set_decoders` runs it between the field forward and nr_render_train and feeds d loss / d features and
d loss / d depth to the render backward (grad_features_extra / grad_depth_extra).
"""
import contextlib
import os
from ctypes import byref
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from . import _lib, ops
from ._lib import NrLidarLosses, check


@dataclass
class DecoderLossSettings:
    """LossSettings of models/neuradar.py:80-115 (the decoder-side entries)."""

    rgb_mult: float = 5.0
    depth_mult: float = 0.01
    intensity_mult: float = 0.1
    quantile_threshold: float = 0.95
    non_return_lidar_distance: float = 150.0
    non_return_loss_mult: float = 0.1
    ray_drop_loss_mult: float = 0.01
    radar_mult: float = 0.02
    radar_loss_type: str = "nll"  # the reference's default (:114); "euclidean" = the deterministic head


class _LidarLosses(torch.autograd.Function):
    """nr_lidar_depth_quantile + nr_lidar_losses: (depth [B], y [n,2]) -> loss scalar; gradients to both."""

    @staticmethod
    def forward(ctx, depth, y, cfg: NrLidarLosses, keep):
        depth, y = ops._f32(depth, "depth"), ops._f32(y, "lidar decoder outputs")
        n = int(cfg.n)
        dev = depth.device
        un = torch.empty(max(n, 1), device=dev, dtype=torch.float32)
        stats = torch.zeros(8, device=dev, dtype=torch.float32)
        slots = torch.zeros(_lib.NR_LOSS_SLOTS, device=dev, dtype=torch.float32)
        g_depth, g_y = torch.zeros_like(depth), torch.empty_like(y)
        lib, p = _lib.lib(), ops._p
        check(lib.nr_lidar_depth_quantile(p(depth), byref(cfg), p(un), p(stats), ops._stream()), "nr_lidar_depth_quantile")
        check(lib.nr_lidar_losses(p(depth), p(y), byref(cfg), p(un), p(stats), p(g_depth), p(g_y), p(slots), ops._stream()),
              "nr_lidar_losses")
        ctx.save_for_backward(g_depth, g_y)
        ctx.mark_non_differentiable(stats)
        return slots.sum(), stats

    @staticmethod
    def backward(ctx, g, _g_stats):
        g_depth, g_y = ctx.saved_tensors
        return g_depth * g, g_y * g, None, None


def lidar_losses(depth: Tensor, y: Tensor, did_return: Tensor, lidar_range: Tensor, target_intensity: Tensor, row0: int, n: int,
                 c: DecoderLossSettings):
    """depth / did_return (uint8) / lidar_range / target_intensity: per ray [B]; the lidar rays are rows [row0, row0 + n);
    y [n,2] = lidar decoder outputs.  Returns (depth_mult * depth_loss + intensity_mult * intensity_loss + ray_drop_mult *
    ray_drop_loss, stats [8]: quantile value at [2], mask count at [5], mask & returned count at [6])."""
    assert did_return.dtype == torch.uint8 and did_return.is_contiguous() and lidar_range.is_contiguous() and target_intensity.is_contiguous()
    cfg = NrLidarLosses()
    cfg.did_return, cfg.range, cfg.target_intensity = did_return.data_ptr(), lidar_range.data_ptr(), target_intensity.data_ptr()
    cfg.row0, cfg.n = int(row0), int(n)
    cfg.non_return_distance, cfg.non_return_loss_mult, cfg.quantile = c.non_return_lidar_distance, c.non_return_loss_mult, c.quantile_threshold
    cfg.depth_mult, cfg.intensity_mult, cfg.ray_drop_mult = c.depth_mult, c.intensity_mult, c.ray_drop_loss_mult
    return _LidarLosses.apply(depth, y, cfg, (did_return, lidar_range, target_intensity))


class _ScaleGrad(torch.autograd.Function):
    """Identity whose backward multiplies the gradient by a device scalar: the loss scale where the gradient enters the RGB
    CNN's 16-bit backward, its inverse where it leaves (no host value involved: the scale may change between graph replays)."""

    @staticmethod
    def forward(ctx, x, factor):
        ctx.save_for_backward(factor)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (factor,) = ctx.saved_tensors
        return g * factor, None


class DecoderLossHead:
    """Decoders + their losses for ONE batch layout (segments of camera / lidar / radar rays at fixed offsets).

    model: a NeuRadarHotPath built with `decoders=True` (rgb_decoder, lidar_decoder, radar_decoder, the three heads,
    appearance_embedding).  layout: {"camera": (row0, n), "lidar": (row0, n), "radar": (row0, n)}; patch = camera patch
    side in rays; n_scans radar scans of n_radar / n_scans rays each.  max_detections: upper bound of a scan's detections."""

    def __init__(self, model, layout: Dict[str, tuple], patch: int, n_scans: int, max_detections: int,
                 settings: Optional[DecoderLossSettings] = None, cnn_autocast: Optional[torch.dtype] = None) -> None:
        self.model, self.layout, self.patch, self.n_scans, self.max_det = model, layout, patch, n_scans, max_detections
        self.c = settings or DecoderLossSettings()
        self.cnn_autocast = cnn_autocast
        if cnn_autocast is None or os.environ.get("NR_CNN_SHADOW", "1") == "0":
            # the RGB decoder's convolutions (and their channels-last BACKWARDS) go through MIOpen on this configuration
            from . import apply_miopen_workaround

            apply_miopen_workaround()
        # fp16 CNN: its backward runs on fp16 activations and gradients, and d loss / d rgb ~ 1e-5 sits below fp16's smallest
        # normal (6e-5) -- the reference trains under GradScaler.  `amp` (step.GradScalerState, set by
        # FusedTrainStep.set_grad_scaler): the camera chain's gradient is multiplied by the dynamic scale where it enters the
        # CNN's backward and divided back where it leaves it (input gradient, parameter gradients), with found-inf detection;
        # without one a static scale of `cnn_loss_scale` is used.  bf16 / fp32 need none.
        self.amp = None
        self.cnn_loss_scale = 8192.0
        self._static_scale = None
        # MIOpen's immediate mode picked its naive (non-tuned) kernels for one 7x7 layer in about every second step of the bench
        # (17 ms instead of 0.3): let it search once per shape and cache the choice
        torch.backends.cudnn.benchmark = True
        dev = next(model.parameters()).device
        n_rad = layout["radar"][1]
        self.radar_ws = (torch.empty(_lib.lib().nr_radar_assign_workspace_bytes(n_scans, n_rad // max(n_scans, 1), max_detections),
                                     device=dev, dtype=torch.uint8) if n_rad else None)
        self.last: Dict[str, Tensor] = {}
        self._shadow = None  # 16-bit copies of the CNN's convolution parameters (see _cnn_shadow)
        # the BasicBlocks' batch norm + ReLU (+ residual) on nr_bn_act_fwd/bwd; their step counters advance in one launch
        self._bn_counters = []
        if os.environ.get("NR_FUSED_BN", "1") != "0" and hasattr(model, "rgb_decoder"):
            for blk in model.rgb_decoder.modules():
                if hasattr(blk, "fused_bn"):
                    blk.fused_bn, blk.fused_bn_counts = True, False
                    self._bn_counters += [b.num_batches_tracked for b in blk.modules()
                                          if isinstance(b, torch.nn.BatchNorm2d) and b.num_batches_tracked is not None]
        self.overlap = os.environ.get("NR_DECODER_STREAMS", "1") != "0"
        self.chainwise = os.environ.get("NR_DECODER_CHAINWISE", "1") != "0"  # backward chain by chain (see _backward_chainwise)
        self._skip = set(filter(None, os.environ.get("NR_DECODER_SKIP", "").split(",")))  # development: time the step without a chain
        self._streams = None
        self._one = None

    def _cnn_shadow(self):
        """16-bit working copies of the RGB CNN's convolution weights / biases, instead of torch.autocast's per-parameter casts:
        autocast launches one cast per parameter in the forward (24) and one per gradient in the backward (24).  Here the 16-bit
        copies are leaves of their own whose storage mirrors the fp32 parameters' flat buffer element for element (same
        offsets and channels-last strides), so ONE copy refreshes all of them before the forward and ONE mixed-precision add
        folds their gradients into the fp32 gradient buffer after the backward (the same values: autocast's backward casts the
        16-bit gradient to fp32 and adds it to a zeroed buffer).  Batch-norm parameters stay fp32, as under autocast.  Needs the
        parameters in one flat buffer (FlatAdam / flatten_parameters); otherwise None -> torch.autocast."""
        if self._shadow is not None or self.cnn_autocast is None or os.environ.get("NR_CNN_SHADOW", "1") == "0":
            return self._shadow or None
        mod = self.model.rgb_decoder
        conv = [(n, p_) for n, p_ in mod.named_parameters() if n.rsplit(".", 1)[0] in
                {k for k, m_ in mod.named_modules() if isinstance(m_, (torch.nn.Conv2d, torch.nn.ConvTranspose2d))}]
        ps = [p_ for _, p_ in conv]
        ok = ps and all(p_.grad is not None and p_.dtype == torch.float32 and p_.grad.dtype == torch.float32
                        and p_.untyped_storage().data_ptr() == ps[0].untyped_storage().data_ptr()
                        and p_.grad.untyped_storage().data_ptr() == ps[0].grad.untyped_storage().data_ptr()
                        and p_.grad.storage_offset() == p_.storage_offset() and p_.grad.stride() == p_.stride() for p_ in ps)
        if not ok:
            self._shadow = False
            return None
        n_all = ps[0].untyped_storage().nbytes() // 4
        flat32 = torch.empty(0, device=ps[0].device, dtype=torch.float32).set_(ps[0].untyped_storage(), 0, (n_all,))
        gflat32 = torch.empty(0, device=ps[0].device, dtype=torch.float32).set_(ps[0].grad.untyped_storage(), 0, (n_all,))
        flat16 = torch.zeros(n_all, device=ps[0].device, dtype=self.cnn_autocast)
        gflat16 = torch.zeros_like(flat16)
        params = {}
        for n, p_ in conv:
            sp = flat16.as_strided(p_.size(), p_.stride(), p_.storage_offset()).detach().requires_grad_(True)
            sp.grad = gflat16.as_strided(p_.size(), p_.stride(), p_.storage_offset())
            params[n] = sp
        self._shadow = dict(flat32=flat32, gflat32=gflat32, flat16=flat16, gflat16=gflat16, params=params)
        # the BasicBlocks' 7 x 7 convolutions (32 -> 32) on nr_conv7_fwd: their weights sit channels-last ([O, kh, kw, I]
        # contiguous) inside flat16, so ONE pack launch per step writes every convolution's two LDS images (NR_CONV7=0: MIOpen)
        self._shadow["conv7"] = None
        if os.environ.get("NR_CONV7", "1") != "0":
            from .decoders import BasicBlock

            blocks = [(k, m_) for k, m_ in mod.named_modules() if isinstance(m_, BasicBlock)]
            w_off, b_off = [], []
            for k, blk in blocks:
                for ci in (0, 3):
                    w_, b_ = params.get(f"{k}.main_branch.{ci}.weight"), params.get(f"{k}.main_branch.{ci}.bias")
                    ok7 = (w_ is not None and tuple(w_.shape) == (32, 32, 7, 7) and w_.stride() == (32 * 49, 1, 7 * 32, 32))
                    if not ok7:
                        w_off = None
                        break
                    w_off.append(w_.storage_offset())
                    b_off.append(b_.storage_offset() if b_ is not None else -1)
                if w_off is None:
                    break
            if w_off:
                images = ops.conv7_pack(flat16, w_off, b_off)
                self._shadow["conv7"] = (images, w_off, b_off)
                for i, (_, blk) in enumerate(blocks):
                    blk.conv7_images = (images[2 * i], images[2 * i + 1])
        # the reference's decoder shape (neuradar.py:225-240): Conv2d 1 x 1 + ReLU | 2 blocks | ConvTranspose2d 3 / 3 | 2 blocks |
        # Conv2d 1 x 1 + Sigmoid -- its three pointwise layers on nr_pw_* (NR_PW=0: the library's)
        self._shadow["pointwise"] = None
        if self._shadow["conv7"] is not None and os.environ.get("NR_PW", "1") != "0" and len(mod) == 9:
            from .decoders import BasicBlock

            c0, r1, t4, c7, s8 = mod[0], mod[1], mod[4], mod[7], mod[8]
            shape_ok = (isinstance(c0, torch.nn.Conv2d) and isinstance(r1, torch.nn.ReLU) and isinstance(t4, torch.nn.ConvTranspose2d)
                        and isinstance(c7, torch.nn.Conv2d) and isinstance(s8, torch.nn.Sigmoid)
                        and all(isinstance(mod[i_], BasicBlock) for i_ in (2, 3, 5, 6))
                        and c0.kernel_size == (1, 1) and c7.kernel_size == (1, 1) and c0.in_channels in (32, 48) and c0.out_channels == 32
                        and t4.kernel_size == (3, 3) and t4.stride == (3, 3) and t4.padding == (0, 0) and t4.in_channels == 32
                        and t4.out_channels == 32 and c7.in_channels == 32 and c7.out_channels <= 32)
            lay_ok = shape_ok and (
                ops.pointwise_ok(torch.empty(1, c0.in_channels, device=flat16.device), params["0.weight"], params.get("0.bias"))
                and ops.pointwise_ok(torch.empty(1, 32, device=flat16.device, dtype=flat16.dtype), params["4.weight"], params.get("4.bias"), True)
                and ops.pointwise_ok(torch.empty(1, 32, device=flat16.device, dtype=flat16.dtype), params["7.weight"], params.get("7.bias")))
            if lay_ok:
                self._shadow["pointwise"] = {i_: {k[len(f"{i_}."):]: v for k, v in params.items() if k.startswith(f"{i_}.")} for i_ in (2, 3, 5, 6)}
        return self._shadow

    def check_radar_status(self) -> None:
        """Host read of the last step's assignment status words (call it OUTSIDE the captured step: after a block of steps, at
        a checkpoint): raises if a scan was left unassigned (more detections than max_detections: its existence probabilities
        would silently be trained towards 0) or the search gave up.  ops.validate_radar_segments does the same check on a
        batch's segments before the step."""
        st = self.last.get("radar_status")
        if st is not None and bool((st != 0).any()):
            raise RuntimeError(f"nr_radar_assign status {st.cpu().tolist()} (2: scan beyond max_detections = {self.max_det} or the "
                               "kernel's limits -- nothing assigned; 1: the search gave up)")

    def _cnn_scale(self):
        """(scale, 1 / scale) device scalars of the camera chain's 16-bit backward, or None (fp32 / bf16 CNN)."""
        if self.cnn_autocast != torch.float16:
            return None
        if self.amp is not None:
            return self.amp.scale, self.amp.inv_scale
        if self._static_scale is None:
            dev = next(self.model.rgb_decoder.parameters()).device
            self._static_scale = (torch.full((1,), self.cnn_loss_scale, device=dev), torch.full((1,), 1.0 / self.cnn_loss_scale, device=dev))
        return self._static_scale

    def _chain_streams(self, device, cur):
        """(lidar stream, radar stream): a stream each beside the CNN on the step's stream -- three concurrent branches, which
        is as many as a captured step runs without losing more than the overlap wins (fused_step.py, `early`; with the step's
        two proposal chains beside the segment the lidar chain is better off on the step's stream: NR_LIDAR_STREAM=main)."""
        if not self.overlap:
            return cur, cur
        if self._streams is None:
            pr = int(os.environ.get("NR_RADAR_PRIORITY", "0"))  # (experiment knob: -1 = the radar chain's stream at high priority)
            self._streams = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device, priority=pr))
        s_lidar = self._streams[0] if os.environ.get("NR_LIDAR_STREAM", "own") == "own" else cur
        return s_lidar, self._streams[1]

    def _lidar_chain(self, xs: Tensor, depth: Tensor, batch: Dict[str, Tensor], out: Dict[str, Tensor]) -> None:
        """Lidar decoder + the quantile-masked lidar losses; xs [n,C]: the lidar rows of the decoders' input, depth [B] (current stream)."""
        r0, n = self.layout["lidar"]
        if n and "lidar" not in self._skip:
            y = self.model.lidar_decoder(xs)
            out["lidar_losses"], self.last["lidar_stats"] = lidar_losses(depth, y, batch["did_return"], batch["range"],
                                                                        batch["target_intensity"], r0, n, self.c)
            self.last["lidar_y"] = y

    def _radar_chain(self, xs: Tensor, depth_rows: Tensor, batch: Dict[str, Tensor], out: Dict[str, Tensor], seed_epoch: Optional[Tensor]) -> None:
        """Radar transformer + heads + the matched radar loss; xs [n,C], depth_rows [n]: the radar rows (current stream)."""
        r0, n = self.layout["radar"]
        if n and "radar" not in self._skip:
            ro = self.model.decode_radar(xs, depth_rows[:, None], batch["directions_spher"][r0:r0 + n], self.n_scans,
                                         seed_epoch=seed_epoch)
            out["radar_loss"], assoc = ops.radar_loss(ro, batch["radar"], batch["radar_seg"], self.max_det, self.c.radar_loss_type,
                                                      mult=self.c.radar_mult, training=True, workspace=self.radar_ws)
            self.last.update(radar_output=ro, assoc=assoc, radar_status=assoc.status)

    def _camera_chain(self, xs: Tensor, batch: Dict[str, Tensor], out: Dict[str, Tensor]) -> None:
        """RGB CNN + the image loss; xs [n,C]: the camera rows, patch after patch (current stream)."""
        m, c = self.model, self.c
        r0, n = self.layout["camera"]
        if n and "cnn" not in self._skip:
            # [P, h, w, C] rows ARE the channels-last layout of [P, C, h, w]; the convolution weights sit channels-last in the
            # optimizer's buffer (fused_step.flatten_parameters), so MIOpen runs NHWC kernels on both without a layout copy.
            # (NHWC input with NCHW weights made it fall back to its naive kernels: 28 ms per step.)
            patches = xs.view(-1, self.patch, self.patch, xs.shape[-1]).permute(0, 3, 1, 2)
            if any(p.dim() == 4 and not p.is_contiguous(memory_format=torch.channels_last) for p in m.rgb_decoder.parameters()):
                patches = patches.contiguous()  # (a model whose parameters were not flattened: packed NCHW for both)
            sh = self._cnn_shadow()
            nhwc = False  # rgb leaves the decoder as [P, 3, h, w] (the library path) or already [P, h, w, 3] (nr_pw_*)
            if sh:
                with torch.no_grad():
                    sh["flat16"].copy_(sh["flat32"])
                    if sh.get("conv7") is not None:  # this step's weights into the convolution kernels' LDS images: one launch
                        ops.conv7_pack(sh["flat16"], sh["conv7"][1], sh["conv7"][2], sh["conv7"][0])
                scale = self._cnn_scale()
                if sh.get("pointwise") is not None and xs.is_contiguous() and xs.dtype == torch.float32:
                    # the whole CNN on hand-written kernels: the pointwise convolutions (head 1 x 1 + ReLU straight from the fp32
                    # rows, transposed 3 x 3, tail 1 x 1 + sigmoid to fp32) on nr_pw_*, the blocks' 7 x 7 on nr_conv7_*
                    prm, subs = sh["params"], sh["pointwise"]
                    n_p, ph = xs.shape[0] // (self.patch * self.patch), self.patch
                    tap = getattr(self, "stage_tap", None)  # tests: the activation after every stage of the decoder ([P, C, H, W])
                    h = ops.pointwise(xs, prm["0.weight"], prm.get("0.bias"), act=1, grad_scale=None if scale is None else scale[1])
                    h = h.view(n_p, ph, ph, h.shape[1]).permute(0, 3, 1, 2)
                    if tap is not None:
                        tap.append(h.detach().float())
                    for i_ in (2, 3):
                        h = torch.func.functional_call(m.rgb_decoder[i_], subs[i_], (h,))
                        if tap is not None:
                            tap.append(h.detach().float())
                    h = ops.conv_transpose3(h, prm["4.weight"], prm.get("4.bias"))
                    if tap is not None:
                        tap.append(h.detach().float())
                    for i_ in (5, 6):
                        h = torch.func.functional_call(m.rgb_decoder[i_], subs[i_], (h,))
                        if tap is not None:
                            tap.append(h.detach().float())
                    rgb = ops.pointwise(h.permute(0, 2, 3, 1).reshape(-1, h.shape[1]), prm["7.weight"], prm.get("7.bias"), act=2, out_f32=True)
                    rgb = rgb.view(n_p, 3 * ph, 3 * ph, rgb.shape[1])
                    if tap is not None:
                        tap.append(rgb.detach().permute(0, 3, 1, 2).float())
                    nhwc = True
                else:
                    if scale is not None:  # d loss / d patches leaves the 16-bit backward scaled: divided back here
                        patches = _ScaleGrad.apply(patches, scale[1])
                    rgb = torch.func.functional_call(m.rgb_decoder, sh["params"], (patches.to(self.cnn_autocast),)).float()
                if scale is not None:  # d loss / d rgb enters it multiplied by the loss scale
                    rgb = _ScaleGrad.apply(rgb, scale[0])
                self._shadow_used = True
            elif self.cnn_autocast is not None:
                with torch.autocast("cuda", dtype=self.cnn_autocast):
                    rgb = m.rgb_decoder(patches)
                rgb = rgb.float()
            else:
                rgb = m.rgb_decoder(patches)
            if self._bn_counters and m.rgb_decoder.training:  # BatchNorm2d.forward's `num_batches_tracked += 1`
                torch._foreach_add_(self._bn_counters, 1)
            if not nhwc:
                rgb = rgb.permute(0, 2, 3, 1)
            out["rgb_loss"] = c.rgb_mult * F.mse_loss(rgb, batch["image"])
            self.last["rgb"] = rgb

    def losses(self, features: Tensor, depth: Tensor, times: Tensor, sensor_idx: Tensor, batch: Dict[str, Tensor],
               seed_epoch: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """features [B,C] rendered features, depth [B]; times [B], sensor_idx [B] int64 (appearance embedding); batch:
        image [P,3h,3w,3], did_return [B] uint8, range [B], target_intensity [B], directions_spher [B,2],
        radar [m,>=3], radar_seg [n_scans+1] int32.  Returns the weighted loss terms (autograd scalars)."""
        m, c = self.model, self.c
        B = features.shape[0]
        if m.config.appearance_dim > 0:  # neuradar.py:518-520: every ray's features are extended by its appearance embedding
            x = ops.appearance_concat(features, m.appearance_embedding.weight, times, sensor_idx, m.config.duration,
                                      m._num_embeds_per_sensor)
        else:
            x = features
        out: Dict[str, Tensor] = {}
        # The three sensors' decoders do not depend on each other: the lidar and radar chains run on side streams beside the
        # CNN (autograd replays every node's backward on the stream of its forward, so the backward overlaps the same way):
        # the segment's critical path is its longest chain instead of their sum.  NR_DECODER_STREAMS=0: one stream.
        cur = torch.cuda.current_stream()
        s_lidar, s_radar = self._chain_streams(features.device, cur)
        for s_ in (s_lidar, s_radar):
            if s_ is not cur:
                s_.wait_stream(cur)
        rows = lambda k: x[self.layout[k][0]:self.layout[k][0] + self.layout[k][1]]  # noqa: E731

        def side_chains():
            with (torch.cuda.stream(s_lidar) if s_lidar is not cur else contextlib.nullcontext()):
                self._lidar_chain(rows("lidar"), depth, batch, out)
            with (torch.cuda.stream(s_radar) if s_radar is not cur else contextlib.nullcontext()):
                self._radar_chain(rows("radar"), depth[self.layout["radar"][0]:self.layout["radar"][0] + self.layout["radar"][1]], batch, out, seed_epoch)

        def camera_chain():
            self._camera_chain(rows("camera"), batch, out)

        if self.overlap:  # the side streams' launches first, the CNN beside them
            side_chains()
            camera_chain()
        else:
            # (one stream: CNN first.  The other order -- MIOpen's convolutions behind the radar chain on ONE captured stream --
            # ended in a GPU memory fault on graph replay on this ROCm build, eagerly it runs; not understood, avoided)
            camera_chain()
            side_chains()
        for s_ in (s_lidar, s_radar):
            if s_ is not cur:
                cur.wait_stream(s_)
        return out

    def backward_into(self, features: Tensor, depth: Tensor, times: Tensor, sensor_idx: Tensor, batch: Dict[str, Tensor],
                      loss_slots: Tensor, seed_epoch: Optional[Tensor] = None):
        """Run the segment on detached leaves of (features, depth), accumulate the parameter gradients into their .grad
        buffers and the loss value into loss_slots[-1]; returns (d loss / d features [B,C], d loss / d depth [B])."""
        if self.overlap and self.chainwise:
            total, terms, g_f, g_d = self._backward_chainwise(features, depth, times, sensor_idx, batch, seed_epoch)
        else:
            f = features.detach().requires_grad_(True)
            d = depth.detach().requires_grad_(True)
            with torch.enable_grad():
                terms = self.losses(f, d, times, sensor_idx, batch, seed_epoch)
                total = sum(terms.values())
            with ops.direct_param_grads():  # the MLP kernels add into the parameters' .grad buffers themselves
                total.backward()
            g_f, g_d = f.grad, (d.grad if d.grad is not None else torch.zeros_like(depth))
        if self._shadow and getattr(self, "_shadow_used", False):
            # the CNN's 16-bit gradients into the fp32 gradient buffer (which holds the batch-norm parameters' gradients of this
            # step -- the optimizers clear gradients every step), unscaled, the 16-bit buffer cleared, found-inf flagged: one launch
            sh, scale = self._shadow, self._cnn_scale()
            flag = None
            if self.amp is not None:
                g_ = self.amp.group_of(next(iter(self.model.rgb_decoder.parameters())))
                flag = self.amp.found(g_) if g_ is not None else None
            check(_lib.lib().nr_unscale_add_16(ops._p(sh["gflat32"]), ops._p(sh["gflat16"]), sh["gflat16"].numel(),
                                               _lib.NR_DTYPES[str(self.cnn_autocast).split(".")[-1]],
                                               ops._p(scale[1]) if scale is not None else None, ops._p(flag), ops._stream()),
                  "nr_unscale_add_16")
            self._shadow_used = False
        loss_slots[-1:].add_(total.detach().reshape(1))  # (the LAST entry: FusedTrainStep keeps it free of the kernels' atomics)
        self.last["terms"] = {k: v.detach() for k, v in terms.items()}
        return g_f, g_d

    def _backward_chainwise(self, features: Tensor, depth: Tensor, times: Tensor, sensor_idx: Tensor, batch: Dict[str, Tensor],
                            seed_epoch: Optional[Tensor]):
        """The segment as three independent forward + backward chains, each entirely on its own stream, cut at the decoders'
        common input x = [features | appearance embedding]: every chain differentiates its loss down to ITS rows of x (a
        detached leaf), the three row blocks are written into one [B, C] buffer and x's own backward (the appearance concat)
        runs once, after the join.  Same numbers as one `backward()` over the sum of the terms (the row blocks are disjoint).
        Why not leave it to autograd: its engine runs the three chains' backward nodes one chain after the other and orders
        each stream switch behind everything launched so far on the producing stream -- measured (rocprofv3 timeline of the
        captured step): CNN backward -> radar backward -> lidar backward strictly in sequence, the
        segment's length their SUM (4.36 ms per step; with the lidar chain alone removed 3.72)."""
        m = self.model
        f = features.detach().requires_grad_(True)
        with torch.enable_grad():
            if m.config.appearance_dim > 0:
                x = ops.appearance_concat(f, m.appearance_embedding.weight, times, sensor_idx, m.config.duration, m._num_embeds_per_sensor)
            else:
                x = f.view_as(f)
        xd, dd = x.detach(), depth.detach()
        B = xd.shape[0]
        segs = {k: self.layout[k] for k in ("camera", "lidar", "radar")}
        live = {k: (r0, n) for k, (r0, n) in segs.items() if n and {"camera": "cnn"}.get(k, k) not in self._skip}
        covered = sum(n for _, n in live.values()) == B
        g_x = torch.empty_like(xd) if covered else torch.zeros_like(xd)
        g_depth = [None, None]
        cur = torch.cuda.current_stream()
        s_lidar, s_radar = self._chain_streams(features.device, cur)
        if self._one is None:
            self._one = torch.ones((), device=features.device)
        terms: Dict[str, Tensor] = {}

        def run(kind, stream):
            if kind not in live:
                return
            r0, n = live[kind]
            if stream is not cur:
                stream.wait_stream(cur)
            with (torch.cuda.stream(stream) if stream is not cur else contextlib.nullcontext()):
                xl = xd[r0:r0 + n].requires_grad_(True)  # the chain's rows of x: a leaf of its own
                out: Dict[str, Tensor] = {}
                with torch.enable_grad():
                    if kind == "lidar":
                        dl = depth.detach().requires_grad_(True)
                        self._lidar_chain(xl, dl, batch, out)
                    elif kind == "radar":
                        dr = dd[r0:r0 + n].requires_grad_(True)  # (the rendered points xyz = origin + depth * direction carry a gradient)
                        self._radar_chain(xl, dr, batch, out, seed_epoch)
                    else:
                        self._camera_chain(xl, batch, out)
                    vals = list(out.values())
                    loss = vals[0] if len(vals) == 1 else sum(vals[1:], vals[0])  # (python's sum starts at 0: one more launch)
                # the MLP / convolution / encoder kernels add into the parameters' .grad buffers themselves.  (The CNN's weight-
                # gradient launches on a stream beside the chain's -- leaves of the backward -- were tried: a fourth concurrent
                # branch, 3.79 -> 4.32 ms per step; profiles/r04_ab_runs.txt item 9)
                with ops.direct_param_grads():
                    loss.backward(self._one)  # (a cached 1.0: no ones_like launch per chain)
                g_x[r0:r0 + n].copy_(xl.grad)
                if kind == "lidar" and dl.grad is not None:
                    g_depth[0] = dl.grad  # [B], zero outside the lidar rows
                if kind == "radar" and dr.grad is not None:
                    g_depth[1] = (r0, n, dr.grad)
                terms.update(out)

        # the side streams' launches first, the CNN beside them on the step's stream
        run("lidar", s_lidar)
        run("radar", s_radar)
        run("camera", cur)
        for s_ in (s_lidar, s_radar):
            if s_ is not cur:
                cur.wait_stream(s_)
        total = sum(v.detach() for v in terms.values())
        with ops.direct_param_grads():
            x.backward(g_x)
        g_d = g_depth[0] if g_depth[0] is not None else torch.zeros_like(dd)
        if g_depth[1] is not None:
            r0, n, g_ = g_depth[1]
            g_d[r0:r0 + n].copy_(g_)  # (disjoint from the lidar rows)
        return total, terms, f.grad, g_d
