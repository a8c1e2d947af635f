"""torch.autograd wrappers over the C ABI (libneuradar_hip.so).

Every function here launches hand-written gfx950 kernels on torch's current stream; tensors must live
on a ROCm device.  torch is plumbing only (memory, streams, autograd bookkeeping).
"""
import os
from ctypes import byref, c_void_p
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import NrField, NrFieldGrads, NrMlp, NrMlpGrads, check

POWER_LAMBDA, POWER_SCALING = -1.0, 0.1  # models/neuradar.py:133-136


def _p(t: Optional[Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def _stream():
    st = torch.cuda.current_stream()
    _lib.ensure_device(st.device_index)  # (nr_init once per device)
    return c_void_p(st.cuda_stream)


def rows_where(t: Tensor, mask: Tensor) -> Tensor:
    """`t[mask]` for a boolean mask over t's first dimension (or over all of t: the result is then 1-D), same values in the same
    order -- through nonzero + index_select, whose backward is index_add_.  The backward of ATen's boolean-mask indexing
    (`indexing_backward_kernel_small_stride`, torch 2.10 / ROCm 7.2) reads behind the end of a tensor: harmless until that tensor is
    the last block of an allocator segment, then a GPU page fault (found with the guard allocator of tests/test_gpu_redzone.py on
    the reference's own `features[is_lidar[..., 0]]`, models/neuradar.py:432-452)."""
    if mask.shape == t.shape and t.dim() != 1:
        return t.reshape(-1).index_select(0, mask.reshape(-1).nonzero().reshape(-1))
    assert mask.dim() == 1 and mask.shape[0] == t.shape[0], "row mask"
    return t.index_select(0, mask.nonzero().reshape(-1))


def _f32(t: Tensor, what: str) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{what} must be a GPU tensor: neuradar_amd has no CPU path (the oracle is test-only)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# ------------------------------------------------------------------------------------------------ hash grid
class _HashEncode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, std, table, scalings, log2_hashmap_size, level_major, sample_major):
        x, table = _f32(x, "x"), _f32(table, "hash_table")
        n, (L, F) = x.shape[0], (scalings.numel(), table.shape[1])
        if level_major:  # [L, n, F] storage, returned as a [n, L*F]-shaped strided view
            buf = torch.empty((L, n, F), device=x.device, dtype=torch.float32)
            sn, sl = F, n * F
        else:
            buf = torch.empty((n, L * F), device=x.device, dtype=torch.float32)
            sn, sl = L * F, F
        check(_lib.lib().nr_hash_encode_fwd(_p(x), _p(std), _p(table), _p(scalings), L, F, log2_hashmap_size,
                                            _p(buf), sn, sl, n, sample_major, _stream()), "nr_hash_encode_fwd")
        ctx.save_for_backward(x, std if std is not None else x.new_empty(0), scalings, table)
        ctx.meta = (table.shape, log2_hashmap_size, level_major, sample_major, std is not None, sn, sl)
        return buf

    @staticmethod
    def backward(ctx, g):
        x, std, scalings, table = ctx.saved_tensors
        shape, log2t, level_major, sample_major, has_std, sn, sl = ctx.meta
        g = g.contiguous()
        gtable = torch.zeros(shape, device=g.device, dtype=torch.float32)
        check(_lib.lib().nr_hash_encode_bwd(_p(x), _p(std) if has_std else None, _p(scalings), scalings.numel(),
                                            shape[1], log2t, _p(g), sn, sl, _p(gtable), x.shape[0], sample_major,
                                            _stream()), "nr_hash_encode_bwd")
        gx = None
        if ctx.needs_input_grad[0]:  # positions that depend on parameters (dynamic-actor trajectories)
            gx = torch.empty_like(x)
            check(_lib.lib().nr_hash_encode_bwd_input(_p(x), _p(std) if has_std else None, _p(table), _p(scalings),
                                                      scalings.numel(), shape[1], log2t, _p(g), sn, sl, _p(gx), x.shape[0],
                                                      _stream()), "nr_hash_encode_bwd_input")
        return gx, None, gtable, None, None, None, None


def hash_encode(x: Tensor, table: Tensor, scalings: Tensor, log2_hashmap_size: int, std: Optional[Tensor] = None,
                level_major: bool = False, sample_major: int = 0) -> Tensor:
    """x [n,3] in [0,1] -> [n, L*F] (or the level-major buffer [L, n, F] when `level_major`)."""
    return _HashEncode.apply(x, std, table, scalings, log2_hashmap_size, level_major, sample_major)


def hash_encode_bwd_binned(x: Tensor, std: Optional[Tensor], scalings: Tensor, log2_hashmap_size: int, grad_out: Tensor,
                           strides: Tuple[int, int], features_per_level: int, grad_table: Tensor,
                           workspace: Optional[Tensor] = None, sum_bits: int = 64) -> Tensor:
    """grad_table += scatter of grad_out (element (i,l,f) at i*strides[0] + l*strides[1] + f) through the two-pass binned
    kernels (nr_hash_encode_bwd_binned: for incoherent rows).  Returns the workspace (reusable for the same sizes).
    sum_bits: 64 (exact to fp32's resolution) or 32 (addends rounded to 2^-22..2^-21 of their tile's largest: the companion
    of 16-bit MLP operands, see include/neuradar_hip.h)."""
    n, L = x.shape[0], scalings.numel()
    need = _lib.lib().nr_hash_encode_bwd_binned_workspace_bytes(L, features_per_level, log2_hashmap_size, n)
    if need < 0:
        raise RuntimeError("nr_hash_encode_bwd_binned: a level of this table has more than 32 slices of 128 KB")
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=x.device, dtype=torch.uint8)
    check(_lib.lib().nr_hash_encode_bwd_binned_lp(_p(x), _p(std), _p(scalings), L, features_per_level, log2_hashmap_size,
                                                  _p(grad_out), strides[0], strides[1], _p(grad_table), n, _p(workspace),
                                                  int(sum_bits), _stream()), "nr_hash_encode_bwd_binned")
    return workspace


def _sm_rays(flag, n_rays: int) -> int:
    """Row-order argument of the ABI: number of leading rays stored sample-major (True = all)."""
    return n_rays if flag is True else int(flag)


def contract_gaussians(origins: Tensor, directions: Tensor, pixel_area: Tensor, euclid: Tensor, scale: float,
                       sample_major_rows=False) -> Tuple[Tensor, Tensor]:
    """Frustum samples -> contracted (x01 [B*S,3], std01 [B*S]).  No gradient (static scene).
    sample_major_rows: True = every ray, or the number of leading rays, stored at row s*sm+b instead of
    b*S+s (include/neuradar_hip.h, nr_contract_gaussians)."""
    B, S = euclid.shape[0], euclid.shape[1] - 1
    x01 = torch.empty((B * S, 3), device=euclid.device, dtype=torch.float32)
    std01 = torch.empty((B * S,), device=euclid.device, dtype=torch.float32)
    check(_lib.lib().nr_contract_gaussians(_p(_f32(origins, "origins")), _p(_f32(directions, "directions")),
                                           _p(_f32(pixel_area, "pixel_area")), _p(_f32(euclid, "euclid")), B, S,
                                           float(scale), _sm_rays(sample_major_rows, B), _p(x01), _p(std01), _stream()),
          "nr_contract_gaussians")
    return x01, std01


# ------------------------------------------------------------------------------------------------ MLPs
def _mlp_struct(weights: Sequence[Tensor], biases: Sequence[Tensor]) -> NrMlp:
    m = NrMlp()
    m.num_layers = len(weights)
    m.in_dim, m.out_dim = weights[0].shape[1], weights[-1].shape[0]
    m.width = weights[0].shape[0] if len(weights) > 1 else weights[0].shape[0]
    for i, (w, b) in enumerate(zip(weights, biases)):
        m.weight[i], m.bias[i] = w.data_ptr(), b.data_ptr()
    return m


def _mlp_grads_struct(gw: Sequence[Tensor], gb: Sequence[Tensor]) -> NrMlpGrads:
    g = NrMlpGrads()
    for i, (w, b) in enumerate(zip(gw, gb)):
        g.weight[i], g.bias[i] = w.data_ptr(), b.data_ptr()
    return g


_DIRECT_PARAM_GRADS = False  # (a plain global, not thread-local: autograd runs GPU nodes on its own device threads)


class direct_param_grads:
    """Inside this context, `mlp`'s backward adds the parameter gradients straight into the parameters' `.grad` buffers (the
    ABI's grads are "+=") and hands autograd None for them: no zero-filled temporaries, no AccumulateGrad adds -- 13 launches
    less per MLP backward.  For callers that want exactly `loss.backward()`'s accumulation into preallocated `.grad`s (the
    decoder-loss segment of the fused step); `torch.autograd.grad(...)` over those parameters would see None."""

    def __enter__(self):
        global _DIRECT_PARAM_GRADS
        self._prev, _DIRECT_PARAM_GRADS = _DIRECT_PARAM_GRADS, True

    def __exit__(self, *exc):
        global _DIRECT_PARAM_GRADS
        _DIRECT_PARAM_GRADS = self._prev


class _Mlp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_layers, *params):
        x = _f32(x, "x")
        ws = [_f32(p, "weight") for p in params[:n_layers]]
        bs = [_f32(p, "bias") for p in params[n_layers:]]
        m = _mlp_struct(ws, bs)
        y = torch.empty((x.shape[0], m.out_dim), device=x.device, dtype=torch.float32)
        check(_lib.lib().nr_mlp_fwd(byref(m), _p(x), x.shape[0], _p(y), _stream()), "nr_mlp_fwd")
        ctx.save_for_backward(x, *ws, *bs)
        ctx.n_layers = n_layers
        ctx.param_refs = params  # the caller's tensors (their .grad, for direct_param_grads)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, *params = ctx.saved_tensors
        nl = ctx.n_layers
        ws, bs = params[:nl], params[nl:]
        m = _mlp_struct(ws, bs)
        refs = ctx.param_refs
        direct = _DIRECT_PARAM_GRADS and all(
            isinstance(p, Tensor) and p.is_leaf and p.requires_grad and p.grad is not None and p.grad.dtype == torch.float32
            and p.grad.is_contiguous() and p.dtype == torch.float32 and p.is_contiguous() for p in refs)
        if direct:
            gws, gbs = [p.grad for p in refs[:nl]], [p.grad for p in refs[nl:]]
        else:
            gws, gbs = [torch.zeros_like(w) for w in ws], [torch.zeros_like(b) for b in bs]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        g = _mlp_grads_struct(gws, gbs)
        check(_lib.lib().nr_mlp_bwd(byref(m), _p(x), _p(gy.contiguous()), x.shape[0], _p(gx), byref(g), _stream()),
              "nr_mlp_bwd")
        if direct:
            return (gx, None, *([None] * len(refs)))
        return (gx, None, *gws, *gbs)


def mlp(x: Tensor, weights: Sequence[Tensor], biases: Sequence[Tensor]) -> Tensor:
    return _Mlp.apply(x, len(weights), *weights, *biases)


class _Attention(torch.autograd.Function):
    """nr_attention_fwd / nr_attention_bwd: single-head softmax attention per scan, q / k / v [N, n, D]."""

    @staticmethod
    def forward(ctx, q, k, v, dropout_p, seed, keep_mask, seed_epoch=None):
        q, k, v = _f32(q, "q"), _f32(k, "k"), _f32(v, "v")
        N, n, D = q.shape
        out, lse = torch.empty_like(q), torch.empty((N, n), device=q.device, dtype=torch.float32)
        ws = torch.empty(max(int(_lib.lib().nr_attention_workspace_floats(N, n, D)), 1), device=q.device, dtype=torch.float32)
        check(_lib.lib().nr_attention_fwd(_p(q), _p(k), _p(v), N, n, D, float(dropout_p), int(seed) & 0xFFFFFFFF, _p(seed_epoch), _p(keep_mask),
                                          _p(out), _p(lse), _p(ws), _stream()), "nr_attention_fwd")
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.cfg, ctx.mask, ctx.ws, ctx.epoch = (float(dropout_p), int(seed) & 0xFFFFFFFF), keep_mask, ws, seed_epoch
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, out, lse = ctx.saved_tensors
        N, n, D = q.shape
        gq, gk, gv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        check(_lib.lib().nr_attention_bwd(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(g.contiguous().float()), N, n, D, ctx.cfg[0], ctx.cfg[1],
                                          _p(ctx.epoch), _p(ctx.mask), _p(gq), _p(gk), _p(gv), _p(ctx.ws), _stream()), "nr_attention_bwd")
        return gq, gk, gv, None, None, None, None


def attention(q: Tensor, k: Tensor, v: Tensor, dropout_p: float = 0.0, seed: int = 0, keep_mask: Optional[Tensor] = None,
              seed_epoch: Optional[Tensor] = None) -> Tensor:
    """softmax(q k^T / sqrt(D)) (with dropout on the probabilities) v, one head, per scan: q, k, v [N, n, D], D in {32, 48, 64}
    (what nn.MultiheadAttention(d_model, nhead=1) computes between its projections).  keep_mask [N, n, n] of 0 / 1 replaces
    the kernel's own dropout decisions (tests).  seed_epoch: device-resident float step counter folded into the seed by the
    kernels (a captured graph then draws new masks on every replay)."""
    return _Attention.apply(q.contiguous(), k.contiguous(), v.contiguous(), dropout_p, seed,
                            None if keep_mask is None else keep_mask.contiguous().float(), seed_epoch)


class _EncoderLayer(torch.autograd.Function):
    """nr_encoder_pre_fwd -> nr_attention_fwd -> nr_encoder_post_fwd and their backwards: the radar decoder's pre-norm encoder
    layer + the encoder's final LayerNorm (detr/models/transformer.py:176-189, :66-68), six launches each way."""

    @staticmethod
    def _struct(params, C, FF, eps, p_drop, seed, seed_epoch):
        enc = _lib.NrEncoder()
        for name, t in zip(_lib._ENC_PARAMS, params):
            setattr(enc, name, t.data_ptr())
        enc.d_model, enc.dim_feedforward, enc.eps, enc.p_drop, enc.seed = C, FF, float(eps), float(p_drop), int(seed) & 0xFFFFFFFF
        enc.seed_epoch = seed_epoch.data_ptr() if seed_epoch is not None else None
        return enc

    @staticmethod
    def forward(ctx, x, pos, cfg, *params):
        p_drop, seed, seed_epoch, eps = cfg
        x, pos = _f32(x, "x"), _f32(pos, "pos")
        N, n, C = x.shape
        FF = params[4].shape[0]
        lib = _lib.lib()
        enc = _EncoderLayer._struct(params, C, FF, eps, p_drop, seed, seed_epoch)
        q, k, v, att, out = (torch.empty_like(x) for _ in range(5))
        lse = torch.empty((N, n), device=x.device, dtype=torch.float32)
        ws = torch.empty(max(int(lib.nr_attention_workspace_floats(N, n, C)), 1), device=x.device, dtype=torch.float32)
        check(lib.nr_encoder_pre_fwd(byref(enc), _p(x), _p(pos), N * n, _p(q), _p(k), _p(v), _stream()), "nr_encoder_pre_fwd")
        check(lib.nr_attention_fwd(_p(q), _p(k), _p(v), N, n, C, float(p_drop), int(seed) & 0xFFFFFFFF, _p(seed_epoch), None, _p(att),
                                   _p(lse), _p(ws), _stream()), "nr_attention_fwd")
        check(lib.nr_encoder_post_fwd(byref(enc), _p(x), _p(att), N * n, _p(out), _stream()), "nr_encoder_post_fwd")
        ctx.save_for_backward(x, pos, q, k, v, att, lse, *params)
        ctx.cfg, ctx.ws, ctx.param_refs = (p_drop, seed, seed_epoch, eps, FF), ws, params
        return out

    @staticmethod
    def backward(ctx, g):
        x, pos, q, k, v, att, lse, *params = ctx.saved_tensors
        p_drop, seed, seed_epoch, eps, FF = ctx.cfg
        N, n, C = x.shape
        lib = _lib.lib()
        enc = _EncoderLayer._struct(params, C, FF, eps, p_drop, seed, seed_epoch)
        refs = ctx.param_refs
        direct = _DIRECT_PARAM_GRADS and all(p.is_leaf and p.requires_grad and p.grad is not None and p.grad.dtype == torch.float32
                                             and p.grad.is_contiguous() for p in refs)
        # (direct: added straight into the parameters' .grad buffers; otherwise zero-filled temporaries handed to autograd)
        gp = [p.grad for p in refs] if direct else [torch.zeros_like(p) for p in params]
        grads = _lib.NrEncoderGrads()
        for name, t in zip(_lib._ENC_PARAMS, gp):
            setattr(grads, name, t.data_ptr())
        g = _f32(g.contiguous(), "grad_out")
        g_att, g_x1, g_x = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        gq, gk, gv = torch.zeros((3,) + tuple(q.shape), device=q.device, dtype=q.dtype).unbind(0)  # (one fill: the attention backward adds)
        check(lib.nr_encoder_post_bwd(byref(enc), _p(x), _p(att), _p(g), N * n, _p(g_att), _p(g_x1), byref(grads), _stream()),
              "nr_encoder_post_bwd")
        check(lib.nr_attention_bwd(_p(q), _p(k), _p(v), _p(att), _p(lse), _p(g_att), N, n, C, float(p_drop), int(seed) & 0xFFFFFFFF,
                                   _p(seed_epoch), None, _p(gq), _p(gk), _p(gv), _p(ctx.ws), _stream()), "nr_attention_bwd")
        check(lib.nr_encoder_pre_bwd(byref(enc), _p(x), _p(pos), _p(gq), _p(gk), _p(gv), _p(g_x1), N * n, _p(g_x), byref(grads),
                                     _stream()), "nr_encoder_pre_bwd")
        return (g_x, None, None) + tuple(None if direct else t for t in gp)


ENCODER_WIDTHS = ((32, 64), (48, 64), (64, 64))  # (d_model, dim_feedforward) the encoder kernels are built for


def encoder_layer(x: Tensor, pos: Tensor, layer, final_norm, p_drop: float = 0.0, seed: int = 0, seed_epoch: Optional[Tensor] = None) -> Tensor:
    """final_norm(forward_pre(layer)(x, pos)) for x, pos [N, n, C] (N scans of n tokens; attention inside a scan): `layer` with the
    reference's TransformerEncoderLayer parameters (self_attn.in_proj_*, self_attn.out_proj, linear1, linear2, norm1, norm2),
    nhead = 1.  pos is a constant.  p_drop: the three dropouts and the attention's (0 in eval mode)."""
    a = layer.self_attn
    params = (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, layer.linear1.weight, layer.linear1.bias,
              layer.linear2.weight, layer.linear2.bias, layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias,
              final_norm.weight, final_norm.bias)
    assert (x.shape[-1], layer.linear1.out_features) in ENCODER_WIDTHS and layer.norm1.eps == layer.norm2.eps == final_norm.eps
    return _EncoderLayer.apply(x.contiguous(), pos.contiguous(), (float(p_drop), int(seed), seed_epoch, float(layer.norm1.eps)), *params)


class _Field(torch.autograd.Function):
    """nr_field_fwd/bwd: feats (+ strides) -> feature, sdf, alpha."""

    @staticmethod
    def forward(ctx, feats, strides, feat_f, directions, n_samples, rows_sm, n, beta, precision, sample_dirs, n_geo, *params):
        geo_w, geo_b = params[:n_geo], params[n_geo:2 * n_geo]
        rest = params[2 * n_geo:]
        n_feat = len(rest) // 2
        feat_w, feat_b = rest[:n_feat], rest[n_feat:]
        fld = NrField()
        fld.geo, fld.feat, fld.beta = _mlp_struct(geo_w, geo_b), _mlp_struct(feat_w, feat_b), beta.data_ptr()
        dtype, grad_scale = precision
        fld.dtype, fld.grad_scale = _lib.NR_DTYPES[dtype], float(grad_scale)
        if sample_dirs is not None:
            fld.sample_dirs = sample_dirs.data_ptr()
        image = None
        if fld.dtype != 0:  # bf16 / fp16 operands: the kernels read the weights from the converted image only
            image = torch.empty(_lib.lib().nr_field_image_floats(byref(fld)), device=feats.device, dtype=torch.float32)
            check(_lib.lib().nr_field_pack(byref(fld), _p(image), _stream()), "nr_field_pack")
            fld.packed = image.data_ptr()
        C = feat_w[-1].shape[0]
        feature = torch.empty((n, C), device=feats.device, dtype=torch.float32)
        sdf = torch.empty((n,), device=feats.device, dtype=torch.float32)
        alpha = torch.empty((n,), device=feats.device, dtype=torch.float32)
        sn, sl = strides
        # a backward will follow: let the forward leave its activations for it (nr_field_t.stash)
        stash = None
        if n > 0 and fld.dtype == 0 and (feats.requires_grad or any(p.requires_grad for p in params)):
            stash = torch.empty(_lib.lib().nr_field_stash_floats(byref(fld), n), device=feats.device, dtype=torch.float32)
            fld.stash = stash.data_ptr()
        check(_lib.lib().nr_field_fwd(byref(fld), _p(feats), sn, sl, feat_f, _p(directions), n_samples, rows_sm, n,
                                      _p(feature), _p(sdf), _p(alpha), _stream()), "nr_field_fwd")
        ctx.stash, ctx.image = stash, image
        ctx.save_for_backward(feats, directions, beta, *params)
        ctx.sample_dirs = sample_dirs
        ctx.meta = (strides, feat_f, n_samples, rows_sm, n, n_geo, n_feat, fld.dtype, fld.grad_scale)
        return feature, sdf, alpha

    @staticmethod
    def backward(ctx, g_feature, g_sdf, g_alpha):
        feats, directions, beta, *params = ctx.saved_tensors
        (sn, sl), feat_f, n_samples, rows_sm, n, n_geo, n_feat, dtype, grad_scale = ctx.meta
        geo_w, geo_b = params[:n_geo], params[n_geo:2 * n_geo]
        feat_w, feat_b = params[2 * n_geo:2 * n_geo + n_feat], params[2 * n_geo + n_feat:]
        fld = NrField()
        fld.geo, fld.feat, fld.beta = _mlp_struct(geo_w, geo_b), _mlp_struct(feat_w, feat_b), beta.data_ptr()
        fld.dtype, fld.grad_scale = dtype, grad_scale
        if ctx.stash is not None:
            fld.stash = ctx.stash.data_ptr()
        if ctx.image is not None:
            fld.packed = ctx.image.data_ptr()
        if ctx.sample_dirs is not None:
            fld.sample_dirs = ctx.sample_dirs.data_ptr()
        grads = [torch.zeros_like(p) for p in params]
        g_beta = torch.zeros_like(beta)
        gs = NrFieldGrads()
        gs.geo = _mlp_grads_struct(grads[:n_geo], grads[n_geo:2 * n_geo])
        gs.feat = _mlp_grads_struct(grads[2 * n_geo:2 * n_geo + n_feat], grads[2 * n_geo + n_feat:])
        gs.beta = g_beta.data_ptr()
        g_feats = torch.empty_like(feats)
        g_feature = torch.zeros((n, feat_w[-1].shape[0]), device=feats.device) if g_feature is None else g_feature.contiguous()
        g_alpha = torch.zeros((n,), device=feats.device) if g_alpha is None else g_alpha.contiguous()
        g_sdf = None if g_sdf is None else g_sdf.contiguous()
        ws = torch.empty(_lib.lib().nr_field_bwd_workspace_floats(byref(fld), n), device=feats.device, dtype=torch.float32)
        check(_lib.lib().nr_field_bwd(byref(fld), _p(feats), sn, sl, feat_f, _p(directions), n_samples, rows_sm, n,
                                      _p(g_feature), _p(g_alpha), _p(g_sdf), _p(g_feats), byref(gs), _p(ws), _stream()),
              "nr_field_bwd")
        return (g_feats, None, None, None, None, None, None, g_beta, None, None, None, *grads)


def field_mlp(feats: Tensor, strides: Tuple[int, int], feat_f: int, directions: Tensor, n_samples: int, n: int,
              geo: Tuple[List[Tensor], List[Tensor]], feat: Tuple[List[Tensor], List[Tensor]], beta: Tensor,
              rows_sample_major: bool = False, dtype: str = "float32", grad_scale: float = 1.0,
              sample_dirs: Optional[Tensor] = None):
    """NeuRADField after the grid (neurad_field.py:137-148).  feats is the raw buffer written by
    hash_encode; `strides` = (stride_n, stride_l) in floats.  Returns feature [n,C], sdf [n], alpha [n]
    (always in [B,S] order; rows_sample_major: feats rows are s*B+b).
    dtype: "float32" (fp32 MFMA, the parity path), "bfloat16" or "float16": 16-bit MFMA operands with fp32 accumulation
    (nr_field_t.dtype); grad_scale: static loss scale of the 16-bit backward (fp16 needs one).
    sample_dirs [n,3]: view directions per sample (ray-major index), replacing `directions` (dynamic actors)."""
    sm = _sm_rays(rows_sample_major, n // n_samples if n_samples else 0)
    if sample_dirs is not None:
        sample_dirs = _f32(sample_dirs, "sample_dirs")
    return _Field.apply(feats, strides, feat_f, _f32(directions, "directions"), n_samples, sm, n, beta, (dtype, grad_scale), sample_dirs,
                        len(geo[0]),
                        *geo[0], *geo[1], *feat[0], *feat[1])


def sh4(dirs01: Tensor) -> Tensor:
    d = _f32(dirs01, "directions").reshape(-1, 3)
    out = torch.empty((d.shape[0], 16), device=d.device, dtype=torch.float32)
    check(_lib.lib().nr_sh4_fwd(_p(d), d.shape[0], _p(out), _stream()), "nr_sh4_fwd")
    return out.view(*dirs01.shape[:-1], 16)


# ------------------------------------------------------------------------------------------------ proposal head
class _PropDensity(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, strides, feat_f, w, n, n_samples, rows_sm):
        sn, sl = strides
        density = torch.empty((n,), device=feats.device, dtype=torch.float32)
        wf = _f32(w, "decoder weight").reshape(-1)
        check(_lib.lib().nr_prop_density_fwd(_p(feats), sn, sl, feat_f, _p(wf), wf.numel(), n, n_samples, rows_sm,
                                             _p(density), _stream()), "nr_prop_density_fwd")
        ctx.save_for_backward(feats, wf, density)
        ctx.meta = (strides, feat_f, n, w.shape, n_samples, rows_sm)
        return density

    @staticmethod
    def backward(ctx, g):
        feats, wf, density = ctx.saved_tensors
        (sn, sl), feat_f, n, wshape, n_samples, rows_sm = ctx.meta
        g_feats = torch.empty_like(feats)
        g_w = torch.zeros_like(wf)
        check(_lib.lib().nr_prop_density_bwd(_p(feats), sn, sl, feat_f, _p(wf), wf.numel(), n, n_samples, rows_sm,
                                             _p(density), _p(g.contiguous()), _p(g_feats), _p(g_w), _stream()),
              "nr_prop_density_bwd")
        return g_feats, None, None, g_w.view(wshape), None, None, None


def prop_density(feats: Tensor, strides: Tuple[int, int], feat_f: int, w: Tensor, n: int, n_samples: int = 0,
                 rows_sample_major: bool = False) -> Tensor:
    """density [n] in [B,S] order; rows_sample_major: feats rows are s*B+b (needs n_samples)."""
    return _PropDensity.apply(feats, strides, feat_f, w, n, n_samples, _sm_rays(rows_sample_major, n // n_samples if n_samples else 0))


# ------------------------------------------------------------------------------------------------ sampling
def power_bins(nears: Tensor, fars: Tensor, n_samples: int, t_rand: Optional[Tensor] = None,
               lam: float = POWER_LAMBDA, scaling: float = POWER_SCALING) -> Tuple[Tensor, Tensor]:
    nears, fars = _f32(nears, "nears").reshape(-1), _f32(fars, "fars").reshape(-1)
    B = nears.shape[0]
    spacing = torch.empty((B, n_samples + 1), device=nears.device, dtype=torch.float32)
    euclid = torch.empty_like(spacing)
    if t_rand is not None:
        t_rand = _f32(t_rand, "t_rand")
        assert t_rand.shape == (B, n_samples + 1)
    check(_lib.lib().nr_power_bins(_p(nears), _p(fars), _p(t_rand), B, n_samples, lam, scaling, _p(spacing),
                                   _p(euclid), _stream()), "nr_power_bins")
    return spacing, euclid


class _Weights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, density, euclid):
        density, euclid = _f32(density, "density"), _f32(euclid, "euclid")
        B, S = density.shape
        w = torch.empty_like(density)
        check(_lib.lib().nr_weights_from_density_fwd(_p(density), _p(euclid), B, S, _p(w), _stream()),
              "nr_weights_from_density_fwd")
        ctx.save_for_backward(density, euclid)
        return w

    @staticmethod
    def backward(ctx, gw):
        density, euclid = ctx.saved_tensors
        B, S = density.shape
        gd = torch.empty_like(density)
        check(_lib.lib().nr_weights_from_density_bwd(_p(density), _p(euclid), _p(gw.contiguous()), B, S, _p(gd),
                                                     _stream()), "nr_weights_from_density_bwd")
        return gd, None


def weights_from_density(density: Tensor, euclid: Tensor) -> Tensor:
    """RaySamples.get_weights: density [B,S], euclid edges [B,S+1] -> weights [B,S]."""
    return _Weights.apply(density, euclid)


def pdf_resample(weights: Tensor, spacing_in: Tensor, nears: Tensor, fars: Tensor, n_out: int,
                 jitter: Optional[Tensor] = None, lam: float = POWER_LAMBDA, scaling: float = POWER_SCALING,
                 sky_distance: float = 0.0) -> Tuple[Tensor, Tensor]:
    weights = _f32(weights.detach(), "weights")
    B, S = weights.shape
    sp = torch.empty((B, n_out + 1), device=weights.device, dtype=torch.float32)
    eu = torch.empty_like(sp)
    if jitter is not None:
        jitter = _f32(jitter, "jitter").reshape(-1)
    check(_lib.lib().nr_pdf_resample(_p(weights), _p(_f32(spacing_in, "spacing")), _p(jitter),
                                     _p(_f32(nears, "nears").reshape(-1)), _p(_f32(fars, "fars").reshape(-1)), B, S,
                                     n_out, lam, scaling, sky_distance, _p(sp), _p(eu), _stream()), "nr_pdf_resample")
    return sp, eu


# ------------------------------------------------------------------------------------------------ compositing
class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, feature, euclid):
        alpha, feature, euclid = _f32(alpha, "alpha"), _f32(feature, "feature"), _f32(euclid, "euclid")
        B, S = alpha.shape
        C = feature.shape[-1]
        w = torch.empty_like(alpha)
        acc = torch.empty((B,), device=alpha.device, dtype=torch.float32)
        feats = torch.empty((B, C), device=alpha.device, dtype=torch.float32)
        depth = torch.empty((B,), device=alpha.device, dtype=torch.float32)
        check(_lib.lib().nr_composite_fwd(_p(alpha), _p(feature), _p(euclid), B, S, C, _p(w), _p(acc), _p(feats),
                                          _p(depth), _stream()), "nr_composite_fwd")
        ctx.save_for_backward(alpha, feature, euclid, w)
        return w, acc, feats, depth

    @staticmethod
    def backward(ctx, g_w, g_acc, g_feats, g_depth):
        alpha, feature, euclid, w = ctx.saved_tensors
        B, S = alpha.shape
        C = feature.shape[-1]
        g_alpha, g_feature = torch.empty_like(alpha), torch.empty_like(feature)
        c = lambda t: None if t is None else t.contiguous()  # noqa: E731
        check(_lib.lib().nr_composite_bwd(_p(alpha), _p(feature), _p(euclid), _p(w), _p(c(g_feats)), _p(c(g_depth)),
                                          _p(c(g_acc)), _p(c(g_w)), B, S, C, _p(g_alpha), _p(g_feature), _stream()),
              "nr_composite_bwd")
        return g_alpha, g_feature, None


def composite(alpha: Tensor, feature: Tensor, euclid: Tensor):
    """alpha [B,S], feature [B,S,C], euclid [B,S+1] -> (weights [B,S] incl. sky fix-up,
    accumulation [B], features [B,C], depth [B]).  models/neuradar.py:504-517."""
    return _Composite.apply(alpha, feature, euclid)


class _RenderWeights(torch.autograd.Function):
    """nr_render_weights_fwd/bwd: nerfacc's batched render_weight_from_alpha / render_weight_from_density."""

    @staticmethod
    def forward(ctx, x, t_starts, t_ends):
        x = _f32(x, "alphas / sigmas")
        shape = x.shape
        S = shape[-1] if x.dim() else 1
        x2 = x.reshape(-1, S)
        density = t_starts is not None
        ts = _f32(t_starts, "t_starts").reshape(-1, S) if density else None
        te = _f32(t_ends, "t_ends").reshape(-1, S) if density else None
        w, T = torch.empty_like(x2), torch.empty_like(x2)
        a = torch.empty_like(x2) if density else None
        check(_lib.lib().nr_render_weights_fwd(_p(x2), _p(ts), _p(te), x2.shape[0], S, _p(w), _p(T), _p(a), _stream()),
              "nr_render_weights_fwd")
        ctx.save_for_backward(x2, ts, te, T)
        ctx.shape = shape
        if density:
            return w.view(shape), T.view(shape), a.view(shape)
        return w.view(shape), T.view(shape)

    @staticmethod
    def backward(ctx, g_w, g_T, g_a=None):
        x2, ts, te, T = ctx.saved_tensors
        S = x2.shape[1]
        c = lambda t: None if t is None else t.contiguous().view(-1, S)  # noqa: E731
        g_in = torch.empty_like(x2)
        check(_lib.lib().nr_render_weights_bwd(_p(x2), _p(ts), _p(te), _p(T), _p(c(g_w)), _p(c(g_T)), _p(c(g_a)), x2.shape[0], S,
                                               _p(g_in), _stream()), "nr_render_weights_bwd")
        return g_in.view(ctx.shape), None, None


def render_weights(alphas: Tensor) -> Tuple[Tensor, Tensor]:
    """alphas [..., S] -> (weights, transmittance), T_i = prod_{j<i}(1 - alpha_j) (models/neuradar.py:1016)."""
    return _RenderWeights.apply(alphas, None, None)


def render_weights_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """(weights, transmittance, alphas) with alpha = 1 - exp(-sigma (t_end - t_start)) (models/neuradar.py:1018-1022)."""
    return _RenderWeights.apply(sigmas, t_starts, t_ends)


class _Accumulate(torch.autograd.Function):
    """nr_accumulate_fwd/bwd: nerfacc's batched accumulate_along_rays."""

    @staticmethod
    def forward(ctx, weights, values):
        w = _f32(weights, "weights")
        B, S = w.shape
        v = _f32(values, "values") if values is not None else None
        C = v.shape[-1] if v is not None else 1
        out = torch.empty((B, C), device=w.device, dtype=torch.float32)
        check(_lib.lib().nr_accumulate_fwd(_p(w), _p(v), B, S, C, _p(out), _stream()), "nr_accumulate_fwd")
        ctx.save_for_backward(w, v)
        return out

    @staticmethod
    def backward(ctx, g):
        w, v = ctx.saved_tensors
        B, S = w.shape
        C = v.shape[-1] if v is not None else 1
        g_w = torch.zeros_like(w) if ctx.needs_input_grad[0] else None
        g_v = torch.zeros_like(v) if (v is not None and ctx.needs_input_grad[1]) else None
        if g_w is not None or g_v is not None:
            check(_lib.lib().nr_accumulate_bwd(_p(w), _p(v), _p(g.contiguous()), B, S, C, _p(g_w), _p(g_v), _stream()),
                  "nr_accumulate_bwd")
        return g_w, g_v


def accumulate_along_rays(weights: Tensor, values: Optional[Tensor] = None) -> Tensor:
    """weights [B,S], values [B,S,C] or None -> [B,C] ([B,1])."""
    return _Accumulate.apply(weights, values)


class _DepthFromWeights(torch.autograd.Function):
    """nr_depth_from_weights; backward through nr_accumulate_bwd with values = the samples' midpoints (d depth / d w_s = mid_s:
    the lidar depth losses on the proposal levels reach the proposal weights this way, neuradar.py:641-648)."""

    @staticmethod
    def forward(ctx, weights, euclid):
        weights, euclid = _f32(weights, "weights"), _f32(euclid, "euclid")
        B, S = weights.shape
        depth = torch.empty((B,), device=weights.device, dtype=torch.float32)
        check(_lib.lib().nr_depth_from_weights(_p(weights), _p(euclid), B, S, _p(depth), _stream()), "nr_depth_from_weights")
        ctx.save_for_backward(weights, euclid)
        return depth

    @staticmethod
    def backward(ctx, g):
        weights, euclid = ctx.saved_tensors
        B, S = weights.shape
        mid = ((euclid[:, :-1] + euclid[:, 1:]) / 2.0).contiguous()
        g_w = torch.empty_like(weights)
        check(_lib.lib().nr_accumulate_bwd(_p(weights), _p(mid), _p(g.contiguous()), B, S, 1, _p(g_w), None, _stream()), "nr_accumulate_bwd")
        return g_w, None


def depth_from_weights(weights: Tensor, euclid: Tensor) -> Tensor:
    """render_depth_simple (models/neurad.py:721-728): sum_s w_s (e_s + e_{s+1}) / 2; differentiable in the weights."""
    return _DepthFromWeights.apply(weights, euclid.detach())


# ------------------------------------------------------------------------------------------------ radar point-set loss
def radar_assign(pred: Tensor, detections: Tensor, seg: Tensor, max_detections: int, cost_type: str = "euclidean",
                 workspace: Optional[Tensor] = None) -> Tensor:
    """nr_radar_assign: pred [scans, n, 7], detections [m_total, >= 3], seg [scans + 1] int32 (device) -> assoc [scans, n] int32
    (matched detection of every prediction inside its scan, -1 = none): the Hungarian association of
    radar_utils.py:75-83, no host read."""
    pred, detections = _f32(pred.detach(), "radar predictions"), _f32(detections, "radar detections")
    assert pred.dim() == 3 and pred.shape[2] == 7 and seg.dtype == torch.int32 and seg.is_cuda and seg.numel() == pred.shape[0] + 1
    N, n = pred.shape[:2]
    need = _lib.lib().nr_radar_assign_workspace_bytes(N, n, max_detections)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=pred.device, dtype=torch.uint8)
    assoc = torch.empty((N, n), device=pred.device, dtype=torch.int32)
    check(_lib.lib().nr_radar_assign(_p(pred), N, n, _p(detections), detections.shape[1], _p(seg), max_detections,
                                     {"euclidean": 0, "nll": 1}[cost_type], _p(assoc), _p(workspace), _stream()), "nr_radar_assign")
    assoc.status = radar_status(workspace, N, n, max_detections)  # [scans] int32 view of the workspace: see radar_status
    return assoc


def radar_status(workspace: Tensor, n_scans: int, n_pred: int, max_detections: int) -> Tensor:
    """The per-scan status words nr_radar_assign leaves in its workspace (a device view; reading it is a host sync -- do that
    outside the step): 0 assigned, 1 the search gave up, 2 the scan exceeds max_detections / the kernel's static limits and
    NOTHING was assigned (assoc = -1: nr_radar_loss then trains every existence probability of the scan towards 0)."""
    off = _lib.lib().nr_radar_assign_status_offset(n_scans, n_pred, max_detections)
    return workspace[off:off + 4 * n_scans].view(torch.int32)


def validate_radar_segments(seg: Tensor, max_detections: int) -> None:
    """Data-load-time check (host): every scan of `seg` [scans + 1] holds at most max_detections detections and at most the
    kernel's 8 192 -- what nr_radar_assign would otherwise answer with status 2 in the middle of a captured step."""
    counts = (seg[1:] - seg[:-1]).cpu()
    if counts.numel() and (int(counts.max()) > max_detections or int(counts.max()) > 8192 or int(counts.min()) < 0):
        raise ValueError(f"radar scans with {counts.tolist()} detections: max_detections = {max_detections} (static limit 8192); "
                         "nr_radar_assign would leave such a scan unassigned (status 2)")


class _RadarLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, detections, seg, assoc, loss_type, mult):
        pred = _f32(pred, "radar predictions")
        N, n = pred.shape[:2]
        g_pred = torch.empty_like(pred)
        slots = torch.zeros(_lib.NR_LOSS_SLOTS, device=pred.device, dtype=torch.float32)
        check(_lib.lib().nr_radar_loss(_p(pred), N, n, _p(detections), detections.shape[1], _p(seg), _p(assoc),
                                       {"euclidean": 0, "nll": 1}[loss_type], mult, _p(g_pred), _p(slots), _stream()), "nr_radar_loss")
        ctx.save_for_backward(g_pred)
        return slots.sum()

    @staticmethod
    def backward(ctx, g):
        (g_pred,) = ctx.saved_tensors
        return g_pred * g, None, None, None, None, None


def radar_loss(pred: Tensor, detections: Tensor, seg: Tensor, max_detections: int, loss_type: str = "nll", mult: float = 1.0,
               training: bool = True, workspace: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """calculate_radar_loss (radar_utils.py:54-93) * mult on the device: association on the euclidean cost while training
    (:77-78; on the loss type's own cost otherwise), then the matched loss.  Returns (loss scalar, assoc [scans, n])."""
    detections = _f32(detections, "radar detections")
    assoc = radar_assign(pred, detections, seg, max_detections, "euclidean" if training else loss_type, workspace)
    return _RadarLoss.apply(pred, detections, seg, assoc, loss_type, float(mult)), assoc


class _AppearanceConcat(torch.autograd.Function):
    """nr_appearance_concat_fwd/bwd: [features | temporal appearance embedding] of every ray (neuradar.py:510-512,550-568)."""

    @staticmethod
    def forward(ctx, features, table, times, sensor_idx, duration, embeds_per_sensor):
        features, table, times = _f32(features, "features"), _f32(table, "appearance table"), _f32(times, "times").reshape(-1)
        sensor_idx = sensor_idx.reshape(-1).contiguous()
        assert sensor_idx.dtype == torch.int64 and sensor_idx.is_cuda and times.numel() == features.shape[0] == sensor_idx.numel()
        n, C = features.shape
        A = table.shape[1]
        out = torch.empty(n, C + A, device=features.device, dtype=torch.float32)
        check(_lib.lib().nr_appearance_concat_fwd(_p(features), C, _p(table), A, _p(times), _p(sensor_idx), float(duration),
                                                  int(embeds_per_sensor), 0, n, _p(out), _stream()), "nr_appearance_concat_fwd")
        ctx.save_for_backward(times, sensor_idx)
        ctx.dims = (n, C, A, float(duration), int(embeds_per_sensor), table.shape[0])
        ctx.table_ref = table
        return out

    @staticmethod
    def backward(ctx, g_out):
        times, sensor_idx = ctx.saved_tensors
        n, C, A, duration, E, rows = ctx.dims
        g_out = _f32(g_out, "g_out")
        g_features = torch.empty(n, C, device=g_out.device, dtype=torch.float32)
        table = ctx.table_ref
        direct = (_DIRECT_PARAM_GRADS and table.is_leaf and table.requires_grad and table.grad is not None
                  and table.grad.dtype == torch.float32 and table.grad.is_contiguous())
        g_table = table.grad if direct else torch.zeros(rows, A, device=g_out.device, dtype=torch.float32)
        check(_lib.lib().nr_appearance_concat_bwd(_p(g_out), C, A, _p(times), _p(sensor_idx), duration, E, 0, n, _p(g_features),
                                                  _p(g_table), rows, _stream()), "nr_appearance_concat_bwd")
        return g_features, (None if direct else g_table), None, None, None, None


def appearance_concat(features: Tensor, table: Tensor, times: Tensor, sensor_idx: Tensor, duration: float, embeds_per_sensor: int) -> Tensor:
    """[B, C] rendered features -> [B, C + A]: every ray's features extended by the linear interpolation of its sensor's two
    nearest appearance embeddings in time (table [sensors * embeds_per_sensor, A]; neuradar.py:518-520,550-568) -- one launch
    each way instead of the ~25 elementwise / gather launches of the torch expression and the sort-based embedding backward."""
    return _AppearanceConcat.apply(features, table, times, sensor_idx, duration, embeds_per_sensor)


_ONES: dict = {}


def _ones(n: int, device) -> Tensor:
    key = (n, str(device))
    if key not in _ONES:
        _ONES[key] = torch.ones(n, device=device, dtype=torch.float32)
    return _ONES[key]


class _LinearDirect(torch.autograd.Function):
    """y = x W^T + b on the library GEMM (torch.addmm) for [rows, in] inputs; W / b may be row slices of larger parameters (the
    attention's in_proj_weight).  Inside `direct_param_grads` the backward is three launches -- d x = g W; W.grad += g^T x and
    b.grad += g^T 1 written straight into the gradient buffers (slices of them for sliced parameters) -- instead of autograd's
    two GEMMs + bias reduction + AccumulateGrad adds + the zero-fill and copy of every parameter slice's backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, owner_w, owner_b, row0, rows):
        w = weight if owner_w is None else owner_w[row0:row0 + rows]
        b = bias if owner_b is None else owner_b[row0:row0 + rows]
        y = torch.addmm(b, x, w.t())
        ctx.save_for_backward(x, w)
        ctx.refs = (weight if owner_w is None else owner_w, bias if owner_b is None else owner_b, row0, rows, owner_w is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        pw, pb, row0, rows, sliced = ctx.refs
        g = g.contiguous()
        gx = g @ w if ctx.needs_input_grad[0] else None
        direct = _DIRECT_PARAM_GRADS and all(p.is_leaf and p.requires_grad and p.grad is not None and p.grad.is_contiguous() for p in (pw, pb))
        if direct:
            gw = pw.grad[row0:row0 + rows] if sliced else pw.grad
            gb = pb.grad[row0:row0 + rows] if sliced else pb.grad
            gw.addmm_(g.t(), x)
            gb.addmv_(g.t(), _ones(g.shape[0], g.device))
            return gx, None, None, None, None, None, None
        gw, gb = g.t() @ x, g.sum(0)
        if sliced:  # gradients of the owners: zero outside the slice
            fw, fb = torch.zeros_like(pw), torch.zeros_like(pb)
            fw[row0:row0 + rows], fb[row0:row0 + rows] = gw, gb
            return gx, None, None, fw, fb, None, None
        return gx, gw, gb, None, None, None, None


def linear_direct(x: Tensor, weight: Tensor, bias: Tensor, rows: Optional[Tuple[int, int]] = None) -> Tensor:
    """F.linear(x, weight, bias) -- or, with rows = (first, count), F.linear(x, weight[first:first + count], bias[...]) -- for
    [..., in] float32 inputs, with the three-launch backward of _LinearDirect."""
    lead = x.shape[:-1]
    x2 = _f32(x, "x").reshape(-1, x.shape[-1])
    if rows is None:
        y = _LinearDirect.apply(x2, weight, bias, None, None, 0, weight.shape[0])
    else:
        y = _LinearDirect.apply(x2, None, None, weight, bias, int(rows[0]), int(rows[1]))
    return y.view(*lead, y.shape[-1])


_BN_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_BN_WS: dict = {}


def _ws_key(device):
    """Partials workspaces are per (device, STREAM): two launches that reduce through one workspace on different streams would
    race (the decoder chains run on three streams; a weight gradient launched beside its chain would be a fourth)."""
    return (str(device), torch.cuda.current_stream(device).cuda_stream)


def _bn_workspace(device) -> Tensor:
    key = _ws_key(device)
    if key not in _BN_WS:
        _BN_WS[key] = torch.empty(int(_lib.lib().nr_bn_act_workspace_floats(1, 64)), device=device, dtype=torch.float32)
    return _BN_WS[key]


class _BnAct(torch.autograd.Function):
    """nr_bn_act_fwd/bwd on a channels-last [N, C, H, W] activation: y = act(batch_norm_train(x) + residual)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, momentum, eps, relu):
        if not x.is_cuda:
            raise RuntimeError("x must be a GPU tensor: neuradar_amd has no CPU path (the oracle is test-only)")
        N, C, H, W = x.shape
        assert x.is_contiguous(memory_format=torch.channels_last) and x.dtype in _BN_DTYPES, "channels-last fp32 / bf16 / fp16 activations"
        assert residual is None or (residual.shape == x.shape and residual.dtype == x.dtype
                                    and residual.is_contiguous(memory_format=torch.channels_last))
        gamma, beta = _f32(gamma, "weight"), _f32(beta, "bias")
        M = N * H * W
        y = torch.empty_like(x)  # (keeps the channels-last strides)
        mean, rstd = torch.empty(C, device=x.device), torch.empty(C, device=x.device)
        ws = _bn_workspace(x.device)
        check(_lib.lib().nr_bn_act_fwd(_p(x), _p(residual), M, C, _BN_DTYPES[x.dtype], _p(gamma), _p(beta), float(eps), float(momentum),
                                       _p(running_mean), _p(running_var), int(relu), _p(y), _p(mean), _p(rstd), _p(ws), _stream()),
              "nr_bn_act_fwd")
        ctx.save_for_backward(x, y, gamma, mean, rstd)
        ctx.relu, ctx.has_res = int(relu), residual is not None
        ctx.param_refs = (gamma, beta)
        ctx.mark_non_differentiable(mean, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y, gamma, mean, rstd = ctx.saved_tensors
        N, C, H, W = x.shape
        g = g.contiguous(memory_format=torch.channels_last)
        if g.dtype != x.dtype:
            g = g.to(x.dtype)
        dx = torch.empty_like(x)
        d_res = torch.empty_like(x) if ctx.has_res else None
        refs = ctx.param_refs
        direct = _DIRECT_PARAM_GRADS and all(p.is_leaf and p.requires_grad and p.grad is not None and p.grad.dtype == torch.float32
                                             and p.grad.is_contiguous() for p in refs)
        gg, gb = (refs[0].grad, refs[1].grad) if direct else (torch.zeros(C, device=x.device), torch.zeros(C, device=x.device))
        check(_lib.lib().nr_bn_act_bwd(_p(g), _p(y), _p(x), N * H * W, C, _BN_DTYPES[x.dtype], _p(gamma), _p(mean), _p(rstd), ctx.relu,
                                       _p(dx), _p(d_res), _p(gg), _p(gb), _p(_bn_workspace(x.device)), _stream()), "nr_bn_act_bwd")
        return dx, (None if direct else gg), (None if direct else gb), None, None, d_res, None, None, None


def bn_act(x: Tensor, weight: Tensor, bias: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor],
           residual: Optional[Tensor] = None, momentum: float = 0.1, eps: float = 1e-5, relu: bool = True) -> Tensor:
    """act(BatchNorm2d_train(x) + residual) on a channels-last activation in two launches each way (nr_bn_act_fwd/bwd); updates
    the running statistics in place like the module does."""
    return _BnAct.apply(x, weight, bias, running_mean, running_var, residual, momentum, eps, relu)


def _radar_heads_struct(tensors) -> "_lib.NrRadarHeads":
    st = _lib.NrRadarHeads()
    for h in range(3):
        for l in range(3):
            st.weight[h][l] = tensors[(h * 3 + l) * 2].data_ptr()
            st.bias[h][l] = tensors[(h * 3 + l) * 2 + 1].data_ptr()
    return st


class _RadarHeads(torch.autograd.Function):
    """nr_radar_heads_fwd/bwd: (transformer output [n, C], rendered points [n, 3], 18 parameter tensors: head (offset, existence,
    uncertainty) x layer x (weight, bias)) -> radar_output [n, 7]."""

    @staticmethod
    def forward(ctx, x, xyz, *params):
        x, xyz = _f32(x, "x"), _f32(xyz, "xyz")
        ps = [_f32(p, "head parameter") for p in params]
        n, C = x.shape
        assert len(ps) == 18 and xyz.shape == (n, 3)
        for h, k_out in enumerate((3, 1, 3)):
            assert ps[h * 6].shape == (16, C) and ps[h * 6 + 2].shape == (16, 16) and ps[h * 6 + 4].shape == (k_out, 16), "heads are C -> 16 -> 16 -> 3 | 1 | 3"
        out = torch.empty(n, 7, device=x.device, dtype=torch.float32)
        check(_lib.lib().nr_radar_heads_fwd(byref(_radar_heads_struct(ps)), _p(x), C, _p(xyz), n, _p(out), _stream()), "nr_radar_heads_fwd")
        ctx.save_for_backward(x, *ps)
        ctx.param_refs = params
        return out

    @staticmethod
    def backward(ctx, g):
        x, *ps = ctx.saved_tensors
        n, C = x.shape
        g = _f32(g, "g")
        refs = ctx.param_refs
        direct = _DIRECT_PARAM_GRADS and all(p.is_leaf and p.requires_grad and p.grad is not None and p.grad.dtype == torch.float32
                                             and p.grad.is_contiguous() and p.dtype == torch.float32 and p.is_contiguous() for p in refs)
        grads = [p.grad for p in refs] if direct else [torch.zeros_like(p) for p in ps]
        gx, gxyz = torch.empty_like(x), torch.empty(n, 3, device=x.device, dtype=torch.float32)
        check(_lib.lib().nr_radar_heads_bwd(byref(_radar_heads_struct(ps)), _p(x), C, _p(g), n, _p(gx), _p(gxyz),
                                            byref(_radar_heads_struct(grads)), _stream()), "nr_radar_heads_bwd")
        return (gx, gxyz, *([None] * 18 if direct else grads))


def radar_heads(x: Tensor, xyz: Tensor, offset_head, existence_head, uncertainty_head) -> Tensor:
    """radar_output [n, 7] = [sigmoid(existence(x)), xyz + 1.5 tanh(offset(x)), softplus(uncertainty(x))] -- the three MLP heads of
    the radar decoder (in -> 16 -> 16 -> 3 | 1 | 3) and the assembly of neuradar.py:480-491 in one launch each way."""
    params = []
    for head in (offset_head, existence_head, uncertainty_head):
        for layer in head.layers:
            params += [layer.weight, layer.bias]
    return _RadarHeads.apply(x, xyz, *params)


_POSEMB_TABLES: dict = {}


def _posemb_tables(C: int, temperature: float, device) -> Tuple[Tensor, Tensor]:
    """Per-channel (dim_t, axis * 2 + is_cos) of PositionEmbeddingCoordsSine for three input axes (position_encoding_3d.py:
    66-84: C // 3 channels per axis rounded down to even, the remainder handed out two at a time from the first axis on),
    dim_t in torch's own float32 arithmetic."""
    key = (C, float(temperature), str(device))
    if key not in _POSEMB_TABLES:
        ndim = C // 3
        ndim -= ndim % 2
        rems = C - ndim * 3
        dim_t, code = [], []
        for d in range(3):
            cdim = ndim
            if rems > 0:
                cdim += 2
                rems -= 2
            k = torch.arange(cdim, dtype=torch.float32, device=device)
            dim_t.append(temperature ** (2 * torch.div(k, 2, rounding_mode="floor") / cdim))
            code.append(2 * d + (torch.arange(cdim, device=device) % 2))
        _POSEMB_TABLES[key] = (torch.cat(dim_t).contiguous(), torch.cat(code).to(torch.int32).contiguous())
    return _POSEMB_TABLES[key]


class _RadarPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, dirs_spher, C, temperature):
        ctx.depth_shape = depth.shape
        depth, dirs_spher = _f32(depth, "depth").reshape(-1), _f32(dirs_spher, "directions_spher").reshape(-1, 2)
        n = depth.shape[0]
        dim_t, code = _posemb_tables(C, temperature, depth.device)
        xyz, dirs = torch.empty(n, 3, device=depth.device), torch.empty(n, 3, device=depth.device)
        pos = torch.empty(n, C, device=depth.device)
        check(_lib.lib().nr_radar_points_fwd(_p(depth), _p(dirs_spher), n, _p(dim_t), _p(code), C, _p(xyz), _p(dirs), _p(pos), _stream()),
              "nr_radar_points_fwd")
        ctx.save_for_backward(dirs)
        ctx.mark_non_differentiable(pos)
        return xyz, pos

    @staticmethod
    def backward(ctx, g_xyz, _g_pos):
        (dirs,) = ctx.saved_tensors
        g_xyz = _f32(g_xyz, "g_xyz")
        g_depth = torch.empty(dirs.shape[0], device=dirs.device)
        check(_lib.lib().nr_radar_points_bwd(_p(g_xyz), _p(dirs), dirs.shape[0], _p(g_depth), _stream()), "nr_radar_points_bwd")
        return g_depth.view(ctx.depth_shape), None, None, None


def radar_points(depth: Tensor, dirs_spher: Tensor, num_channels: int, temperature: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """nr_radar_points_fwd: depth [n], (azimuth, elevation) [n, 2] -> (xyz [n, 3], differentiable in depth; sine position
    embedding of xyz [n, num_channels], constant) -- neuradar.py:470-476 + position_encoding_3d.py:56-103."""
    return _RadarPoints.apply(depth, dirs_spher, int(num_channels), float(temperature))


# ------------------------------------------------------------------------------------------------ 7 x 7 convolutions of the RGB decoder
_DT16 = {torch.bfloat16: 1, torch.float16: 2}


def conv7_pack(flat16: Tensor, weight_offsets, bias_offsets, images: Optional[Tensor] = None) -> Tensor:
    """nr_conv7_pack: the 7 x 7 convolutions whose [32, 7, 7, 32] (channels-last) 16-bit weights sit at element offsets
    `weight_offsets` of the flat 16-bit buffer `flat16` (biases [32] at `bias_offsets`, -1 = none) -> images uint8
    [n, 2, nr_conv7_image_bytes() / 2]: [k, 0] the convolution's LDS image, [k, 1] its data gradient's."""
    n = len(weight_offsets)
    half = _lib.lib().nr_conv7_image_bytes() // 2
    if images is None:
        images = torch.empty(n, 2, half, device=flat16.device, dtype=torch.uint8)
    lst = _lib.NrConv7List()
    lst.n = n
    for k in range(n):
        lst.offset[k], lst.bias_offset[k] = int(weight_offsets[k]), int(bias_offsets[k])
    check(_lib.lib().nr_conv7_pack(_p(flat16), byref(lst), _DT16[flat16.dtype], _p(images), _stream()), "nr_conv7_pack")
    return images


def conv7_fold_pack(pairs, dtype: torch.dtype, images: Optional[Tensor], state: Tensor, force: bool) -> Tensor:
    """nr_conv7_fold_pack: `pairs` = [(Conv2d 32 -> 32 7 x 7, BatchNorm2d)] in eval mode -> images uint8 [n, 2, bytes / 2] whose
    [k, 0] is the LDS image of BN_k(conv_k(.)) as ONE convolution, built from the fp32 master parameters in one launch that
    early-outs ON THE DEVICE while `state[0]` (two device int32 of the caller, zero-initialised) matches the device's
    parameter-generation word -- unless `force`.  state[1] counts the rebuilds."""
    n = len(pairs)
    half = _lib.lib().nr_conv7_image_bytes() // 2
    dev = pairs[0][0].weight.device
    if images is None:
        images = torch.zeros(n, 2, half, device=dev, dtype=torch.uint8)
        force = True
    lst = _lib.NrConv7Fold()
    lst.n = n
    for k, (cv, bn) in enumerate(pairs):
        w = cv.weight
        if w.dtype != torch.float32 or w.shape != (32, 32, 7, 7) or w.stride(2) != 7 * w.stride(3):
            raise NotImplementedError("conv7_fold_pack: fp32 [32, 32, 7, 7] weights whose taps are one strided axis")
        for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var):
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise NotImplementedError("conv7_fold_pack: fp32 contiguous batch-norm tensors")
        lst.weight[k], lst.stride_o[k], lst.stride_i[k], lst.stride_t[k] = w.data_ptr(), w.stride(0), w.stride(1), w.stride(3)
        lst.bias[k] = cv.bias.data_ptr() if cv.bias is not None else None
        lst.gamma[k], lst.beta[k] = bn.weight.data_ptr(), bn.bias.data_ptr()
        lst.mean[k], lst.var[k], lst.eps[k] = bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.eps)
    _stream_ = _stream()  # (also: nr_init on this device -- the generation word exists)
    check(_lib.lib().nr_conv7_fold_pack(byref(lst), _DT16[dtype], _p(images), _p(state), int(bool(force)), _stream_), "nr_conv7_fold_pack")
    return images


def conv7_forward(x: Tensor, image: Tensor, residual: Optional[Tensor] = None, relu: bool = False) -> Tensor:
    """nr_conv7_fwd on a channels-last 16-bit activation x [P, 32, H, W] (memory [P, H, W, 32]) -> the same shape and format."""
    P, C, H, W = x.shape
    assert C == 32 and x.dtype in _DT16 and x.is_contiguous(memory_format=torch.channels_last) and x.is_cuda
    if residual is not None:
        assert residual.shape == x.shape and residual.dtype == x.dtype and residual.is_contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(x, memory_format=torch.channels_last)
    check(_lib.lib().nr_conv7_fwd(_p(x), _p(image), _p(residual), int(relu), _p(y), P, H, W, _DT16[x.dtype], _stream()), "nr_conv7_fwd")
    return y


_conv7_ws = {}


def conv7_wgrad(x: Tensor, grad_y: Tensor, want_bias: bool = True, into: Optional[Tuple[Tensor, Optional[Tensor]]] = None):
    """nr_conv7_wgrad: x, grad_y [P, 32, H, W] channels-last 16-bit -> (grad_weight, logically [32, 32, 7, 7] in the parameter's
    channels-last memory [O, kh, kw, I]; grad_bias [32] or None), 16-bit, fp32 accumulation.  into = (weight.grad, bias.grad):
    ADDED into those buffers instead (weight.grad in the same channels-last memory), returns (None, None)."""
    P, C, H, W = x.shape
    assert C == 32 and x.dtype in _DT16 and grad_y.dtype == x.dtype and grad_y.shape == x.shape
    assert x.is_contiguous(memory_format=torch.channels_last) and grad_y.is_contiguous(memory_format=torch.channels_last)
    ws = _conv7_ws.get(_ws_key(x.device))
    if ws is None:
        ws = _conv7_ws[_ws_key(x.device)] = torch.empty(_lib.lib().nr_conv7_wgrad_workspace_bytes(), device=x.device, dtype=torch.uint8)
    if into is not None:
        check(_lib.lib().nr_conv7_wgrad(_p(x), _p(grad_y), _p(into[0]), _p(into[1]), 1, _p(ws), P, H, W, _DT16[x.dtype], _stream()),
              "nr_conv7_wgrad")
        return None, None
    gw = torch.empty(32 * 49 * 32, device=x.device, dtype=x.dtype)
    gb = torch.empty(32, device=x.device, dtype=x.dtype) if want_bias else None
    check(_lib.lib().nr_conv7_wgrad(_p(x), _p(grad_y), _p(gw), _p(gb), 0, _p(ws), P, H, W, _DT16[x.dtype], _stream()), "nr_conv7_wgrad")
    return gw.view(32, 7, 7, 32).permute(0, 3, 1, 2), gb


class _Conv7(torch.autograd.Function):
    """Conv2d(32, 32, 7, padding=3) on channels-last 16-bit activations: forward and DATA gradient on nr_conv7_fwd (the two
    orientations of the packed weights), weight / bias gradient on nr_conv7_wgrad (NR_CONV7_WGRAD=0: aten.convolution_backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, images):
        ctx.save_for_backward(x, weight)
        ctx.images, ctx.has_bias = images, bias is not None
        ctx.param_refs = (weight, bias)  # the caller's tensors (their .grad, for direct_param_grads)
        return conv7_forward(x, images[0])

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous(memory_format=torch.channels_last)
        gw = gb = None
        if (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])) and os.environ.get("NR_CONV7_WGRAD", "1") != "0":
            w_, b_ = ctx.param_refs
            chl = (32 * 49, 1, 7 * 32, 32)  # the channels-last memory [O, kh, kw, I] of a [32, 32, 7, 7] tensor
            direct = (_DIRECT_PARAM_GRADS and w_.is_leaf and w_.grad is not None and w_.grad.dtype == x.dtype and w_.grad.stride() == chl
                      and (b_ is None or (b_.is_leaf and b_.grad is not None and b_.grad.dtype == x.dtype and b_.grad.is_contiguous())))
            # (direct: added straight into the parameters' .grad buffers -- no temporaries, no AccumulateGrad adds)
            if direct:
                conv7_wgrad(x, g, ctx.has_bias, into=(w_.grad, None if b_ is None else b_.grad))
            else:
                gw, gb = conv7_wgrad(x, g, ctx.has_bias)
            gx = conv7_forward(g, ctx.images[1]) if ctx.needs_input_grad[0] else None
            return gx, gw, gb, None
        gx = conv7_forward(g, ctx.images[1]) if ctx.needs_input_grad[0] else None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            _, gw, gb = torch.ops.aten.convolution_backward(g, x, weight, [32] if ctx.has_bias else None, [1, 1], [3, 3], [1, 1], False,
                                                            [0, 0], 1, [False, True, ctx.has_bias])
        return gx, gw, gb, None


def conv7(x: Tensor, weight: Tensor, bias: Optional[Tensor], images: Tensor) -> Tensor:
    """x [P, 32, H, W] channels-last 16-bit; weight / bias: the convolution's (16-bit) parameters, for the library's weight
    gradient; images [2, bytes]: their nr_conv7_pack images of THIS optimizer step."""
    return _Conv7.apply(x, weight, bias, images)


# ------------------------------------------------------------------------------------------------ pointwise convolutions (pw.hip)
_pw_ws: dict = {}


class _Pointwise(torch.autograd.Function):
    """nr_pw_fwd / nr_pw_bwd_data / nr_pw_bwd_weight: y = act(x W^T + b) per pixel on channels-last activations -- a 1 x 1
    Conv2d, or (transposed) ConvTranspose2d(3, stride 3).  x2d [pixels, K] fp32 or 16-bit, y [pixels (x 9), O] 16-bit or fp32."""

    @staticmethod
    def forward(ctx, x2d, weight, bias, cfg):
        act, transposed, H, W, out_f32, scale = cfg
        P, K = x2d.shape
        O = weight.shape[1] if transposed else weight.shape[0]
        dt = _DT16[weight.dtype]
        y = torch.empty((9 * P if transposed else P, O), device=x2d.device, dtype=torch.float32 if out_f32 else weight.dtype)
        check(_lib.lib().nr_pw_fwd(_p(x2d), int(x2d.dtype == torch.float32), _p(weight), _p(bias), _p(y), int(out_f32), P, K, O, act,
                                   int(transposed), H, W, dt, _stream()), "nr_pw_fwd")
        ctx.save_for_backward(x2d, weight, y)
        ctx.cfg, ctx.param_refs = cfg, (weight, bias)
        return y

    @staticmethod
    def backward(ctx, g):
        x2d, weight, y = ctx.saved_tensors
        act, transposed, H, W, out_f32, scale = ctx.cfg
        P, K = x2d.shape
        O = y.shape[1]
        dt = _DT16[weight.dtype]
        g = g.contiguous()
        assert g.dtype == y.dtype
        lib = _lib.lib()
        x_f32 = int(x2d.dtype == torch.float32)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x2d)
            check(lib.nr_pw_bwd_data(_p(g), _p(y), int(out_f32), _p(weight), _p(gx), x_f32, P, K, O, act, int(transposed), H, W, _p(scale),
                                     dt, _stream()), "nr_pw_bwd_data")
        w_, b_ = ctx.param_refs
        ws = _pw_ws.get(_ws_key(x2d.device))
        if ws is None:
            ws = _pw_ws[_ws_key(x2d.device)] = torch.empty(lib.nr_pw_workspace_bytes(), device=x2d.device, dtype=torch.uint8)
        direct = (_DIRECT_PARAM_GRADS and w_.is_leaf and w_.grad is not None and w_.grad.dtype == w_.dtype and w_.grad.stride() == w_.stride()
                  and (b_ is None or (b_.is_leaf and b_.grad is not None and b_.grad.dtype == b_.dtype and b_.grad.is_contiguous())))
        if direct:  # added straight into the parameters' .grad buffers (same memory layout as the parameters)
            gw, gb = w_.grad, (None if b_ is None else b_.grad)
        else:
            gw = torch.empty_strided(weight.shape, weight.stride(), device=weight.device, dtype=weight.dtype)
            gb = None if b_ is None else torch.empty_like(b_)
        check(lib.nr_pw_bwd_weight(_p(x2d), x_f32, _p(g), _p(y), int(out_f32), _p(gw), _p(gb), int(direct), _p(ws), P, K, O, act,
                                   int(transposed), H, W, dt, _stream()), "nr_pw_bwd_weight")
        return gx, (None if direct else gw), (None if direct else gb), None


def pointwise_ok(x2d: Tensor, weight: Tensor, bias: Optional[Tensor], transposed: bool = False) -> bool:
    """The layouts nr_pw_* are built for: x2d [pixels, K] contiguous (fp32 or the parameters' 16-bit type), K in {32, 48}; a 1 x 1
    Conv2d weight [O, K, 1, 1] with O <= 32, or the ConvTranspose2d(3, stride 3) weight [K, O, 3, 3] in channels-last memory."""
    if not (x2d.is_cuda and x2d.dim() == 2 and x2d.is_contiguous() and weight.dtype in _DT16 and x2d.dtype in (torch.float32, weight.dtype)):
        return False
    K = x2d.shape[1]
    if K not in (32, 48) or (bias is not None and (bias.dtype != weight.dtype or not bias.is_contiguous())):
        return False
    if transposed:
        return (weight.dim() == 4 and weight.shape[0] == K and tuple(weight.shape[2:]) == (3, 3) and weight.shape[1] % 8 == 0
                and weight.shape[1] <= 32 and weight.stride() == (9 * weight.shape[1], 1, 3 * weight.shape[1], weight.shape[1]))
    return (weight.dim() == 4 and weight.shape[1] == K and tuple(weight.shape[2:]) == (1, 1) and weight.shape[0] <= 32
            and weight.stride(0) == K and weight.stride(1) == 1)


def pointwise(x2d: Tensor, weight: Tensor, bias: Optional[Tensor], act: int = 0, out_f32: bool = False, grad_scale: Optional[Tensor] = None) -> Tensor:
    """act(Conv2d(K, O, 1)(x)) on the pixels' rows: x2d [pixels, K] (= a channels-last [P, K, H, W] tensor's memory) ->
    [pixels, O]; act 0 none / 1 ReLU / 2 sigmoid; out_f32: fp32 output (and incoming gradient); grad_scale: device scalar the
    input gradient is multiplied by (the inverse loss scale where the gradient leaves a loss-scaled 16-bit backward)."""
    assert pointwise_ok(x2d, weight, bias)
    return _Pointwise.apply(x2d, weight, bias, (int(act), False, 0, 0, bool(out_f32), grad_scale))


def conv_transpose3(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """ConvTranspose2d(K, O, 3, stride=3) on a channels-last 16-bit [P, K, H, W] tensor -> channels-last [P, O, 3H, 3W]."""
    P, K, H, W = x.shape
    assert x.is_contiguous(memory_format=torch.channels_last)
    x2d = x.permute(0, 2, 3, 1).reshape(-1, K)
    assert pointwise_ok(x2d, weight, bias, transposed=True)
    y = _Pointwise.apply(x2d, weight, bias, (0, True, H, W, False, None))
    return y.view(P, 3 * H, 3 * W, weight.shape[1]).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------ optimizer
def adam_step(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, lr: float, step: int,
              betas=(0.9, 0.999), eps: float = 1e-15, weight_decay: float = 0.0, adamw: bool = False,
              grad_scale: float = 1.0, zero_grad: bool = True, dev_hyper: Optional[Tensor] = None,
              seen_grad: Optional[Tensor] = None, marked: bool = False, skip: Optional[Tensor] = None,
              delta16: Optional[Tensor] = None) -> None:
    """seen_grad: optional uint8 [numel/4], zero while exp_avg / exp_avg_sq are zero (see include/neuradar_hip.h).
    marked: the scatter sets the bytes (nr_hash_encode_bwd_marked) -- never-marked groups are skipped without reading their gradient.
    skip: optional device float (a loss scaler's found-inf flag): non-zero = no update, the gradient is still cleared (2.0: kept).
    delta16: optional bf16 [numel]: the update leaves as a rounded delta that is also what the owner applies (sharded DP step)."""
    if delta16 is not None:
        assert delta16.dtype == torch.bfloat16 and delta16.numel() == param.numel() and delta16.is_contiguous() and not marked
    if seen_grad is not None:
        assert seen_grad.dtype == torch.uint8 and seen_grad.numel() >= param.numel() // 4
    # moments: two contiguous arrays, or the two halves of one array of [exp_avg x 4 | exp_avg_sq x 4] records (FlatAdam's tables)
    stride = 1
    if not exp_avg.is_contiguous():
        assert exp_avg.dim() == 2 and exp_avg.stride() == (8, 1) and exp_avg_sq.stride() == (8, 1) and \
            exp_avg_sq.data_ptr() == exp_avg.data_ptr() + 16, "moments must be contiguous or the halves of one interleaved array"
        stride = 2
    if marked:
        assert seen_grad is not None and weight_decay == 0.0 and param.numel() % 4 == 0
        check(_lib.lib().nr_adam_step_marked(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, betas[0], betas[1],
                                             eps, step, grad_scale, int(zero_grad), _p(dev_hyper), _p(seen_grad), _p(skip), stride, _stream()),
              "nr_adam_step_marked")
        return
    check(_lib.lib().nr_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, betas[0],
                                  betas[1], eps, weight_decay, int(adamw), step, grad_scale, int(zero_grad),
                                  _p(dev_hyper), _p(seen_grad), _p(skip), _p(delta16), stride, _stream()), "nr_adam_step")


def hash_mark_vertices(x: Tensor, scalings: Tensor, log2_hashmap_size: int, stamp: Tensor, epoch: Tensor) -> None:
    """nr_hash_mark_vertices: stamp (uint8 [L * T]) <- the step's value at every table entry the rows' gradients can reach."""
    assert stamp.dtype == torch.uint8 and stamp.numel() >= scalings.numel() << log2_hashmap_size and epoch.dtype == torch.float32
    check(_lib.lib().nr_hash_mark_vertices(_p(x), _p(scalings), scalings.numel(), log2_hashmap_size, x.shape[0], _p(stamp), _p(epoch), _stream()),
          "nr_hash_mark_vertices")


def adam_step_split(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, betas, eps: float, grad_scale: float, dev_hyper: Tensor,
                    seen_grad: Tensor, stamp: Tensor, epoch: Tensor, phase: int) -> None:
    """nr_adam_step_split: phase 1 = the zero-gradient update of the groups with a history that this step does not touch (may run
    beside the step's forward / backward); phase 2 = the full update of the groups stamped this step (after the scatter)."""
    stride = 1
    if not exp_avg.is_contiguous():
        assert exp_avg.dim() == 2 and exp_avg.stride() == (8, 1) and exp_avg_sq.data_ptr() == exp_avg.data_ptr() + 16
        stride = 2
    assert param.numel() % 4 == 0 and seen_grad.numel() >= param.numel() // 4 and stamp.numel() >= param.numel() // 4
    check(_lib.lib().nr_adam_step_split(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), betas[0], betas[1], eps, grad_scale,
                                        _p(dev_hyper), _p(seen_grad), _p(stamp), _p(epoch), int(phase), stride, _stream()), "nr_adam_step_split")


def apply_delta16(param: Tensor, delta16: Tensor, lo: int, hi: int) -> None:
    """param[i] += float(delta16[i]) outside [lo, hi): the receiving side of the sharded table step (nr_apply_delta16)."""
    assert delta16.dtype == torch.bfloat16 and delta16.numel() == param.numel() and param.is_contiguous() and delta16.is_contiguous()
    check(_lib.lib().nr_apply_delta16(_p(param), _p(delta16), param.numel(), int(lo), int(hi), _stream()), "nr_apply_delta16")


def grad_to16_clear(grad: Tensor, low16: Tensor) -> None:
    """low16 = bf16(grad); grad = 0 (nr_grad_to16_clear)."""
    assert low16.dtype == torch.bfloat16 and low16.numel() == grad.numel() and grad.is_contiguous() and low16.is_contiguous()
    check(_lib.lib().nr_grad_to16_clear(_p(grad), _p(low16), grad.numel(), _stream()), "nr_grad_to16_clear")


# ------------------------------------------------------------------------------------------------ loss tail
def supervision_loss(features: Tensor, target_f: Tensor, depth: Tensor, target_d: Tensor, rgb_mult: float,
                     depth_mult: float, loss: Tensor, g_features: Optional[Tensor] = None,
                     g_depth: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """loss[0] += rgb_mult*MSE(features[:, :C], target_f) + depth_mult*L1(depth, target_d); returns the
    gradients (g_features zero beyond the first C columns)."""
    B, C = target_f.shape
    g_features = torch.zeros_like(features) if g_features is None else g_features
    g_depth = torch.empty_like(depth) if g_depth is None else g_depth
    check(_lib.lib().nr_supervision_loss(_p(features), features.shape[1], _p(target_f), C, _p(depth), _p(target_d), B,
                                         rgb_mult, depth_mult, _p(g_features), _p(g_depth), _p(loss), _stream()),
          "nr_supervision_loss")
    return g_features, g_depth


def distortion_loss(spacing: Tensor, weights: Tensor, n_used: int, mult: float, loss: Tensor,
                    g_w: Optional[Tensor] = None) -> Tensor:
    """loss[0] += mult*distortion(spacing[:, :n_used+1], weights[:, :n_used]); returns d loss / d weights."""
    g_w = torch.empty_like(weights) if g_w is None else g_w
    check(_lib.lib().nr_distortion_loss(_p(spacing), spacing.shape[1], _p(weights), weights.shape[1], n_used,
                                        weights.shape[0], mult, _p(g_w), _p(loss), _stream()), "nr_distortion_loss")
    return g_w


def interlevel_loss(spacing: Tensor, weights: Tensor, n_used: int, prop_spacing: Tensor, prop_weights: Tensor,
                    pulse: float, mult: float, loss: Tensor, g_wp: Optional[Tensor] = None) -> Tensor:
    """loss[0] += mult * zipnerf inter-level loss of one proposal level; returns d loss / d prop_weights."""
    g_wp = torch.empty_like(prop_weights) if g_wp is None else g_wp
    check(_lib.lib().nr_interlevel_loss(_p(spacing), spacing.shape[1], _p(weights), weights.shape[1], n_used,
                                        _p(prop_spacing), _p(prop_weights), prop_weights.shape[1], weights.shape[0],
                                        pulse, mult, _p(g_wp), _p(loss), _stream()), "nr_interlevel_loss")
    return g_wp


# ------------------------------------------------------------------------------------------------ sparse gradient lists
def grad_compact(grad: Tensor, row_width: int, idx: Tensor, val: Tensor, count: Tensor) -> None:
    """Move the non-zero rows of grad [rows*row_width] into (idx, val), zero them in grad; count += rows found."""
    g = _f32(grad, "grad")
    if g.data_ptr() != grad.data_ptr():
        raise RuntimeError("grad_compact works in place: grad must be a contiguous float32 tensor")
    check(_lib.lib().nr_grad_compact(_p(g), g.numel() // row_width, row_width, idx.numel(), _p(idx), _p(val), _p(count),
                                     _stream()), "nr_grad_compact")


def grad_apply(idx: Tensor, val: Tensor, count: Tensor, row_width: int, grad: Tensor) -> None:
    """grad[idx[i]] += val[i] for i < min(count, len(idx))."""
    g = _f32(grad, "grad")
    if g.data_ptr() != grad.data_ptr():
        raise RuntimeError("grad_apply works in place: grad must be a contiguous float32 tensor")
    if idx.numel() == 0:
        return
    check(_lib.lib().nr_grad_apply(_p(idx), _p(val), _p(count), idx.numel(), row_width, _p(g), _stream()), "nr_grad_apply")


def grad_apply_guarded(idx: Tensor, val: Tensor, counts: Tensor, list_rank: int, own_rank: int, m: int, row_width: int, grad: Tensor,
                       flag: Tensor) -> None:
    """nr_grad_apply_guarded: the list of `list_rank` (capacity m) into grad unless some rank's count exceeds m -- then only the
    own list, and flag = 2 (the optimizer keeps the gradient for the next step)."""
    check(_lib.lib().nr_grad_apply_guarded(_p(idx), _p(val), _p(counts), counts.numel(), int(list_rank), int(own_rank), int(m), row_width,
                                           _p(grad), _p(flag), _stream()), "nr_grad_apply_guarded")


def grad_compact_shards(grad: Tensor, row_width: int, world: int, caps: Tensor, idx: Tensor, val: Tensor, counts: Tensor) -> None:
    """nr_grad_compact_shards: the non-zero rows of each of the `world` equal row shards of grad move into that destination's
    segment of (idx, val) (segment d: caps[d] rows starting at sum(caps[:d]); indices relative to the shard) and are cleared;
    counts[d] += all non-zero rows of shard d."""
    g = _f32(grad, "grad")
    if g.data_ptr() != grad.data_ptr() or g.numel() % (world * row_width) != 0:
        raise RuntimeError("grad_compact_shards works in place on a contiguous float32 gradient of world equal row shards")
    check(_lib.lib().nr_grad_compact_shards(_p(g), g.numel() // (world * row_width), row_width, world, _p(caps), _p(idx), _p(val), _p(counts),
                                            _stream()), "nr_grad_compact_shards")


def grad_lists_apply(idx: Tensor, val: Tensor, counts: Tensor, caps: Tensor, src_rank: int, own_rank: int, row_width: int,
                     shard: Tensor, flag: Optional[Tensor]) -> None:
    """nr_grad_lists_apply: the list received from src_rank onto this rank's shard of the gradient, unless any list of the
    exchange overflowed (then flag = 2: the owner's Adam skips the step and the gradient is kept)."""
    world = caps.numel()
    check(_lib.lib().nr_grad_lists_apply(_p(idx), _p(val), idx.numel(), _p(counts), _p(caps), world, int(src_rank), int(own_rank), row_width,
                                         _p(shard), _p(flag), _stream()), "nr_grad_lists_apply")


def grad_lists_restore(idx: Tensor, val: Tensor, max_cap: int, counts: Tensor, caps: Tensor, own_rank: int, row_width: int, grad: Tensor,
                       found_inf: Optional[Tensor] = None) -> None:
    """nr_grad_lists_restore: after an overflowed exchange the rank's own send lists go back into its local gradient (no-op
    otherwise) -- unless the loss scaler's flag `found_inf` is raised: then the whole local gradient is cleared instead."""
    world = caps.numel()
    check(_lib.lib().nr_grad_lists_restore(_p(idx), _p(val), int(max_cap), _p(counts), _p(caps), world, int(own_rank),
                                           grad.numel() // (world * row_width), row_width, _p(grad), _p(found_inf), _stream()),
          "nr_grad_lists_restore")


def uniform_fill(out: Tensor, seed: int, epoch: Optional[Tensor] = None) -> Tensor:
    """out <- U[0,1), draw number `epoch[0]` (device float counter) of the stream `seed` (nr_uniform_fill)."""
    check(_lib.lib().nr_uniform_fill(_p(_f32(out, "out")), out.numel(), seed & 0xFFFFFFFF, _p(epoch), _stream()), "nr_uniform_fill")
    return out
