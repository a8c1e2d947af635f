"""Oracle: the NeuRADField MLP stack in reduced precision (bf16 / fp16 operands, fp32 accumulation).
Test infrastructure only.

The reference reaches reduced precision through `torch.autocast` around the whole model plus tcnn's fp16
`FullyFusedMLP` (engine/trainer.py:189-200,564-595; field_components/mlp.py:109-127): every Linear's input and weight
are rounded to the 16-bit type, products are accumulated wider, gradients are loss-scaled (GradScaler).  This file
restates THAT arithmetic at the rounding points the HIP kernels use (neuradar_amd/csrc/mlp_lp.hip), forward and a
hand-written backward, so that the kernels can be checked to accumulation-order accuracy instead of only to a loose
"it is bf16" tolerance:

  * rounded (round-to-nearest-even) to 16 bit: the grid features, every weight matrix, every hidden activation after
    its ReLU, the embedding e and the SH vector where they enter mlp_feature, and every gradient tile where it enters a
    matrix product (after multiplication by grad_scale);
  * kept fp32: biases, all accumulations, the sdf row of mlp_geo.layers[1] (a dot product on the un-rounded hidden
    activations), the residual e + o, the sigmoid, d_beta, the bias gradients' sums.

Structure of the network: fields/neurad_field.py:128-152 (see oracle/field.py for the fp32 restatement).
"""
from typing import Dict, List, Tuple

import torch

from .field import BETA_MIN, direction_encoding

DTYPES = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}  # float32: no rounding (self-check)


def q(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Round to the 16-bit type (nearest even) and back to fp32."""
    return x.to(dtype).to(torch.float32)


def field_mlp_lp(feats: torch.Tensor, directions: torch.Tensor, geo: List[Tuple[torch.Tensor, torch.Tensor]],
                 feat: List[Tuple[torch.Tensor, torch.Tensor]], beta: torch.Tensor, dtype: str, g_feature=None,
                 g_alpha=None, grad_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """feats [N, 32] (grid features, after the per-level rescale), directions [N, 3] -> feature [N,C], sdf [N], alpha [N];
    with g_feature [N,C] / g_alpha [N] also every gradient (g_feats, g_geo_w*, g_geo_b*, g_feat_w*, g_feat_b*, g_beta)."""
    dt = DTYPES[dtype]
    (wg0, bg0), (wg1, bg1) = geo
    (wf0, bf0), (wf1, bf1), (wf2, bf2) = feat
    C = wf2.shape[0]
    x0 = q(feats, dt)
    h1 = torch.relu(x0 @ q(wg0, dt).T + bg0)
    sdf = h1 @ wg1[0] + bg1[0]  # fp32 row on the un-rounded activations
    h1q = q(h1, dt)
    e = h1q @ q(wg1[1:], dt).T + bg1[1:]
    sh = direction_encoding(directions)
    cat = torch.cat([q(e, dt), q(sh, dt)], dim=-1)
    f1q = q(torch.relu(cat @ q(wf0, dt).T + bf0), dt)
    f2q = q(torch.relu(f1q @ q(wf1, dt).T + bf1), dt)
    o = f2q @ q(wf2, dt).T + bf2
    b_eff = beta.abs() + BETA_MIN
    alpha = torch.sigmoid(-sdf * b_eff)
    out = {"feature": e + o, "sdf": sdf, "alpha": alpha}
    if g_feature is None:
        return out
    gs = float(grad_scale) if grad_scale > 0 else 1.0
    d_o = g_feature * gs
    d_oq = q(d_o, dt)
    out["g_feat_w2"], out["g_feat_b2"] = (d_oq.T @ f2q) / gs, d_oq.sum(0) / gs
    d_f2q = q((d_oq @ q(wf2, dt)) * (f2q > 0), dt)
    out["g_feat_w1"], out["g_feat_b1"] = (d_f2q.T @ f1q) / gs, d_f2q.sum(0) / gs
    d_f1q = q((d_f2q @ q(wf1, dt)) * (f1q > 0), dt)
    out["g_feat_w0"], out["g_feat_b0"] = (d_f1q.T @ cat) / gs, d_f1q.sum(0) / gs
    d_e = (d_o + d_f1q @ q(wf0, dt)[:, :C]) / gs  # residual; the SH columns carry no gradient
    dsig = g_alpha * alpha * (1 - alpha)
    d_sdf = dsig * (-b_eff)
    out["g_beta"] = (dsig * (-sdf)).sum().reshape(1) * torch.sign(beta).where(beta != 0, torch.ones_like(beta))
    d_eq = q(d_e * gs, dt)
    g_w1 = torch.empty_like(wg1)
    g_w1[1:] = (d_eq.T @ h1q) / gs
    g_w1[0] = d_sdf @ h1
    out["g_geo_w1"] = g_w1
    out["g_geo_b1"] = torch.cat([d_sdf.sum().reshape(1), d_eq.sum(0) / gs])
    d_h1 = d_eq @ q(wg1[1:], dt) + (d_sdf * gs)[:, None] * wg1[0][None, :]
    d_h1q = q(d_h1 * (h1 > 0), dt)
    out["g_geo_w0"], out["g_geo_b0"] = (d_h1q.T @ x0) / gs, d_h1q.sum(0) / gs
    out["g_feats"] = (d_h1q @ q(wg0, dt)) / gs
    return out
