"""Oracle: multiresolution hash grid, torch semantics (NOT tcnn's).  Test infrastructure only.

Restates field_components/encodings.py:348-352 (level scalings), :380-384 (table init),
:406-423 (spatial hash) and :425-466 (trilinear gather).
"""
import numpy as np
import torch

# encodings.py:418 -- per-axis multipliers of the spatial hash
PRIMES = (1, 2654435761, 805459861)


def level_scalings(num_levels: int, min_res: int, max_res: int) -> torch.Tensor:
    """Per-level grid scale.  encodings.py:348-350.

    The reference evaluates floor(min_res * g**l) with `g` a numpy float64 and `l` a torch int64
    arange; torch promotes that to float32, which is why the top NeuRadar level is 8191 and not 8192
    (SURVEY.md Appendix B).  The same expression is used here so the dtype path is identical.
    """
    levels = torch.arange(num_levels)
    growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1.0
    return torch.floor(min_res * growth**levels)


def init_table(num_levels: int, log2_hashmap_size: int, features_per_level: int, scale: float = 1e-3,
               generator: torch.Generator = None) -> torch.Tensor:
    """Uniform(-scale, scale) table of shape [L*T, F].  encodings.py:380-384."""
    t = torch.rand((2**log2_hashmap_size * num_levels, features_per_level), generator=generator) * 2 - 1
    return t * scale


def hash_slots(ix: torch.Tensor, iy: torch.Tensor, iz: torch.Tensor, table_size: int) -> torch.Tensor:
    """Spatial hash of integer corner coordinates -> slot inside ONE level.  encodings.py:406-421.

    int32 coordinates are widened to int64, multiplied by PRIMES, xor-ed and reduced mod T
    (python-style, non-negative).  T is a power of two, so only the low bits matter: uint32
    wrap-around arithmetic gives the same slot.
    """
    h = (ix.to(torch.int64) * PRIMES[0]) ^ (iy.to(torch.int64) * PRIMES[1]) ^ (iz.to(torch.int64) * PRIMES[2])
    return h % table_size


def encode(x: torch.Tensor, table: torch.Tensor, scalings: torch.Tensor, table_size: int) -> torch.Tensor:
    """x [N,3] in [0,1] -> features [N, L*F].  encodings.py:425-466.

    Every level is hashed (no dense coarse levels).  The interpolation weight `frac = p - floor(p)`
    multiplies the CEIL corner, (1-frac) the FLOOR corner; axes are reduced x, then y, then z.
    """
    num_levels = scalings.numel()
    p = x[:, None, :] * scalings.to(x).view(num_levels, 1)  # [N,L,3]
    hi = torch.ceil(p).to(torch.int32)
    lo = torch.floor(p).to(torch.int32)
    frac = p - lo
    base = torch.arange(num_levels, device=x.device) * table_size  # encodings.py:352,422

    def corner(sel_x, sel_y, sel_z):
        return table[hash_slots(sel_x[..., 0], sel_y[..., 1], sel_z[..., 2], table_size) + base]  # [N,L,F]

    wx, wy, wz = frac[..., 0:1], frac[..., 1:2], frac[..., 2:3]

    def along_x(sel_y, sel_z):
        return corner(hi, sel_y, sel_z) * wx + corner(lo, sel_y, sel_z) * (1 - wx)

    def along_xy(sel_z):
        return along_x(hi, sel_z) * wy + along_x(lo, sel_z) * (1 - wy)

    out = along_xy(hi) * wz + along_xy(lo) * (1 - wz)  # [N,L,F]
    return out.flatten(-2, -1)
