"""Shared test helpers: golden loading and oracle parameter construction."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: torch.from_numpy(z[k]) if z[k].ndim > 0 else z[k].item() for k in z.files}


def t(x, requires_grad=False):
    x = x.clone()
    x.requires_grad_(requires_grad)
    return x


def oracle_field_params(g, prefix="", log2t_key="log2t", requires_grad=False, static_scale=100.0):
    from oracle.field import FieldParams, GridParams

    def layers(tag):
        out, i = [], 0
        while f"{prefix}{tag}_w{i}" in g:
            out.append((t(g[f"{prefix}{tag}_w{i}"], requires_grad), t(g[f"{prefix}{tag}_b{i}"], requires_grad)))
            i += 1
        return out

    grid = GridParams(t(g[f"{prefix}table"], requires_grad), g[f"{prefix}scalings"], int(g[log2t_key]))
    return FieldParams(grid, layers("geo"), layers("feat"), t(g[f"{prefix}beta"], requires_grad), static_scale)


def oracle_prop_params(g, prefix="prop_", log2t_key="prop_log2t", requires_grad=False, static_scale=100.0):
    from oracle.field import GridParams, ProposalParams

    grid = GridParams(t(g[f"{prefix}table"], requires_grad), g[f"{prefix}scalings"], int(g[log2t_key]))
    return ProposalParams(grid, t(g[f"{prefix}decoder"], requires_grad), static_scale)


def assert_close(actual, expected, rtol=1e-4, atol_scale=1e-5, what=""):
    """rel tolerance `rtol` with an absolute floor of atol_scale * max|expected| (the north-star's
    "within 1e-4 rel" is meaningless for entries that are ~0 against O(1) neighbours)."""
    if not isinstance(expected, torch.Tensor):
        expected = torch.as_tensor(expected, dtype=actual.dtype).reshape(actual.shape)
    expected = expected.to(actual.dtype)
    scale = float(expected.abs().max()) if expected.numel() else 1.0
    torch.testing.assert_close(actual, expected, rtol=rtol, atol=atol_scale * max(scale, 1e-30), msg=lambda m: f"{what}: {m}")


def assert_close_but_few(actual, expected, rtol, atol_scale, max_outlier_frac, what=""):
    """assert_close for results that contain DISCRETE decisions (ReLU masks, round-to-nearest ties) which a change in
    accumulation order may flip for isolated elements: every element within tolerance except at most
    `max_outlier_frac` of them, and those few must still be finite."""
    expected = expected.to(actual.dtype).reshape(actual.shape)
    scale = float(expected.abs().max()) if expected.numel() else 1.0
    bad = (actual - expected).abs() > (atol_scale * max(scale, 1e-30) + rtol * expected.abs())
    frac = float(bad.float().mean()) if bad.numel() else 0.0
    assert bool(torch.isfinite(actual).all()), f"{what}: non-finite values"
    assert frac <= max_outlier_frac, (f"{what}: {int(bad.sum())} of {bad.numel()} elements ({frac:.2e}) outside rtol={rtol:.1e} / "
                                      f"atol={atol_scale:.1e}*scale; worst |d| = {float((actual - expected).abs().max()):.3e}")


def run_child(test_file: str, func: str, timeout: float = 900.0, env=None, **kwargs):
    """Run `func(**kwargs)` of `test_file` in a child python process (tests/child_main.py) and return what it returned.  The legs
    that once took the whole pytest process down with them (a GPU fault is a SIGABRT from the ROCr runtime) run this way: the
    child's death -- any non-zero exit, a signal, the timeout -- is an ordinary test FAILURE carrying the tail of its output, and
    every other test of the session still reports.  (A child process, never an exec: this process has initialised the GPU.)"""
    import pytest

    here = os.path.dirname(os.path.abspath(__file__))
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "result.pt")
        cmd = [sys.executable, os.path.join(here, "child_main.py"), os.path.abspath(test_file), func, json.dumps(kwargs), out]
        child_env = dict(os.environ)
        child_env.update(env or {})
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=os.path.dirname(here), env=child_env)
        except subprocess.TimeoutExpired as e:
            pytest.fail(f"child {func}({kwargs}) did not finish within {timeout:.0f} s\n{(e.stderr or '')[-3000:]}")
        if r.stdout:
            print(r.stdout[-6000:])
        if r.returncode != 0 or not os.path.exists(out):
            sig = f" (signal {-r.returncode})" if r.returncode < 0 else ""
            pytest.fail(f"child {func}({kwargs}) exited with {r.returncode}{sig}\n--- tail of its stderr ---\n{r.stderr[-4000:]}")
        return torch.load(out, weights_only=False)
