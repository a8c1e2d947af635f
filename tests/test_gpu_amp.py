"""Dynamic loss scale with found-inf guard on the device (step.GradScalerState, nr_amp_* / `skip` of the Adam kernels):
the semantics of the reference's torch.cuda.amp.GradScaler use (engine/trainer.py:200,572-594; engine/optimizers.py:154-166)
without a host read --
  * nr_amp_update against torch's own update rule (torch._amp_update_scale_, what GradScaler.update launches);
  * the Adam kernels' skip (parameters and moments untouched, gradient cleared) and the schedule kernel not counting a
    skipped step;
  * the fused fp16 step with an INJECTED overflow: nothing non-finite reaches a parameter or a moment, the scale backs off,
    the next step trains as if the skipped one had not happened;
  * the 16-bit CNN chain's scale / unscale launch."""
import math

import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_amp_update_follows_gradscaler_update():
    from neuradar_amd import _lib, ops
    from neuradar_amd.step import GradScalerState

    amp = GradScalerState(DEV, init_scale=1024.0, growth_interval=3)
    amp.n_groups = 3
    scale, tracker = torch.full((1,), 1024.0, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    seq = [(), (), (), (1,), (), (0, 2), (), (), (), (), (2,), ()]  # groups that found an inf / NaN in each step
    for k, found in enumerate(seq):
        for g in found:
            amp.found(g).fill_(1.0)
        amp.update()
        torch._amp_update_scale_(scale, tracker, torch.full((1,), float(bool(found)), device=DEV), 2.0, 0.5, 3)
        buf = amp.buf.cpu()
        assert float(buf[_lib.NR_AMP_SCALE]) == float(scale) and int(buf[_lib.NR_AMP_GROWTH_TRACKER]) == int(tracker), (k, buf, scale, tracker)
        assert float(buf[_lib.NR_AMP_INV_SCALE]) == 1.0 / float(scale)
        assert float(buf[_lib.NR_AMP_SKIPPED_PREV]) == float(bool(found))
        prev = [int(buf[_lib.NR_AMP_FOUND_PREV + g]) for g in range(3)]
        assert prev == [int(g in found) for g in range(3)] and float(buf[_lib.NR_AMP_FOUND:_lib.NR_AMP_FOUND + 8].abs().sum()) == 0.0
    assert amp.skipped_steps() == sum(1 for f in seq if f)
    sd = amp.state_dict()
    other = GradScalerState(DEV)
    other.load_state_dict(sd)
    assert torch.equal(other.buf, amp.buf) and other.growth_interval == 3


@pytest.mark.parametrize("marked", [False, True])
@pytest.mark.parametrize("n", [4096, 70001 * 4])
def test_adam_kernels_skip_on_the_flag(marked, n):
    """skip != 0: parameters, moments (and the `seen` bytes) untouched, the gradient -- incl. its inf / NaN entries -- cleared;
    skip == 0: bit-identical to the launch without a flag."""
    from neuradar_amd import ops

    gen = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=gen).to(DEV)
    g0 = torch.randn(n, generator=gen).to(DEV) * (torch.rand(n, generator=gen).to(DEV) < 0.3)
    g0.view(-1, 4)[::3] = 0.0
    m0, v0 = 0.01 * torch.randn(n, generator=gen).to(DEV), 0.001 * torch.rand(n, generator=gen).to(DEV)
    nz = g0.view(-1, 4).ne(0).any(dim=1)
    seen0 = (nz | (torch.rand(n // 4, generator=gen).to(DEV) < 0.2)).to(torch.uint8)
    m0.view(-1, 4)[seen0 == 0] = 0.0  # (never-seen groups have zero moments by construction)
    v0.view(-1, 4)[seen0 == 0] = 0.0

    def run(flag, poison):
        p, g, m, v, seen = p0.clone(), g0.clone(), m0.clone(), v0.clone(), seen0.clone()
        if poison:
            idx = torch.nonzero(nz)[:5, 0] * 4
            g[idx[0]], g[idx[1] + 1], g[idx[2] + 3] = float("inf"), float("nan"), float("-inf")
        ops.adam_step(p, g, m, v, 1e-2, 3, eps=1e-15, seen_grad=seen, marked=marked,
                      skip=None if flag is None else torch.full((1,), flag, device=DEV))
        return p, g, m, v, seen

    ref = run(None, False)
    same = run(0.0, False)
    for a, b, what in zip(ref, same, ("param", "grad", "exp_avg", "exp_avg_sq", "seen")):
        assert torch.equal(a, b), what
    p, g, m, v, seen = run(1.0, True)
    assert torch.equal(p, p0) and torch.equal(m, m0) and torch.equal(v, v0) and torch.equal(seen, seen0)
    assert float(g.abs().max()) == 0.0 and bool(torch.isfinite(g).all())
    assert not torch.equal(ref[0], p0)


def test_adam_hyper_does_not_count_a_skipped_step():
    """Scheduler steps stop while ANY optimizer skipped (trainer.py:590-594), the bias-correction count of an optimizer while
    ITS update was skipped (torch.optim.Adam's `step`)."""
    from neuradar_amd import _lib, ops
    from neuradar_amd.step import GradScalerState

    amp = GradScalerState(DEV, init_scale=256.0)
    amp.n_groups = 2
    lib, p = _lib.lib(), ops._p
    st = [torch.zeros(2, device=DEV) for _ in range(2)]
    hy = [torch.zeros(3, device=DEV) for _ in range(2)]
    sched, upd = 0, [0, 0]  # host model: steps the schedulers / each optimizer have really taken
    found_seq = [(), (1,), (), (0, 1), (), ()]
    for found in found_seq:
        for g in range(2):
            ops.check(lib.nr_adam_hyper(p(st[g]), p(hy[g]), 1e-2, 1e-3, 500, 20001, 0.9, 0.999, p(amp.buf), g, ops._stream()), "hyper")
            lr = 1e-8 + (1e-2 - 1e-8) * math.sin(0.5 * math.pi * sched / 500)
            k = upd[g] + 1
            want = torch.tensor([lr, 1 - 0.9 ** k, math.sqrt(1 - 0.999 ** k)])
            torch.testing.assert_close(hy[g].cpu(), want.float(), rtol=1e-5, atol=1e-12)
        for g in found:
            amp.found(g).fill_(1.0)
        amp.update()
        if not found:
            sched += 1
        for g in range(2):
            if g not in found:
                upd[g] += 1
    assert sched == 4 and upd == [5, 4]


def test_nonfinite_check_and_unscale_add():
    from neuradar_amd import _lib, ops

    lib, p = _lib.lib(), ops._p
    for n in (8, 1000, 4 * 100003 + 3):
        x = torch.randn(n, device=DEV)
        flag = torch.zeros(1, device=DEV)
        ops.check(lib.nr_nonfinite_check(p(x), n, p(flag), ops._stream()), "check")
        assert float(flag) == 0.0
        for bad, at in ((float("inf"), 0), (float("nan"), n - 1), (float("-inf"), n // 2)):
            y = x.clone()
            y[at] = bad
            flag.zero_()
            ops.check(lib.nr_nonfinite_check(p(y), n, p(flag), ops._stream()), "check")
            assert float(flag) == 1.0, (n, bad, at)
    for dt, code in ((torch.float16, 2), (torch.bfloat16, 1)):
        for n in (16, 4 * 5003 + 2):
            dst0 = torch.randn(n, device=DEV)
            src0 = (100.0 * torch.randn(n, device=DEV)).to(dt)
            inv = torch.full((1,), 1.0 / 512.0, device=DEV)
            dst, src, flag = dst0.clone(), src0.clone(), torch.zeros(1, device=DEV)
            ops.check(lib.nr_unscale_add_16(p(dst), p(src), n, code, p(inv), p(flag), ops._stream()), "unscale")
            assert torch.equal(dst, (dst0 + src0.float()) * inv) and float(src.float().abs().max()) == 0.0 and float(flag) == 0.0
            dst, src = dst0.clone(), src0.clone()
            src[n - 1] = float("inf")
            ops.check(lib.nr_unscale_add_16(p(dst), p(src), n, code, None, p(flag), ops._stream()), "unscale")
            assert float(flag) == 1.0 and torch.equal(dst[:-1], dst0[:-1] + src0[:-1].float())


def _fp16_step(n_actors=0):
    from test_gpu_parity import build_hot_path
    from neuradar_amd.fused_step import FusedTrainStep
    from neuradar_amd.step import FlatAdam, GradScalerState

    g = load_golden("pipeline")
    torch.manual_seed(0)
    model = build_hot_path(g).train()
    model.field.config.mlp_dtype = "float16"
    model.field.config.mlp_grad_scale = 1024.0
    groups = model.get_param_groups()
    unused = list(model.proposal_fields[0].parameters())
    opts = [FlatAdam(groups["hashgrids"], lr=1e-2, eps=1e-15, warmup_steps=0, skip=unused),
            FlatAdam(groups["fields"], lr=1e-2, eps=1e-15, weight_decay=1e-7, adamw=True, warmup_steps=0, skip=unused)]
    d = lambda k: g[k].to(DEV)  # noqa: E731
    B = g["origins"].shape[0]
    fused = FusedTrainStep(model, B)
    amp = GradScalerState(DEV, init_scale=1024.0, growth_interval=2).attach(opts)
    fused.set_grad_scaler(amp)
    args = (d("origins"), d("directions"), d("pixel_area")[:, 0].contiguous(), d("fars")[:, 0].contiguous())
    rnd = (d("t_rand"), d("jitter1")[:, 0].contiguous(), d("jitter2")[:, 0].contiguous())
    tf, td = d("target_features"), d("target_depth")[:, 0].contiguous()
    return model, opts, fused, amp, args, rnd, tf, td


def _snapshot(model, opts):
    s = {n: p.detach().clone() for n, p in model.named_parameters()}
    for i, o in enumerate(opts):
        for j, (m, v) in enumerate(o.state):
            s[f"opt{i}.m{j}"], s[f"opt{i}.v{j}"] = m.clone(), v.clone()
        for j, sn in enumerate(o.seen):
            if sn is not None:
                s[f"opt{i}.seen{j}"] = sn.clone()
    return s


def test_fused_fp16_step_skips_an_injected_overflow():
    """A target of 1e30 makes d loss / d features overflow the fp16 operands of the field backward (x scale): the step must
    leave every parameter and moment exactly as it was, clear the gradients, halve the scale -- and the following steps must
    be bit-identical to those of a run that never saw the bad batch (the schedule does not count the skipped step)."""
    from neuradar_amd import _lib

    def run(poison_at):
        model, opts, fused, amp, args, rnd, tf, td = _fp16_step()
        scales, snaps = [], [_snapshot(model, opts)]
        for k in range(4 + (poison_at is not None)):
            bad = poison_at is not None and k == poison_at
            before = _snapshot(model, opts)
            loss = fused.forward_backward(*args, torch.full_like(tf, 1e30) if bad else tf, td, *rnd, optimizers=opts)
            torch.cuda.synchronize()
            after = _snapshot(model, opts)
            if bad:
                assert not math.isfinite(float(loss.sum())) or float(loss.sum()) > 1e30
                for key in before:
                    if "seen" in key:  # (a marking scatter may set bytes of groups whose moments then stay zero: harmless)
                        continue
                    assert torch.equal(before[key], after[key]), f"a skipped step changed {key}"
                assert float(amp.buf[_lib.NR_AMP_SKIPPED_PREV]) == 1.0
            else:
                assert any(not torch.equal(before[k_], after[k_]) for k_ in before)
                snaps.append(after)
            for n_, p_ in model.named_parameters():
                if p_.grad is not None:
                    assert float(p_.grad.abs().max()) == 0.0 and bool(torch.isfinite(p_.grad).all()), f"gradient of {n_} not cleared"
            for key, val in after.items():
                assert bool(torch.isfinite(val.float()).all()), key
            scales.append(amp.get_scale())
        state = [o.step_t.cpu().tolist() for o in opts] + [o.hyper.cpu().tolist() for o in opts]
        return scales, snaps, amp.skipped_steps(), state

    clean_scales, clean, skipped0, clean_state = run(None)
    assert skipped0 == 0 and clean_scales == [1024.0, 2048.0, 2048.0, 4096.0]  # growth_interval = 2
    scales, snaps, skipped, poisoned_state = run(1)
    assert skipped == 1
    assert scales == [1024.0, 512.0, 512.0, 1024.0, 1024.0], scales  # backoff at the poisoned step, growth after 2 clean ones
    # The skipped step is not counted: after 4 performed updates both runs' schedulers and optimizers stand at step 4 and hold
    # bit-identical (lr, bias-correction) triples -- had it been counted, the bias corrections alone would differ
    # (1 - 0.9^5 vs 1 - 0.9^4).  The parameters themselves are only loosely comparable: the two runs use different scales
    # (512 vs 2048 at the same step), which decides which of the tiny gradients flush in fp16, and Adam turns "zero or 1e-7"
    # into "no update or a full-size one" for that entry (measured: 16 % of the update's norm on the main table).
    assert clean_state == poisoned_state, (clean_state, poisoned_state)
    assert clean_state[0] == [4.0, 4.0]
    start = clean[0]
    for a, b in zip(clean[1:], snaps[1:]):
        for key in a:
            if "seen" in key:
                continue
            moved = float((a[key].double() - start[key].double()).norm())
            if moved == 0.0:
                assert torch.equal(a[key], b[key]), key
                continue
            err = float((a[key].double() - b[key].double()).norm()) / moved
            assert err < 0.5, f"{key}: the run with a skipped step differs from the clean one by {err:.2e} of the update"


def test_fused_fp16_step_with_grad_scaler_replays_as_a_graph():
    """The guard has no host read: the step with it is captured and replayed; a poisoned replay is skipped like an eager one."""
    model, opts, fused, amp, args, rnd, tf, td = _fp16_step()
    tf_buf = tf.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fused.forward_backward(*args, tf_buf, td, *rnd, optimizers=opts)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fused.forward_backward(*args, tf_buf, td, *rnd, optimizers=opts)
        g.replay()
        torch.cuda.synchronize()
        before = _snapshot(model, opts)
        s0 = amp.get_scale()
        tf_buf.fill_(1e30)
        g.replay()
        torch.cuda.synchronize()
        after = _snapshot(model, opts)
        for key in before:
            if "seen" in key:  # (the marking scatter may set bytes of groups whose moments then stay zero: harmless)
                continue
            assert torch.equal(before[key], after[key]), f"a skipped replay changed {key}"
        assert amp.get_scale() == 0.5 * s0 and amp.skipped_steps() == 1
        tf_buf.copy_(tf)
        g.replay()
        torch.cuda.synchronize()
        again = _snapshot(model, opts)
        assert any(not torch.equal(after[k_], again[k_]) for k_ in after)
        for key, val in again.items():
            assert bool(torch.isfinite(val.float()).all()), key
    torch.cuda.current_stream().wait_stream(side)
