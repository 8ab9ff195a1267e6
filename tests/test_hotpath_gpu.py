"""The hot path at the sizes where bench.py runs it: the LDS-DMA ring kernels, the fused dense-block weight gradient, the persistent Winograd
F(4x4,3x3) kernel and the pool / loss fusions are only selected from ~2^17 pixels per launch upwards, which the oracle-parity tests (small
tiles) never reach.  This test takes one optimize_parameters step of the BASELINE configs[1] workload (ESRGAN 4x 23 blocks, L1 + fp32
perceptual, bf16) at batch 2, 256 -> 1024, once with the default kernel selection and once with every one of those selections switched off
(register-staged convs and weight gradients, direct fp32 VGG convs, separate pool kernels) -- the configuration the small-tile tests pin to
the oracle -- and compares the loss terms and the weights after the step.

Tolerances: both runs compute the same bf16-operand / fp32-accumulate arithmetic in other summation orders (ring vs register tiling, 32 vs
256 partial slots) and the Winograd head adds ~2e-6 per VGG layer.  Forward: losses agree to ~1e-6 (asserted 2e-5).  Backward: every
16-bit gradient tensor on the way down is re-rounded, so an fp32 difference in the last place flips some of those roundings and the
generator's parameter gradient (read back as AdamW's first moment) agrees to the measured figure below, not to fp32 precision."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SWITCHES = {"TRAINNER_VGG_WINOGRAD": "0", "TRAINNER_NO_POOL_FUSION": "1", "TRAINNER_NO_STEM": "1"}     # host-side selections (perceptual_loss.py)
TUNE_PLAIN = dict(conv_impl=1, wgrad_impl=1, wgrad_multi=0)                                            # library-side selections (tr_tune_set)


def _one_step(monkeypatch, plain: bool):
    import bench
    from trainner_redux_amd.models import build_model

    for k, v in SWITCHES.items():
        if plain:
            monkeypatch.setenv(k, v)
        else:
            monkeypatch.delenv(k, raising=False)
    argv = sys.argv
    sys.argv = ["bench.py", "--batch", "2", "--lq", "256"]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    torch.manual_seed(1024)
    model = build_model(bench.make_opt(args))
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(7)
    gt = torch.rand(2, 3, 1024, 1024, generator=g).to(dev)
    lq = torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", antialias=True).clamp(0, 1)
    from trainner_redux_amd import ops

    model.feed_data({"lq": lq, "gt": gt})
    ops.tune_set("ring_launches", 0)
    with ops.tuned(**(TUNE_PLAIN if plain else {})):
        model.optimize_parameters(1, 0, True)
    assert (ops.tune_get("ring_launches") == 0) == plain      # the default selection really is the ring kernel at this size
    log = {k: float(v) for k, v in model.get_current_log().items()}
    torch.cuda.synchronize()
    # AdamW's first moment after step 1 is (1 - beta1) * gradient: the gradient itself has been zeroed by the fused optimizer sweep.  (The
    # weights are no measure: the first Adam step moves every weight by lr * sign(g), so rounding-level differences of near-zero
    # gradients show up as 2 * lr.)
    opt = model.optimizers[0]
    grads = torch.cat([opt.state[p]["exp_avg"].detach().float().flatten() for p in model.net_g.parameters() if p.requires_grad]).cpu()
    return log, grads


def test_fast_kernel_selection_matches_the_plain_paths(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    log_f, w_f = _one_step(monkeypatch, plain=False)
    log_p, w_p = _one_step(monkeypatch, plain=True)
    assert set(log_f) == set(log_p) and "l_g_total" in log_p and any(k.startswith("l_g_perceptual") for k in log_p)
    for k in log_p:
        if k.startswith("l_g_"):
            assert abs(log_f[k] - log_p[k]) <= 2e-5 * abs(log_p[k]), (k, log_f[k], log_p[k])
    rel = ((w_f - w_p).double().norm() / w_p.double().norm()).item()
    cos = (torch.dot(w_f.double(), w_p.double()) / (w_f.double().norm() * w_p.double().norm())).item()
    print(f"gradient rel-L2 {rel:.3e}, cosine {cos:.8f}, losses {log_f} vs {log_p}")
    assert rel < 5e-4 and cos > 0.999999, (rel, cos)        # measured 5.0e-5 / 1.00000000


@pytest.mark.parametrize("flags,batch,lq", [("", 16, 256), ("--arch swinir_m", 8, 64), ("--arch span --compute f16", 64, 64), ("--arch mosr --compute f16", 64, 64),
                                            ("--compute f16", 16, 256), ("--losses l1+perceptualfp16", 8, 256), ("--arch swinir_s", 8, 64), ("--arch span_s --compute f16", 64, 64),
                                            ("--arch mosr_t --compute f16", 64, 64), ("--arch mosr", 64, 64), ("--arch span", 64, 64), ("--arch swinir_m --compute f16", 8, 64)],
                         ids=["configs1_esrgan", "configs3_swinir_m", "configs4_span_f16", "configs4_mosr_f16", "esrgan_f16", "esrgan_perceptualfp16", "swinir_s", "span_s_f16",
                              "mosr_t_f16", "mosr_bf16", "span_bf16", "swinir_m_f16"])
def test_bench_size_batch_is_the_union_of_its_halves(flags, batch, lq):
    """Every bench workload at its FULL size (configs[1]: batch 16, 256 -> 1024, 23 blocks, L1 + fp32 perceptual, bf16; configs[3] SwinIR-M batch 8; configs[4]
    SPAN / MoSR 2x fp16 batch 64) -- sizes no oracle run reaches -- through a size-independent property: samples do not interact, so (a) the generator's
    output for the batch is, bit for bit, the outputs of its two halves, whatever tiles the persistent kernels deal to which workgroup, and (b) the
    parameter gradient of the full-batch step is the mean of the two half-batch steps' gradients (d loss / d output differs by an exact factor 2, which
    16-bit rounding commutes with; only the fp32 summation order over samples differs) and its loss terms are the means of theirs.
    (SwinIR's stochastic depth is switched off here: its keep masks are drawn per call.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bench
    from trainner_redux_amd.models import build_model

    for k in SWITCHES:
        os.environ.pop(k, None)
    argv = sys.argv
    sys.argv = ["bench.py", *flags.split(), "--batch", str(batch), "--lq", str(lq)]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    torch.manual_seed(1024)
    model = build_model(bench.make_opt(args))
    net = model.net_g
    if hasattr(net, "drop_path_rate"):
        net.drop_path_rate = 0.0
    sc, h = args.scale, batch // 2
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(11)
    gt = torch.rand(batch, 3, sc * lq, sc * lq, generator=g).to(dev)
    lqt = torch.nn.functional.interpolate(gt, scale_factor=1.0 / sc, mode="bicubic", antialias=True).clamp(0, 1)
    _, garr = net.flat_parameters()

    def grads_of(parts):
        garr.zero_()
        model.reset_current_log()
        outs = []
        for sl in parts:
            model.feed_data({"lq": lqt[sl], "gt": gt[sl]})
            model.optimize_parameters(1, 0, False)          # accumulate only: the weights stay where they are
            outs.append(model.output.clone())
        log = {k: float(v) for k, v in model.get_current_log().items()}
        torch.cuda.synchronize()
        return torch.cat(outs), garr.clone(), log

    out_f, g_f, log_f = grads_of([slice(0, batch)])
    out_h, g_h, log_h = grads_of([slice(0, h), slice(h, batch)])
    assert torch.equal(out_f, out_h), (out_f - out_h).abs().max().item()
    for k in log_f:
        if k.startswith("l_g_"):
            assert abs(log_f[k] - log_h[k]) <= 2e-6 * abs(log_h[k]), (k, log_f[k], log_h[k])
    half = g_h.double() * 0.5
    rel = ((g_f.double() - half).norm() / half.norm()).item()
    print(f"{flags or 'esrgan'} batch {batch} vs {h} + {h}: outputs identical, gradient rel-L2 {rel:.3e}, losses {log_f}")
    assert rel < 2e-5, rel


@pytest.mark.parametrize("flags,batch,lq,bound", [("--arch mosr --compute f16", 64, 64, None), ("--arch span --compute f16", 64, 64, None), ("--arch swinir_m", 8, 64, None),
                                                  ("--arch mosr", 64, 64, None)],
                         ids=["mosr_f16", "span_f16", "swinir_m", "mosr_bf16"])
def test_default_kernel_selection_matches_the_plain_kernels_at_bench_size(flags, batch, lq, bound):
    """Differential check of the kernel families on the transformer / gated archs at their bench sizes: one accumulate-only step with the default selection
    (LDS-DMA ring convs and weight gradients, LDS-DMA token GEMM with straight-line epilogues, 16-byte weight-gradient reduction) against the same step on the
    register-staged kernels with the generic epilogues -- the families the small-tile oracle tests pin.  Same arithmetic in another summation order:
    outputs agree to 16-bit rounding flips, parameter gradients to the cosine / rel-L2 printed below."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bench
    from trainner_redux_amd import ops
    from trainner_redux_amd.models import build_model

    argv = sys.argv
    sys.argv = ["bench.py", *flags.split(), "--batch", str(batch), "--lq", str(lq)]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    torch.manual_seed(1024)
    model = build_model(bench.make_opt(args))
    net = model.net_g
    if hasattr(net, "drop_path_rate"):
        net.drop_path_rate = 0.0
    sc = args.scale
    g = torch.Generator().manual_seed(11)
    gt = torch.rand(batch, 3, sc * lq, sc * lq, generator=g).cuda()
    lqt = torch.nn.functional.interpolate(gt, scale_factor=1.0 / sc, mode="bicubic", antialias=True).clamp(0, 1)
    _, garr = net.flat_parameters()

    def run(**tune):
        garr.zero_()
        model.reset_current_log()
        with ops.tuned(**tune):
            model.feed_data({"lq": lqt, "gt": gt})
            model.optimize_parameters(1, 0, False)
        torch.cuda.synchronize()
        return model.output.float().clone(), garr.double().clone(), float(model.get_current_log()["l_g_total"])

    out_d, g_d, l_d = run()
    out_p, g_p, l_p = run(conv_impl=1, wgrad_impl=1, wgrad_multi=0, linear_impl=1, linear_epilogue=1, wgrad_reduce=1)
    e_out = ((out_d - out_p).double().norm() / out_p.double().norm()).item()
    rel = ((g_d - g_p).norm() / g_p.norm()).item()
    cos = (torch.dot(g_d, g_p) / (g_d.norm() * g_p.norm())).item()
    print(f"{flags}: output rel-L2 {e_out:.3e}, loss {l_d:.7f} vs {l_p:.7f}, gradient rel-L2 {rel:.3e}, cosine {cos:.7f}")
    assert e_out < 2e-3 and abs(l_d - l_p) <= 1e-4 * abs(l_p) and cos > 0.9995 and rel < 3e-2, (e_out, l_d, l_p, rel, cos)
