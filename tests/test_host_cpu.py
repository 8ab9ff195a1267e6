"""CPU suite (no GPU): host logic of the package and the C-ABI surface (load + symbols only, no compute)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as ge

    ge.build()
    return True


def test_c_abi_exports_every_declared_symbol(built):
    from trainner_redux_amd import _lib

    header = open(os.path.join(ROOT, "include", "trainner_hip.h")).read()
    declared = set(re.findall(r"\b(tr_[a-z0-9_]+)\s*\(", header))
    declared -= {"tr_view"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.lib()
    assert lib.tr_abi_version() == _lib.ABI_VERSION
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.tr_conv3x3_packed_bytes(64, 32, _lib.BF16) == 2 * 9 * 32 * 64
    assert lib.tr_conv3x3_packed_bytes(192, 64, _lib.BF16) == 6 * 9 * 64 * 64
    assert lib.tr_conv3x3_packed_bytes(192, 64, _lib.F32) == 12 * 9 * 64 * 64
    assert lib.tr_wgrad3x3_workspace_bytes(2, 64, 64, 192, 64) > 0


def test_c_abi_rejects_bad_arguments_without_a_gpu(built):
    """Argument validation runs before any launch, so it is testable here; messages avoid the words the reference's
    loop treats as out-of-memory (train.py:757-849)."""
    from trainner_redux_amd import _lib

    lib = _lib.lib()
    d = _lib.Conv3x3Desc()
    assert lib.tr_conv3x3(None, ctypes.byref(d)) == -1
    msg = lib.tr_last_error().decode()
    assert "bad dims" in msg and "allocate" not in msg and "CUDA" not in msg
    d.n, d.h, d.w, d.cin, d.cout, d.compute = 1, 8, 8, 12, 32, _lib.BF16
    buf = (ctypes.c_char * 64)()
    d.inp = _lib.View(ctypes.addressof(buf), _lib.BF16, 12, 0, 12)
    d.out = _lib.View(ctypes.addressof(buf), _lib.BF16, 32, 0, 32)
    d.wpacked = ctypes.addressof(buf)
    assert lib.tr_conv3x3(None, ctypes.byref(d)) == -3          # cin not a multiple of 8
    assert lib.tr_pack_conv3x3(None, None, 0, 0) == -1


def test_product_fails_loudly_without_library_or_gpu(built, tmp_path):
    from trainner_redux_amd.archs import build_network

    net = build_network({"type": "esrgan", "scale": 4, "num_blocks": 1})
    with pytest.raises(RuntimeError, match="GPU only"):
        net(torch.rand(1, 3, 8, 8))
    code = ("import sys; sys.path.insert(0, %r); import trainner_redux_amd; from trainner_redux_amd import _lib; "
            "_lib.LIB_PATH = %r; _lib.lib()") % (ROOT, str(tmp_path / "missing.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "no non-HIP execution path" in r.stderr


def test_registry_matches_reference_golden():
    from trainner_redux_amd.utils import registry as R

    gold = torch.load(os.path.join(GOLD, "registry.pt"), weights_only=True)
    r = R.Registry("t")

    @r.register()
    class FooBar:  # noqa: N801
        pass

    @r.register(suffix="traiNNer")
    def baz():
        pass

    assert list(r.keys()) == gold["keys"]
    assert r.get("FOOBAR").__name__ == gold["get_FOOBAR"] and r.get("baz").__name__ == gold["get_baz"]
    assert ("FooBar" in r) == gold["contains_FooBar"]
    with pytest.raises(AssertionError):
        r.register(FooBar)
    with pytest.raises(KeyError):
        r.get("nope")
    assert sorted(k for k in vars(R) if k.endswith("_REGISTRY")) == gold["singletons"]


def test_sampler_matches_reference_golden():
    from trainner_redux_amd.data import EnlargedSampler

    gold = torch.load(os.path.join(GOLD, "sampler.pt"), weights_only=True)
    data = list(range(10))
    for world in (1, 2, 4):
        for rank in range(world):
            s = EnlargedSampler(data, world, rank, ratio=3)
            assert len(s) == gold[f"{world}/{rank}/len"]
            for epoch in (0, 1):
                s.set_epoch(epoch)
                assert list(iter(s)) == gold[f"{world}/{rank}/{epoch}"].tolist()
    # ranks partition the enlarged epoch without overlap of slots
    all_idx = sorted(sum((list(iter(EnlargedSampler(data, 4, r, ratio=2))) for r in range(4)), []))
    assert len(all_idx) == 20


def test_plugin_surface_names_and_checksums():
    from trainner_redux_amd.archs import build_network
    from trainner_redux_amd.losses import build_loss
    from trainner_redux_amd.optimizers import build_optimizer
    from trainner_redux_amd.utils.misc import loss_type_to_label
    from trainner_redux_amd.utils.registry import LOSS_REGISTRY, OPTIMIZER_REGISTRY, SPANDREL_REGISTRY

    assert {"esrgan", "esrgan_lite"} <= set(SPANDREL_REGISTRY.keys())
    assert {"l1loss", "mseloss", "charbonnierloss", "perceptualloss", "perceptualfp16loss"} <= set(LOSS_REGISTRY.keys())
    assert "adamw" in OPTIMIZER_REGISTRY
    n = build_network({"type": "ESRGAN", "scale": 4})     # lookup is case-insensitive like the reference
    assert sum(p.numel() for p in n.parameters()) == 16_697_987
    assert sum(p.numel() for p in build_network({"type": "esrgan", "scale": 2}).parameters()) == 16_703_171
    assert sum(p.numel() for p in build_network({"type": "esrgan_lite", "scale": 4}).parameters()) == 5_021_155
    from trainner_redux_amd.utils.registry import ARCH_REGISTRY
    assert {"mosr", "mosr_t"} <= set(ARCH_REGISTRY.keys())       # archs/mosr_arch.py:23,50 registers them in ARCH_REGISTRY
    for name, sc, cnt in (("mosr", 4, 4_287_600), ("mosr", 2, 4_266_828), ("mosr_t", 4, 609_888), ("mosr_t", 2, 594_300)):   # benchmarks.md:43,226,29,211
        assert sum(p.numel() for p in build_network({"type": name, "scale": sc}).parameters()) == cnt
    with pytest.raises(NotImplementedError):
        build_network({"type": "mosr", "scale": 2, "upsampler": "dysample"})
    assert loss_type_to_label("L1Loss") == "l_g_l1" and loss_type_to_label("PerceptualLoss") == "l_g_perceptual"
    assert loss_type_to_label("multiscaleganloss") == "l_g_gan"
    l1 = build_loss({"type": "l1loss", "loss_weight": 0.5})
    assert l1.loss_weight == 0.5
    with pytest.raises(ValueError, match="Unsupported reduction mode"):
        build_loss({"type": "L1Loss", "loss_weight": 1.0, "reduction": "bogus"})
    with pytest.raises(KeyError):
        build_loss({"type": "nosuchloss", "loss_weight": 1.0})
    opt = build_optimizer(list(n.parameters()), {"type": "AdamW", "lr": 2e-4, "betas": [0.9, 0.99], "weight_decay": 0})
    assert opt.param_groups[0]["lr"] == 2e-4 and opt.param_groups[0]["betas"] == (0.9, 0.99)


def test_state_dict_interchanges_with_oracle_keys_and_arena_survives_moves():
    from oracle import esrgan as OE
    from trainner_redux_amd.archs import build_network

    hp = OE.esrgan_hparams(scale=4, num_blocks=2)
    sd = OE.init_state_dict(hp, 0)
    net = build_network({"type": "esrgan", "scale": 4, "num_blocks": 2})
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    p, g = net.flat_parameters()
    assert p.numel() == g.numel() and p.numel() >= sum(v.numel() for v in sd.values())
    first = next(net.parameters())
    assert first.data_ptr() == p.data_ptr() and first.grad.data_ptr() == g.data_ptr()
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd[k])
    net = net.to(memory_format=torch.channels_last)   # the reference's model_to_device does this (base_model.py:202-244)
    p2, _ = net.flat_parameters()
    assert next(net.parameters()).data_ptr() == p2.data_ptr()
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd[k])
    for q in net.parameters():
        q.grad = None                                   # optimizer.zero_grad(set_to_none=True)
    _, g3 = net.flat_parameters()
    assert next(net.parameters()).grad.data_ptr() == g3.data_ptr()


def test_options_accept_reference_keys_and_reject_unknown(tmp_path):
    from trainner_redux_amd.utils.options import load_options, parse_options_dict

    yml = tmp_path / "o.yml"
    yml.write_text("""
name: t
scale: 4
use_amp: true
amp_bf16: true
use_channels_last: true
num_gpu: auto
manual_seed: 1024
datasets:
  train:
    name: Train
    type: pairedimagedataset
    dataroot_gt: [a]
    dataroot_lq: [b]
    lq_size: 64
    batch_size_per_gpu: 8
    accum_iter: 1
  val_1:
    name: V
    type: pairedimagedataset
network_g:
  type: esrgan
path:
  pretrain_network_g: ~
train:
  ema_decay: 0.999
  ema_power: 0.75
  grad_clip: false
  optim_g: {type: AdamW, lr: !!float 2e-4, weight_decay: 0, betas: [0.9, 0.99]}
  scheduler: {type: MultiStepLR, milestones: [100], gamma: 0.5}
  total_iter: 1000
  warmup_iter: -1
  losses:
    - {type: l1loss, loss_weight: 1.0}
logger:
  print_freq: 100
  save_checkpoint_freq: 1000
  save_checkpoint_format: safetensors
  use_tb_logger: true
val:
  val_enabled: false
  save_img: false
queue_size: 180
""")
    o = load_options(str(yml))
    assert o.scale == 4 and o.use_amp and o.amp_bf16 and o.train.ema_power == 0.75 and o.train.optim_g["lr"] == 2e-4
    assert o.datasets["val_1"].phase == "val" and o.datasets["train"].scale == 4 and isinstance(o.num_gpu, int)
    assert o.logger.save_checkpoint_format == "safetensors" and o.queue_size == 180
    with pytest.raises(ValueError, match="unknown field `bogus_key`"):
        parse_options_dict(dict(name="t", scale=4, num_gpu=0, path={}, bogus_key=1))
    with pytest.raises(ValueError, match="unknown field"):
        parse_options_dict(dict(name="t", scale=4, num_gpu=0, path={}, train=dict(total_iter=1, nope=2)))
    with pytest.raises(ValueError, match="missing required field `scale`"):
        parse_options_dict(dict(name="t", num_gpu=0, path={}))


def test_ema_schedule_matches_oracle():
    from collections import OrderedDict

    from oracle import step as OS
    from trainner_redux_amd.models.ema import EMA

    lin = torch.nn.Linear(3, 2)
    ema = EMA(lin, beta=0.999, update_after_step=0, update_every=1, power=0.75)
    ora = OS.EMAState(OrderedDict(w=lin.weight.data.clone()), beta=0.999, power=0.75)
    for _ in range(6):
        with torch.no_grad():
            lin.weight.add_(torch.randn(2, 3))
        ema.update()
        ora.update(OrderedDict(w=lin.weight.data))
        assert torch.allclose(ema.ema_model.weight, ora.shadow["w"], atol=1e-7)


def test_mfma_kernels_keep_accumulators_in_registers():
    """The compiler's own resource report (written by build.py): no MFMA kernel may use scratch memory -- an accumulator array
    demoted to scratch (a rolled loop indexing it) costs 10x and is invisible to the parity tests."""
    import json

    from trainner_redux_amd import build as B

    B.build(verbose=False)
    res = json.load(open(B.RESOURCES))
    hot = {k: v for k, v in res.items() if any(t in k for t in ("conv3x3_kernel_reg", "linear_kernel", "linear_dma_kernel", "wgrad_kernel", "wgrad_linear_kernel", "window_attn", "dwconv", "winograd_f32"))}
    assert len(hot) >= 20
    # The persistent 3x3 kernel carries three epilogue bodies since the store-wait rework (chunked with / without side inputs, and the
    # row-at-a-time one for ragged views); at its 256-VGPR limit the compiler parks 6-15 dwords of tile-loop state (piece offsets of the
    # next tile) in scratch around them.  That is tolerated up to half of the smallest accumulator array of any tiling (2 MFMA tiles =
    # 32 dwords = 128 bytes per lane): a demoted accumulator array can not hide below it.  Every other kernel family stays at zero.
    def limit(name):
        return 64 if "conv3x3_kernel_reg" in name else 0
    bad = {k: v["scratch"] for k, v in hot.items() if v.get("scratch", 0) > limit(k) or v.get("vgpr_spill", 0) > limit(k) // 4}
    assert not bad, bad
    # Round-2 kernels.  The ring weight-gradient kernel keeps its nine accumulators and everything else in registers.  The persistent Winograd
    # F(4,3) kernel parks tile-loop state around its epilogue (12 dwords; 39 in the masked form, which holds four mask vectors across the
    # exchange barrier) and the generic-epilogue ring conv instances 12: all of it outside the K loops, and well below
    # one Winograd accumulator set (6 x 16 dwords = 384 bytes) or a ring conv accumulator tile (R x NT x 16 dwords >= 256 bytes).
    r2 = {k: v for k, v in res.items() if any(t in k for t in ("conv_ring_kernel", "wgrad_ring_kernel", "winograd43_f32_kernel"))}
    assert len(r2) >= 40
    def limit2(name):
        return 0 if "wgrad_ring" in name else 96
    bad = {k: v["scratch"] for k, v in r2.items() if v.get("scratch", 0) > limit2(k)}
    assert not bad, bad


def test_update_learning_rate_warmup_ramp_and_skipped_step_gating():
    """`SRModel.update_learning_rate` (reference: models/base_model.py:369-390, called once per applied step by train.py:854-857) on the host, next
    to the reference's rule spelled out on torch's own MultiStepLR: every call steps the scheduler of each optimizer whose AMP step was NOT skipped,
    then -- while current_iter < warmup_iter -- sets lr = initial_lr / warmup_iter * current_iter.  (torch's chainable MultiStepLR derives the next lr
    from the group's CURRENT lr, so the warm-up override and the skipped steps both leave traces in later values: the sequences must match call
    for call, not just at the end.)"""
    import torch

    from trainner_redux_amd.models.sr_model import SRModel

    def pair():
        ps = [torch.nn.Parameter(torch.zeros(2))]
        og, od = torch.optim.AdamW(ps, lr=2e-4), torch.optim.AdamW([torch.nn.Parameter(torch.zeros(2))], lr=1e-4)
        return [og, od], [torch.optim.lr_scheduler.MultiStepLR(o, milestones=[3, 6, 9], gamma=0.5) for o in (og, od)]

    class Fake:                      # the attributes the method reads; the trainer itself needs a GPU to construct
        pass

    me = Fake()
    me.optimizers, me.schedulers = pair()
    me._init_lr = [[g["lr"] for g in o.param_groups] for o in me.optimizers]
    me.use_amp = True
    me.optimizers_skipped = [False, False]
    ref_opts, ref_scheds = pair()
    warm = 5
    skipped_at = {4: (True, False), 7: (False, True), 8: (True, True)}      # loss-scaler overflows: G only, D only, both
    seen = []
    for it in range(1, 13):
        sk = skipped_at.get(it, (False, False))
        me.optimizers_skipped = list(sk)
        SRModel.update_learning_rate(me, it, warm)
        for i, sch in enumerate(ref_scheds):                                  # base_model.py:377-379
            if not sk[i]:
                sch.step()
        if it < warm:                                                         # :381-390
            for o in ref_opts:
                for g in o.param_groups:
                    g["lr"] = g["initial_lr"] / warm * it
        got = [o.param_groups[0]["lr"] for o in me.optimizers]
        want = [o.param_groups[0]["lr"] for o in ref_opts]
        assert got == want, (it, got, want)
        seen.append(got[0])
    assert seen[0] == 2e-4 / 5 and seen[1] == 2e-4 / 5 * 2 and seen[3] == 2e-4 / 5 * 4          # the ramp (iterations 1, 2, 4)
    assert [s.last_epoch for s in me.schedulers] == [10, 10]                                   # 12 calls, two skipped per optimizer
    assert SRModel.get_current_learning_rate(me) == [seen[-1]]
    # no warm-up (warmup_iter = -1, the default): plain MultiStepLR
    me2 = Fake()
    me2.optimizers, me2.schedulers = pair()
    me2._init_lr = [[2e-4], [1e-4]]
    me2.use_amp, me2.optimizers_skipped = False, [False, False]
    for it in range(1, 8):
        SRModel.update_learning_rate(me2, it)
    assert abs(me2.optimizers[0].param_groups[0]["lr"] - 2e-4 * 0.25) < 1e-12            # two milestones (3, 6) passed after 7 steps


def test_iterative_loss_wrapper_matches_reference_golden():
    """tests/golden/iterative_schedule.pt (make_golden_twins.py `sched`, from the reference's losses/iterative_loss_wrapper.py): schedule weights on
    iterations 0..44 for eight schedules, `is_active`, the tensor form on scalar / dict-valued modules, the constructor's rejections; and `build_loss`
    splitting the schedule keys off the loss options (losses/__init__.py:79-110)."""
    import warnings

    from torch import nn

    from trainner_redux_amd.losses import build_loss
    from trainner_redux_amd.losses.iterative_loss_wrapper import IterativeLossWrapper, create_iterative_loss

    fix = torch.load(os.path.join(GOLD, "iterative_schedule.pt"), weights_only=True)

    class Scalar(nn.Module):
        def forward(self, a, b):
            return (a - b).abs().mean()

    class Pair(nn.Module):
        returns_dict = True

        def forward(self, a, b):
            return {"p": (a - b).abs().mean(), "q": ((a - b) ** 2).mean()}

    a, b = fix["a"], fix["b"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for row in fix["rows"]:
            w, wp = create_iterative_loss(Scalar(), row["cfg"]), create_iterative_loss(Pair(), row["cfg"])
            assert [w.get_current_weight(i) for i in fix["iters"]] == row["weights"], row["cfg"]      # same float expressions: exact
            assert [w.is_active(i) for i in fix["iters"]] == row["active"]
            assert w.get_schedule_info() == row["info"]
            assert w.loss_weight == row["cfg"]["loss_weight"] and wp.returns_dict and not w.returns_dict
            for k, i in enumerate(fix["fwd_iters"]):
                assert abs(float(w(a, b, current_iter=i)) - row["scalar"][k]) <= 1e-7 * max(1.0, abs(row["scalar"][k]))
                got = wp(a, b, current_iter=i)
                assert set(got) == set(row["pair"][k]) and all(abs(float(got[n]) - v) <= 1e-7 for n, v in row["pair"][k].items())
    for kw, msg in zip((dict(loss_weight=1.0, start_iter=10, target_iter=5), dict(loss_weight=1.0, target_weight=-0.5)), fix["rejected"]):
        with pytest.raises(ValueError) as e:
            create_iterative_loss(Scalar(), kw)
        assert str(e.value) == msg
    with pytest.raises(ValueError, match="current_iter must be provided"):
        create_iterative_loss(Scalar(), dict(loss_weight=1.0))(a, b)
    with pytest.warns(UserWarning, match="very small effective weight"):
        create_iterative_loss(Scalar(), dict(loss_weight=1.0, start_iter=5)).step_weight(0)

    wrapped = build_loss({"type": "l1loss", "loss_weight": 0.5, "start_iter": 4, "target_iter": 8, "target_weight": 1.0, "loss_decay": 0.1})
    assert isinstance(wrapped, IterativeLossWrapper) and type(wrapped.loss_module).__name__ == "L1Loss"
    assert wrapped.loss_weight == 0.5 and wrapped.get_current_weight(6) == 0.75 and wrapped.get_current_weight(3) == 0.0
    assert not isinstance(build_loss({"type": "l1loss", "loss_weight": 0.5}), IterativeLossWrapper)


def test_library_binds_to_the_hip_runtime_torch_already_mapped(built):
    """`_lib.lib()` in a fresh interpreter that has NOT imported torch yet (what `__graft_entry__.build()` does before `smoke()` in one process): the
    process must end up with ONE libamdhip64 -- torch's bundled copy -- or every launch fails with "no ROCm-capable device is detected"."""
    code = ("import sys; sys.path.insert(0, %r); import trainner_redux_amd._lib as L; L.lib(); import torch\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}); print(libs); assert len(libs) == 1, libs" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
