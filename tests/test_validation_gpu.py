"""Validation path: device PSNR / SSIM against the float64 restatement of the reference's metrics (oracle/metrics.py),
tiled inference against the restated infer_tiled, and the validation loop's bookkeeping.  Parity unpinned (cv2 absent)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("crop,ychan", [(0, False), (4, False), (4, True)])
def test_psnr_ssim_match_float64_restatement(cuda, crop, ychan):
    from oracle import metrics as OM
    from trainner_redux_amd.metrics import calculate_metric, calculate_psnr, calculate_ssim

    g = torch.Generator().manual_seed(crop + 3)
    gt = torch.rand(1, 3, 40, 52, generator=g)
    sr = (gt + 0.05 * torch.randn(gt.shape, generator=g)).clamp(-0.1, 1.1)          # out-of-range values exercise the clamp
    a, b = sr.to(cuda), gt.to(cuda)
    p, s = calculate_psnr(a, b, crop, ychan), calculate_ssim(a, b, crop, ychan)
    rp, rs = OM.psnr(sr, gt, crop, ychan), OM.ssim(sr, gt, crop, ychan)
    assert abs(p - rp) < 1e-6 * abs(rp), (p, rp)
    assert abs(s - rs) < 1e-6, (s, rs)
    assert calculate_psnr(b, b, crop) == float("inf") and abs(calculate_ssim(b, b, crop) - 1.0) < 1e-12
    assert calculate_metric({"img": a, "img2": b}, {"type": "calculate_psnr", "crop_border": crop, "test_y_channel": ychan, "better": "higher"}) == p


def test_tiled_inference_and_validation_loop(cuda, tmp_path):
    from oracle import esrgan as OE, metrics as OM
    from tests.test_sr_model_gpu import _opt
    from trainner_redux_amd.models import SRModel

    opt = _opt([dict(type="l1loss", loss_weight=1.0)], blocks=1)
    model = SRModel(opt)
    hp = OE.esrgan_hparams(scale=4, num_blocks=1)
    sd = OE.init_state_dict(hp, 0)
    model.net_g.load_state_dict(sd)
    model.net_g_ema.ema_model.load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    lq = torch.rand(1, 3, 40, 56, generator=g)
    gt = torch.rand(1, 3, 160, 224, generator=g)
    ref_net = lambda t: OE.forward(sd, t, hp)  # noqa: E731
    with torch.no_grad():
        whole = ref_net(lq)
        tiled = OM.infer_tiled(ref_net, lq, 24, 8, 4)
    from trainner_redux_amd.utils.options import ValOptions

    opt.val = ValOptions(dict(val_enabled=True, save_img=False, tile_size=0, tile_overlap=0, metrics_enabled=True,
                              metrics=dict(psnr=dict(type="calculate_psnr", crop_border=4, test_y_channel=False),
                                           ssim=dict(type="calculate_ssim", crop_border=4, test_y_channel=True, better="higher"))))
    model.feed_data({"lq": lq, "gt": gt})
    model.test()
    assert ((model.output.cpu() - whole).norm() / whole.norm()).item() < 1e-5
    opt.val.tile_size, opt.val.tile_overlap = 24, 8
    model.test()
    e = ((model.output.cpu() - tiled).norm() / tiled.norm()).item()
    assert model.output.shape == (1, 3, 160, 224) and e < 1e-5, e

    class DS(torch.utils.data.Dataset):
        opt = {"name": "toy"}

        def __len__(self):
            return 2

        def __getitem__(self, i):
            return {"lq": lq[0] * (1 - 0.1 * i), "gt": gt[0], "lq_path": f"{i}.png", "gt_path": f"{i}.png"}

    loader = torch.utils.data.DataLoader(DS(), batch_size=1)
    model.validation(loader, 100)
    exp = 0.0
    with torch.no_grad():
        for i in range(2):
            exp += OM.psnr(OM.infer_tiled(ref_net, lq * (1 - 0.1 * i), 24, 8, 4), gt, 4) / 2
    assert abs(model.metric_results["psnr"] - exp) < 1e-4 * abs(exp)
    assert model.best_metric_results["toy"]["psnr"] == {"better": "higher", "val": model.metric_results["psnr"], "iter": 100}
    model.validation(loader, 200)
    assert model.best_metric_results["toy"]["ssim"]["iter"] == 200         # ties update (>=), as in the reference

    # image dumps (sr_model.py:1053-1113): <visualization>/<dataset> - <name>/<name>_<iter>.png while training with several val sets,
    # <visualization>/<dataset>/<name>[_suffix].png when testing; the pixels are tensor2img's round(clamp(x) * 255)
    import numpy as np
    from PIL import Image

    vis = tmp_path / "visualization"
    opt.path.visualization = str(vis)
    model.validation(loader, 300, save_img=True, multi_val_datasets=True)
    f = vis / "toy - 1" / "1_000300.png"
    assert f.exists() and (vis / "toy - 0" / "0_000300.png").exists()
    want = (model.output[0].float().clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0).cpu().numpy()      # the loader's last item
    assert np.array_equal(np.asarray(Image.open(f)), want)
    model.opt.is_train, opt.val.suffix = False, "x4"
    try:
        model.validation(loader, 300, save_img=True)
    finally:
        model.opt.is_train = True
    assert np.array_equal(np.asarray(Image.open(vis / "toy" / "1_x4.png")), want)
