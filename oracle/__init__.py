"""CPU oracle for the SRModel.optimize_parameters hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a dependency-free pure-PyTorch (CPU, fp32) restatement of the reference
arithmetic on the hot path named by BASELINE.json `north_star`.  It is the *checker*:
only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it.  Nothing under `trainner-redux_amd/` (the product) imports or falls back to it; the
product fails loudly when the HIP library is missing.

Why PyTorch and not C/numpy: the reference is 100 % Python/PyTorch, its
CPU path *is* stock `torch.nn.functional` in fp32, so the faithful restatement of a
floating-point path is the same ATen ops driven in the same order.

Pinning status, round 3:
  * pinned by the reference's own numbers: parameter-count checksums
    (docs/source/benchmarks.md:64,221,42 -> ESRGAN 16,697,987 / 16,703,171 / 5,021,155; :71,75 SwinIR) and the
    behavioural invariants of tests/test_losses/test_losses.py:114-206 and tests/test_archs/test_archs.py:148-268;
  * pinned by outputs of the reference itself run in this container (tests/golden/*.pt made by
    tests/golden/make_golden*.py from reference files that load by path; nothing was denied):
      - pixel losses, GANLoss, UNetDiscriminatorSN, SPAN, EnlargedSampler, registry, filter2d, resize_pt, kernel synthesis, BT.601 colour;
      - ESRGAN: the reference's ResidualDenseBlock_ / RRDB_ / RRDBUpsample classes (archs/scunet_aaf6aa_arch.py:310-384) composed in
        ESRGAN's layer order -> oracle/esrgan.py forward + every parameter gradient (tests/golden/esrgan_twin.pt, tests/test_twins_cpu.py);
      - SwinIR: the reference's Mlp / window_partition / WindowAttention / SwinTransformerBlock / PatchEmbed / Upsample classes
        (archs/drct_arch.py) alone and composed in SwinIR's order -> oracle/swinir.py (tests/golden/swin_twin.pt);
      - OTF: every codec-free static stage of ParagonOTF (models/paragon_otf_degradations.py:251-575) under seeded RNGs -> oracle/otf.py
        (tests/golden/paragon_otf.pt);
      - calculate_psnr / calculate_psnr_pt (metrics/psnr_ssim.py:13-98) -> oracle/metrics.py (tests/golden/psnr.pt);
  * what the composed fixtures do NOT pin is the glue that lives in the absent spandrel wheel: ESRGAN's layer ORDER around the RRDB
    blocks, SwinIR's mean / img_range handling, RSTB residual and LeakyReLU(0.01) before the upsampler -- those stay restated from
    the published topologies and are pinned by the parameter checksums only;
  * NUMERICALLY UNPINNED ("parity unpinned"): VGG19 feature values (torchvision weights / layer indices), the ema-pytorch schedule,
    MoSR (spandrel; no v1 twin in the tree), SSIM and the demosaicing stage (cv2) -- their arithmetic lives in packages that are
    neither in /root/reference nor installable here.
"""
