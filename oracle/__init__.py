"""CPU oracle for the SRModel.optimize_parameters hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a dependency-free pure-PyTorch (CPU, fp32) restatement of the reference
arithmetic on the hot path named by BASELINE.json `north_star`.  It is the *checker*:
only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it.  Nothing under `trainner-redux_amd/` (the product) imports or falls back to it; the
product fails loudly when the HIP library is missing.

Why PyTorch and not C/numpy: the reference is 100 % Python/PyTorch, its
CPU path *is* stock `torch.nn.functional` in fp32, so the faithful restatement of a
floating-point path is the same ATen ops driven in the same order.

Pinning status:
  * pinned by the reference's own numbers: parameter-count checksums
    (docs/source/benchmarks.md:64,221,42 -> ESRGAN 16,697,987 / 16,703,171 / 5,021,155) and the
    behavioural invariants of tests/test_losses/test_losses.py:114-206 and
    tests/test_archs/test_archs.py:148-268;
  * pinned by outputs of the reference itself run in this container (tests/golden/*.pt made
    by tests/golden/make_golden.py from the reference files that load by path): pixel losses,
    GANLoss, UNetDiscriminatorSN, SPAN, EnlargedSampler;
  * NUMERICALLY UNPINNED ("parity unpinned"): ESRGAN/RRDBNet forward values, VGG19 feature
    values and the ema-pytorch schedule -- their arithmetic lives in spandrel / torchvision /
    ema-pytorch, none of which is present in /root/reference or installable here.  They are
    restated from the in-tree twins cited in each docstring and checked by checksum +
    invariants only.
"""
