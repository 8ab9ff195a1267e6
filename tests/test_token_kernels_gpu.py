"""Token-path kernels of the SwinIR block against PyTorch fp32 references of the same ops (oracle/swinir.py pieces)."""
import pytest
import torch
from torch.nn import functional as F

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("c,cp", [(180, 192), (60, 64), (240, 256)])
def test_layernorm_forward_backward(cuda, c, cp):
    from trainner_redux_amd import ops

    g = torch.Generator().manual_seed(c)
    ntok = 8 * 24 * 32
    x = torch.randn(ntok, c, generator=g) * 2 + 0.5
    gam, bet = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    xr, gr, br = x.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    y = F.layer_norm(xr, (c,), gr, br, 1e-5)
    gy = torch.randn(y.shape, generator=g)
    res = torch.randn(ntok, c, generator=g)
    y.backward(gy)
    xd = torch.zeros(ntok, cp, device=cuda)
    xd[:, :c] = x.to(cuda)
    gd, bd = gam.to(cuda), bet.to(cuda)
    for dt in (torch.float32, torch.bfloat16):
        yd = torch.full((ntok, cp), 7.0, dtype=dt, device=cuda)
        mean, rstd = torch.empty(ntok, device=cuda), torch.empty(ntok, device=cuda)
        ops.layernorm(ops.view(xd, 0, c), ops.view(yd), ntok, c, gd, bd, 1e-5, mean, rstd)
        assert rel_l2(yd[:, :c].float().cpu(), y.detach()) < (2e-6 if dt == torch.float32 else 3e-3)
        assert yd[:, c:].abs().max().item() == 0
    gyd = torch.zeros(ntok, cp, device=cuda)
    gyd[:, :c] = gy.to(cuda)
    resd = torch.zeros(ntok, cp, device=cuda)
    resd[:, :c] = res.to(cuda)
    dx = torch.zeros(ntok, cp, device=cuda)
    dgam, dbet = torch.full((c,), 2.0, device=cuda), torch.full((c,), 2.0, device=cuda)
    ws = torch.empty(ops.layernorm_bwd_workspace_bytes(c) // 4, device=cuda)
    ops.layernorm_bwd(ops.view(gyd, 0, c), ops.view(xd, 0, c), ntok, c, gd, mean, rstd, ops.view(resd, 0, c), ops.view(dx, 0, c), dgam, dbet, ws)
    assert rel_l2(dx[:, :c].cpu(), xr.grad + res) < 3e-6
    assert rel_l2(dgam.cpu() - 2.0, gr.grad) < 1e-5 and rel_l2(dbet.cpu() - 2.0, br.grad) < 1e-5


@pytest.mark.parametrize("c,nh,shift", [(180, 6, 0), (180, 6, 4), (60, 6, 4), (64, 2, 0)])
def test_window_attention_forward_backward(cuda, c, nh, shift):
    """softmax(q*scale k^T + table[rel_index] + shift mask) v with roll / partition / reverse as index arithmetic, against the
    explicit roll + window_partition formulation of drct_arch.py:175-223,532-568 (restated in oracle/swinir.py)."""
    from oracle import swinir as OS
    from trainner_redux_amd import ops

    b, h, w, ws = 2, 16, 24, 8
    hd = c // nh
    g = torch.Generator().manual_seed(c + shift)
    qkv = torch.randn(b, h, w, 3 * c, generator=g)
    table = 0.5 * torch.randn((2 * ws - 1) ** 2, nh, generator=g)
    qr, tr_ = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)

    def ref(qkv_t, tab):
        t = torch.roll(qkv_t, (-shift, -shift), (1, 2)) if shift else qkv_t
        win = t.view(b, h // ws, ws, w // ws, ws, 3 * c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, 3, nh, hd).permute(2, 0, 3, 1, 4)
        q, k, v = win[0] * hd ** -0.5, win[1], win[2]
        attn = q @ k.transpose(-2, -1) + tab[OS.rel_index(ws).view(-1)].view(ws * ws, ws * ws, nh).permute(2, 0, 1).unsqueeze(0)
        if shift:
            m = OS.shift_mask(h, w, ws, shift)
            attn = (attn.view(b, m.shape[0], nh, 64, 64) + m.unsqueeze(1).unsqueeze(0)).view(-1, nh, 64, 64)
        o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(-1, 64, c)
        o = o.view(b, h // ws, w // ws, ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, c)
        return torch.roll(o, (shift, shift), (1, 2)) if shift else o

    o = ref(qr, tr_)
    go = torch.randn(o.shape, generator=g)
    o.backward(go)
    cp3, cp = (3 * c + 31) // 32 * 32, (c + 31) // 32 * 32
    qd = torch.zeros(b, h, w, cp3, device=cuda)
    qd[..., : 3 * c] = qkv.to(cuda)
    td = table.to(cuda)
    od = torch.zeros(b, h, w, cp, device=cuda)
    ops.window_attention(ops.view(qd), ops.view(od), td, b=b, h=h, w=w, c=c, nh=nh, shift=shift)
    assert rel_l2(od[..., :c].cpu(), o.detach()) < 3e-6
    god = torch.zeros(b, h, w, cp, device=cuda)
    god[..., :c] = go.to(cuda)
    dq = torch.zeros(b, h, w, cp3, device=cuda)
    dtab = torch.full(table.shape, 1.0, device=cuda)
    ws_ = torch.empty(ops.window_attention_bwd_workspace_bytes(b, h, w, nh) // 4, device=cuda)
    ops.window_attention_bwd(ops.view(qd), ops.view(god), ops.view(dq), td, dtab, ws_, b=b, h=h, w=w, c=c, nh=nh, shift=shift)
    assert rel_l2(dq[..., : 3 * c].cpu(), qr.grad) < 5e-6
    assert rel_l2(dtab.cpu() - 1.0, tr_.grad) < 1e-5


@pytest.mark.parametrize("c,nh,shift,dims", [(180, 6, 0, (2, 16, 24)), (180, 6, 4, (2, 16, 24)), (60, 6, 4, (1, 24, 16)), (64, 2, 0, (1, 8, 8)), (240, 8, 4, (3, 16, 16))])
def test_window_attention_mfma_bf16(cuda, c, nh, shift, dims):
    """bf16 views select the MFMA kernels (attn_mfma.hip): q, k, v, P, dO, dS are bf16 operands, everything else fp32 -- compared with
    the same formulation under the oracle's `bf16_ops` matmul policy.  Outputs are stored in bf16 (2^-9 relative rounding)."""
    from oracle import swinir as OS
    from oracle.precision import BF16_OPS as P
    from trainner_redux_amd import ops

    b, h, w = dims
    ws = 8
    hd = c // nh
    bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    g = torch.Generator().manual_seed(c + shift)
    qkv = bf(torch.randn(b, h, w, 3 * c, generator=g))
    table = 0.5 * torch.randn((2 * ws - 1) ** 2, nh, generator=g)
    qr, tr_ = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)

    def ref(qkv_t, tab):
        t = torch.roll(qkv_t, (-shift, -shift), (1, 2)) if shift else qkv_t
        win = t.view(b, h // ws, ws, w // ws, ws, 3 * c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, 3, nh, hd).permute(2, 0, 3, 1, 4)
        q, k, v = win[0], win[1], win[2]
        attn = P.matmul(q, k.transpose(-2, -1)) * hd ** -0.5 + tab[OS.rel_index(ws).view(-1)].view(ws * ws, ws * ws, nh).permute(2, 0, 1).unsqueeze(0)
        if shift:
            m = OS.shift_mask(h, w, ws, shift)
            attn = (attn.view(b, m.shape[0], nh, 64, 64) + m.unsqueeze(1).unsqueeze(0)).view(-1, nh, 64, 64)
        o = P.matmul(attn.softmax(-1), v).transpose(1, 2).reshape(-1, 64, c)
        o = o.view(b, h // ws, w // ws, ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, c)
        return torch.roll(o, (shift, shift), (1, 2)) if shift else o

    o = ref(qr, tr_)
    go = bf(torch.randn(o.shape, generator=g))
    o.backward(go)
    cp3, cp = (3 * c + 31) // 32 * 32, (c + 31) // 32 * 32
    qd = torch.zeros(b, h, w, cp3, device=cuda, dtype=torch.bfloat16)
    qd[..., : 3 * c] = qkv.to(cuda)
    qd[..., 3 * c:] = 9.0                      # row padding must never leak into the last head
    td = table.to(cuda)
    od = torch.zeros(b, h, w, cp, device=cuda, dtype=torch.bfloat16)
    ops.window_attention(ops.view(qd), ops.view(od), td, b=b, h=h, w=w, c=c, nh=nh, shift=shift)
    e_o = rel_l2(od[..., :c].float().cpu(), o.detach())
    assert od[..., c:].abs().max().item() == 0 if cp > c else True
    god = torch.zeros(b, h, w, cp, device=cuda, dtype=torch.bfloat16)
    god[..., :c] = go.to(cuda)
    dq = torch.zeros(b, h, w, cp3, device=cuda, dtype=torch.bfloat16)
    dtab = torch.full(table.shape, 1.0, device=cuda)
    ws_ = torch.empty(ops.window_attention_bwd_workspace_bytes(b, h, w, nh) // 4, device=cuda)
    ops.window_attention_bwd(ops.view(qd), ops.view(god), ops.view(dq), td, dtab, ws_, b=b, h=h, w=w, c=c, nh=nh, shift=shift)
    e_q = rel_l2(dq[..., : 3 * c].float().cpu(), qr.grad)
    e_t = rel_l2(dtab.cpu() - 1.0, tr_.grad)
    print(f"mfma attention c={c} nh={nh} shift={shift}: out {e_o:.2e} dqkv {e_q:.2e} dtable {e_t:.2e}")
    assert e_o < 4e-3 and e_q < 4e-3 and e_t < 2e-3
    assert dq[..., 3 * c:].abs().max().item() == 0 if cp3 > 3 * c else True


def test_pixel_shuffle_and_affine_output(cuda):
    from trainner_redux_amd import ops

    g = torch.Generator().manual_seed(3)
    for c, r in ((64, 2), (3, 4)):
        x = torch.randn(2, c * r * r, 6, 10, generator=g)
        ref = F.pixel_shuffle(x, r)
        xd = x.permute(0, 2, 3, 1).contiguous().to(cuda)
        cpo = (c + 3) // 4 * 4
        out = torch.zeros(2, 6 * r, 10 * r, cpo, device=cuda)
        ops.pixel_shuffle(ops.view(xd), ops.view(out), 2, 6, 10, c, r)
        assert torch.equal(out[..., :c].permute(0, 3, 1, 2).cpu(), ref)
        back = torch.zeros_like(xd)
        ops.pixel_shuffle(ops.view(out), ops.view(back), 2, 6, 10, c, r, inverse=True)
        assert torch.equal(back, xd)
    img = torch.zeros(2, 3, 24, 40, device=cuda)
    ops.nhwc_to_nchw_affine(ops.view(out), img, [2.0, 2.0, 2.0], [0.1, 0.2, 0.3])
    assert rel_l2(img.cpu(), ref * 2 + torch.tensor([0.1, 0.2, 0.3]).view(1, 3, 1, 1)) < 1e-6
