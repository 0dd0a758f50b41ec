"""The arithmetic claim behind the opt-in SAGE_GEMM_X3 mode (graphsage_amd/csrc/linear.hip,
split_bf16x3), checked on the CPU with torch's round-to-nearest bfloat16 conversion: an fp32 value
is EXACTLY the sum of three bf16 terms hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid),
and the six cross products the kernels accumulate reproduce a * b to better than fp32 rounding."""
import torch


def _split(x):
    hi = x.to(torch.bfloat16).float()
    r = x - hi                      # exact in fp32
    mid = r.to(torch.bfloat16).float()
    q = r - mid                     # exact in fp32
    lo = q.to(torch.bfloat16).float()
    return hi, mid, lo


def _values(n, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, generator=g) * torch.exp2(torch.randint(-20, 20, (n,), generator=g).float())
    edge = torch.tensor([0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.5, 3.0e38, -1.17549435e-38 * 2 ** 30,
                         0.1, 1.0 / 3.0, 16777215.0, 1.0 + 2.0 ** -8 + 2.0 ** -16 + 2.0 ** -23])
    return torch.cat([x, edge])


def test_three_bf16_terms_reconstruct_fp32_exactly():
    x = _values(1_000_000, 0)
    hi, mid, lo = _split(x)
    assert torch.equal(hi.double() + mid.double() + lo.double(), x.double())
    # each term really is a bf16 value (8 significand bits): converting again changes nothing
    for t in (hi, mid, lo):
        assert torch.equal(t.to(torch.bfloat16).float(), t)
    # sizes of the terms relative to x: what makes the three dropped products negligible
    nz = x != 0
    assert float((mid[nz] / x[nz]).abs().max()) <= 2.0 ** -8
    assert float((lo[nz] / x[nz]).abs().max()) <= 2.0 ** -16


def test_six_products_are_within_fp32_rounding_of_the_product():
    a, b = _values(500_000, 1)[:500_000], _values(500_000, 2)[:500_000]
    ah, am, al = (t.double() for t in _split(a))
    bh, bm, bl = (t.double() for t in _split(b))
    six = al * bh + ah * bl + am * bm + am * bh + ah * bm + ah * bh      # the kernels' order: smallest first
    exact = a.double() * b.double()
    nz = exact != 0
    rel = ((six - exact)[nz] / exact[nz]).abs().max()
    assert float(rel) <= 2.0 ** -24, float(rel)       # dropped: mid*lo, lo*mid, lo*lo
