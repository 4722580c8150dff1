"""CPU: the index algebra shared by the HIP kernels (gan-ode_amd/csrc/conv_geom.h: stride-phase tables, packed
weight map, gather/scatter formulas) evaluated on the host (tests/hostcheck/geom_check.cpp, built here with g++)
against torch's convolutions, for every layer shape of the MNIST and UCF networks at tiny widths plus edge cases
(odd extents, negative padding, k<stride phases, strided first-layer input, permuted latent columns)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import REPO

import gan_ode_amd._lib as L
from gan_ode_amd.engine import conv_out, make_geom

SRC = os.path.join(REPO, "tests", "hostcheck", "geom_check.cpp")
SO = os.path.join(REPO, "tests", "hostcheck", "libgeomcheck.so")


@pytest.fixture(scope="module")
def hc():
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(
            os.path.join(REPO, "gan-ode_amd", "csrc", "conv_geom.h"))):
        # UBSan in trap mode (no runtime library needed in a ctypes-loaded .so): signed overflow, bad shifts, OOB on fixed arrays
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-fsanitize=undefined,bounds",
                               "-fsanitize-undefined-trap-on-error", SRC, "-o", SO])
    lib = C.CDLL(SO)
    lib.hc_igemm.argtypes = [C.c_void_p] * 7
    lib.hc_wgrad.argtypes = [C.c_void_p] * 5
    return lib


def cl(t):      # [N,C,D,H,W] -> channels-last contiguous [N,D,H,W,C]
    return t.permute(0, 2, 3, 4, 1).contiguous()


def uncl(t):    # [N,D,H,W,C] -> [N,C,D,H,W]
    return t.permute(0, 4, 1, 2, 3)


CASES = [
    # (Ci, Co, (Di,Hi,Wi), k, s, p)
    (1, 8, (16, 28, 28), (2, 2, 2), (1, 2, 2), (0, 1, 1)),     # MNIST video-D layer 0
    (8, 16, (15, 15, 15), (2, 2, 2), (1, 2, 2), (0, 1, 1)),    # odd extents, conv drops the last row
    (16, 4, (13, 5, 5), (2, 2, 2), (1, 2, 2), (0, 1, 1)),
    (8, 1, (12, 3, 3), (2, 2, 2), (1, 1, 1), (0, 0, 0)),       # last video-D layer
    (3, 8, (6, 16, 16), (4, 4, 4), (1, 2, 2), (0, 1, 1)),      # UCF video-D layer 0 (shrunk)
    (4, 8, (1, 8, 8), (1, 4, 4), (1, 2, 2), (0, 1, 1)),        # decoder stride-2 layers / image-D
    (1, 8, (1, 28, 28), (1, 4, 4), (1, 2, 2), (0, 1, 1)),
    (12, 1, (1, 3, 3), (1, 4, 4), (1, 2, 2), (0, 1, 1)),       # image-D last layer 3x3 -> 1x1
    (1, 8, (1, 28, 28), (1, 1, 1), (1, 1, 1), (0, 2, 2)),      # MNIST decoder last layer: k1 p2 = crop
    (8, 12, (1, 4, 4), (1, 4, 4), (1, 1, 1), (0, 0, 0)),       # decoder layer 0 (FULLK)
    (5, 7, (3, 7, 9), (3, 2, 2), (1, 3, 2), (1, 1, 0)),        # k < s in one dim (a phase with no tap), 6 phases
]


def _geom(case, N=2):
    Ci, Co, xi, k, s, p = case
    yo = tuple(conv_out(xi[a], k[a], s[a], p[a]) for a in range(3))
    return make_geom(N, Ci, Co, xi, yo, k, s, p), yo


@pytest.mark.parametrize("case", CASES)
def test_fprop_dgrad_wgrad_maps(hc, case):
    Ci, Co, xi, k, s, p = case
    g, yo = _geom(case)
    gen = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(2, Ci, *xi, generator=gen)
    w = torch.randn(Co, Ci, *k, generator=gen)
    zeros = (C.c_int64 * 5)()
    neg_pad = any(v < 0 for v in p) or case[5] == (0, 2, 2) and k == (1, 1, 1)
    if case[5] == (0, 2, 2) and k == (1, 1, 1):
        y_ref = F.conv3d(x, w, stride=s, padding=p)
    else:
        y_ref = F.conv3d(x, w, stride=s, padding=p)
    assert tuple(y_ref.shape[2:]) == yo
    # FPROP
    xc = cl(x)
    out = torch.full((2, *yo, Co), float("nan"))
    assert hc.hc_igemm(C.byref(g), L.FPROP, xc.data_ptr(), zeros, w.data_ptr(), None, out.data_ptr()) == 0
    assert torch.allclose(uncl(out), y_ref, rtol=1e-4, atol=1e-4)
    # FPROP reading the caller's NCDHW tensor in place through strides
    st = (C.c_int64 * 5)(x.stride(0), x.stride(2), x.stride(3), x.stride(4), x.stride(1))
    out2 = torch.full((2, *yo, Co), float("nan"))
    assert hc.hc_igemm(C.byref(g), L.FPROP, x.data_ptr(), st, w.data_ptr(), None, out2.data_ptr()) == 0
    assert torch.allclose(out2, out, rtol=1e-5, atol=1e-5)
    # DGRAD == conv_transpose (every x position must be written, including ones no y reaches)
    gy = torch.randn(2, Co, *yo, generator=gen)
    opad = tuple(xi[a] - ((yo[a] - 1) * s[a] - 2 * p[a] + k[a]) for a in range(3))
    gx_ref = F.conv_transpose3d(gy, w, stride=s, padding=p, output_padding=opad)
    assert tuple(gx_ref.shape[2:]) == xi
    gx = torch.full((2, *xi, Ci), float("nan"))
    assert hc.hc_igemm(C.byref(g), L.DGRAD, cl(gy).data_ptr(), zeros, w.data_ptr(), None, gx.data_ptr()) == 0
    assert torch.allclose(uncl(gx), gx_ref, rtol=1e-4, atol=1e-4)
    # WGRAD
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv3d(xr, wr, stride=s, padding=p).backward(gy)
    dw = torch.full_like(w, float("nan"))
    assert hc.hc_wgrad(C.byref(g), xc.data_ptr(), cl(gy).data_ptr(), None, dw.data_ptr()) == 0
    assert torch.allclose(dw, wr.grad, rtol=1e-4, atol=1e-3)


def test_latent_column_permutation(hc):
    """Generator layer 0 with the internal latent layout [motion 16 | content 50 | pad 6]: packing with co_perm must
    reproduce ConvTranspose2d(66, C, 4, 1, 0) applied to the reference's [content | motion] rows."""
    Cn = 8
    g = make_geom(3, Cn, 72, (1, 4, 4), (1, 1, 1), (1, 4, 4), (1, 1, 1), (0, 0, 0))
    gen = torch.Generator().manual_seed(3)
    w = torch.randn(66, Cn, 4, 4, generator=gen)              # ConvTranspose2d weight [in, out, kh, kw]
    z_ref = torch.randn(3, 66, generator=gen)                 # [content 50 | motion 16]
    y_ref = F.conv_transpose2d(z_ref.view(3, 66, 1, 1), w)    # [3, Cn, 4, 4]
    z_int = torch.zeros(3, 72)
    z_int[:, :16] = z_ref[:, 50:]
    z_int[:, 16:66] = z_ref[:, :50]
    z_int[:, 66:] = 123.0                                     # pad columns must be ignored (zero weights)
    perm = torch.tensor([50 + i for i in range(16)] + list(range(50)) + [-1] * 6, dtype=torch.int32)
    out = torch.full((3, 1, 4, 4, Cn), float("nan"))
    zeros = (C.c_int64 * 5)()
    w5 = w.view(66, Cn, 1, 4, 4).contiguous()
    assert hc.hc_igemm(C.byref(g), L.DGRAD, z_int.data_ptr(), zeros, w5.data_ptr(), perm.data_ptr(), out.data_ptr()) == 0
    assert torch.allclose(out[:, 0].permute(0, 3, 1, 2), y_ref, rtol=1e-4, atol=1e-4)
    # its input gradient (FPROP of the same geometry) lands in internal column order, pads get zero
    gy = torch.randn(3, 1, 4, 4, Cn, generator=gen)
    gz = torch.full((3, 1, 1, 1, 72), float("nan"))
    assert hc.hc_igemm(C.byref(g), L.FPROP, gy.data_ptr(), zeros, w5.data_ptr(), perm.data_ptr(), gz.data_ptr()) == 0
    zr = z_ref.clone().requires_grad_(True)
    F.conv_transpose2d(zr.view(3, 66, 1, 1), w).backward(gy[:, 0].permute(0, 3, 1, 2))
    gz = gz.view(3, 72)
    assert torch.allclose(gz[:, :16], zr.grad[:, 50:], rtol=1e-4, atol=1e-4)
    assert torch.allclose(gz[:, 16:66], zr.grad[:, :50], rtol=1e-4, atol=1e-4)
    assert float(gz[:, 66:].abs().max()) == 0.0
    # and the weight gradient scatters back to canonical rows
    wr = w.clone().requires_grad_(True)
    F.conv_transpose2d(z_ref.view(3, 66, 1, 1), wr).backward(gy[:, 0].permute(0, 3, 1, 2))
    dw = torch.zeros(66, Cn, 1, 4, 4)
    assert hc.hc_wgrad(C.byref(g), gy.data_ptr(), z_int.view(3, 1, 1, 1, 72).data_ptr(), perm.data_ptr(), dw.data_ptr()) == 0
    assert torch.allclose(dw.view(66, Cn, 4, 4), wr.grad, rtol=1e-4, atol=1e-4)
