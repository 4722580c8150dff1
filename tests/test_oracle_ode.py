"""CPU: independent checks of the restated integrator (it has no reference fixture: torchdiffeq is absent, see
oracle/ode_ref.py header, "parity unpinned").  (1) 4th-order convergence to scipy's DOP853 solution;
(2) the continuous adjoint agrees with autograd through the unrolled 3/8 steps up to O(dt^4) and the gap shrinks
at the expected rate; (3) dopri5 reaches its tolerance."""
import numpy as np
import torch
from scipy.integrate import solve_ivp

from oracle import ode_ref
from oracle.mocogan_ref import OdeRhs


def _rhs(seed=0, scale=2.0):
    torch.manual_seed(seed)
    f = OdeRhs(16, 16).double()
    with torch.no_grad():
        for p in f.parameters():
            p.mul_(scale)  # make the dynamics non-trivial over t in [0,1]
    return f


def _scipy_solution(f, y0, t_end):
    def fun(t, y):
        with torch.no_grad():
            return f(None, torch.from_numpy(y)[None])[0].numpy()
    return solve_ivp(fun, (0.0, t_end), y0.numpy(), method="DOP853", rtol=1e-12, atol=1e-14).y[:, -1]


def test_rk4_38_fourth_order_convergence():
    f = _rhs()
    y0 = torch.randn(16, dtype=torch.float64)
    exact = _scipy_solution(f, y0, 1.0)
    errs = []
    for n in (8, 16, 32, 64):
        with torch.no_grad():
            sol = ode_ref.odeint(f, y0[None], torch.linspace(0, 1, n + 1, dtype=torch.float64), method="rk4")
        errs.append(np.abs(sol[-1, 0].numpy() - exact).max())
    rates = [np.log2(errs[i] / errs[i + 1]) for i in range(3)]
    assert all(3.6 < r < 4.5 for r in rates), (errs, rates)


def test_adjoint_matches_unrolled_autograd_to_fourth_order():
    f = _rhs(1)
    y0 = torch.randn(4, 16, dtype=torch.float64)
    gaps = []
    for n in (8, 16, 32):
        t = torch.linspace(0, 1, n + 1, dtype=torch.float64)
        w = torch.randn(n + 1, 4, 16, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
        w[1:-1] = 0  # weight only the end points so that grids are comparable
        a = y0.clone().requires_grad_(True)
        (ode_ref.odeint_adjoint(f, a, t, method="rk4") * w).sum().backward()
        g_adj = [a.grad.clone()] + [p.grad.clone() for p in f.parameters()]
        f.zero_grad()
        b = y0.clone().requires_grad_(True)
        (ode_ref.odeint(f, b, t, method="rk4") * w).sum().backward()
        g_unr = [b.grad.clone()] + [p.grad.clone() for p in f.parameters()]
        f.zero_grad()
        gaps.append(max(float((x - y).abs().max() / y.abs().max()) for x, y in zip(g_adj, g_unr)))
    assert gaps[0] < 1e-3 and gaps[2] < 1e-5, gaps
    assert gaps[0] / gaps[1] > 8 and gaps[1] / gaps[2] > 8, gaps  # ~16x per halving


def test_step_size_option_gives_finer_grid():
    f = _rhs(2).float()
    y0 = torch.randn(3, 16)
    t = torch.linspace(0, 1, 16)
    with torch.no_grad():
        coarse = ode_ref.odeint(f, y0, t, method="rk4")
        fine = ode_ref.odeint(f, y0, t, method="rk4", options={"step_size": 1.0 / 60})
    assert coarse.shape == fine.shape == (16, 3, 16)
    assert 0 < float((coarse - fine).abs().max()) < 1e-3


def test_dopri5_reaches_tolerance():
    f = _rhs(3)
    y0 = torch.randn(16, dtype=torch.float64)
    exact = _scipy_solution(f, y0, 1.0)
    with torch.no_grad():
        sol = ode_ref.odeint(f, y0[None], torch.tensor([0.0, 1.0], dtype=torch.float64))
    assert np.abs(sol[-1, 0].numpy() - exact).max() < 1e-6
