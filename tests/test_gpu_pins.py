"""-m gpu: the HIP path against the pins of tests/pins.py -- expected values that come from the MATLAB lines by exact
rational arithmetic, closed forms and fixed-point arguments, NOT from either restatement under oracle/.  Same bar as the
parity tests: <= 1e-5 relative Frobenius on W / H, cost <= 1e-6 relative."""
import numpy as np
import pytest

import pins

pytestmark = pytest.mark.gpu
TOL, CTOL = 1e-5, 1e-6


def test_hip_hand_derived_kats(gpu_lib):
    pins.pin_kat1(gpu_lib, TOL, CTOL)
    pins.pin_w_step(gpu_lib, TOL)
    pins.pin_h_step(gpu_lib, TOL, CTOL)
    pins.pin_two_sources(gpu_lib, TOL)
    pins.pin_cnmf_kat(gpu_lib, TOL, CTOL)


def test_hip_nmf_fixed_points_generic_path(gpu_lib):
    # euclidean cost at a fixed point is quadratic in the fp32 rounding of V_hat (1e-7^2 of sum V^2), KL / IS linear
    pins.pin_nmf_fixed_point(gpu_lib, TOL, 1e-12)
    pins.pin_nmf_fixed_point(gpu_lib, TOL, 1e-12, shapes=((70, 90, 5),), extra_cfg=dict(nmfx_path=1))


@pytest.mark.parametrize("shape", [(256, 384, 64), (129, 200, 32), (512, 640, 256), (130, 257, 12)])
def test_hip_nmf_fixed_points_fused_kernels(gpu_lib, shape):
    pins.pin_nmf_fixed_point(gpu_lib, TOL, 1e-12, shapes=(shape,), extra_cfg=dict(nmfx_path=2), divs=("euclidean", "kl"))


def test_hip_cnmf_fixed_points(gpu_lib):
    pins.pin_cnmf_fixed_point(gpu_lib, TOL, 1e-12)
    pins.pin_cnmf_fixed_point(gpu_lib, TOL, 1e-12, shapes=((128, 256, 16, 4), (96, 300, 7, 5)))


def test_hip_projfunc_closed_forms(gpu_lib):
    pins.pin_projfunc(gpu_lib, 1e-12)       # float64 in, float64 arithmetic, float64 out


# ---- nmfsc / cnmfsc / lnmf / constrainednmf (tests/pins_sc.py): hand-derived cases, exact rationals, 50-digit decimals with the closed
# form of the Hoyer projection -- none of it from oracle/ ---------------------------------------------------------------------------
import pins_sc


def test_hip_nmfsc_underflow_return(gpu_lib):
    pins_sc.pin_nmfsc_underflow(gpu_lib, TOL, CTOL)                               # 665 rejected tries, cost trimmed to cost(1:1)


def test_hip_nmfsc_mu_branches(gpu_lib):
    pins_sc.pin_nmfsc_mu(gpu_lib, TOL, CTOL)


def test_hip_nmfsc_line_searches(gpu_lib):
    pins_sc.pin_nmfsc_linesearch(gpu_lib, TOL, CTOL)                              # tries [2], step 0.6, both searches over two iterations


@pytest.mark.parametrize("path", [1, 2])
def test_hip_nmfsc_pins_on_named_kernel_paths(gpu_lib, path):
    """path 1: materialised V_hat on the general GEMM; path 2: the fused MFMA kernels (K padded to 32, masked edges) -- the default for
    problems this small is the float64 VALU path, which the three tests above cover"""
    cfg = dict(nmfx_path=path)
    pins_sc.pin_nmfsc_mu(gpu_lib, TOL, CTOL, extra_cfg=cfg)
    pins_sc.pin_nmfsc_linesearch(gpu_lib, TOL, CTOL, extra_cfg=cfg)
    pins_sc.pin_nmfsc_underflow(gpu_lib, TOL, CTOL, extra_cfg=cfg)


def test_hip_cnmfsc_pins(gpu_lib):
    pins_sc.pin_cnmfsc(gpu_lib, TOL, CTOL)                                        # pos + eps, shift-less line search, early return, MU branches


def test_hip_lnmf_pins(gpu_lib):
    pins_sc.pin_lnmf(gpu_lib, TOL, CTOL)                                          # incl. `<=` stop at equal costs, untrimmed cost vector


def test_hip_constrainednmf_pins(gpu_lib):
    pins_sc.pin_constrainednmf(gpu_lib, TOL, CTOL)
