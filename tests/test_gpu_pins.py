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
