"""Stop rules of the RRI driver (mirror of the pieces of the reference's optimization module that
nmf() uses: /root/reference/src/rri_nmf/optimization.py:284-297).

`qf_min` -- the closed-form minimiser of w.x + 0.5 x.diag(c).x under x >= 0 (+ sum / upper bound),
optimization.py:12-88 -- has no host implementation here: it is the fused epilogue of the device
kernels k_trow_numer / k_trow_final (T rows) and k_wcol (W columns); see csrc/rri_kernels.hpp.
"""
import numpy as np

eps_div_by_zero = np.spacing(10)             # optimization.py:5
constraint_violation_tolerance = 1e-13       # optimization.py:6


def universal_stopping_condition(obj_history, eps_stop=1e-4):
    """Stop when the last objective change is <= eps_stop * the first one (optimization.py:284-291)."""
    if len(obj_history) < 2:
        return False
    return abs(obj_history[-1] - obj_history[-2]) <= eps_stop * abs(obj_history[0] - obj_history[1])


def first_last_stopping_condition(obj_history, eps_stop=1e-4):
    """optimization.py:294-297"""
    if len(obj_history) < 2:
        return False
    return obj_history[-1] <= obj_history[0] * eps_stop
