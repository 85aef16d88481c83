"""Synthetic periodic boxes of the sizes BASELINE.json names."""
from .synthetic import lj_fluid, system_from_arrays, tip3p_box  # noqa: F401
