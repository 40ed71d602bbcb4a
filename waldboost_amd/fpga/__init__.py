"""Integer channel functions of the reference's FPGA flavour -- drop-in for the detection-side
names of ``waldboost.fpga`` (reference fpga/__init__.py:12): ``grad_hist_4_u1`` and
``grad_mag_u1`` as ``channel_opts["channels"]``.  uint8 channels quarter the cascade's HBM
traffic (one dword per pixel instead of a float4).  ``waldboost.fpga.DTree`` / ``train`` are
training code and outside this build.
"""
from ..channels import grad_hist_4_u1, grad_mag_u1

__all__ = ["grad_hist_4_u1", "grad_mag_u1"]
