"""`import DCN` — the module name the reference's NLSPN code imports
(external_src/NLSPN/src/model/deformconv/functions/modulated_deform_conv_func.py:13).  Backed by libptta_hip."""
from proxytta.dcn import modulated_deform_conv_backward, modulated_deform_conv_forward  # noqa: F401
