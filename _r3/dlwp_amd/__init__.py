"""
dlwp_amd -- MI355X-native (gfx950) implementation of the DLWP convolutional forecast-step hot path.

Sub-modules import the HIP library lazily through dlwp_amd._lib; there is no CPU compute path.
"""
__version__ = '0.1.0'
