"""Layer stacks shared by the tests: the canonical reference architectures of dlwp_amd.presets."""
from dlwp_amd.presets import CF, cnn2_layers, lstm_unet_layers, unet_layers  # noqa: F401
