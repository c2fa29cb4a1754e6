"""adaface_dev_amd: MI355X-native (gfx950) denoising hot path of AdaFace.

Only the SD-1.5 U-Net epsilon-prediction path and the pieces on either side of
it live here.  The arithmetic runs in hand-written HIP
kernels behind the C-ABI library ``csrc/libadaface_hip.so`` (declared in
``include/adaface_hip.h``); the Python in this package is the host-side mirror
of the reference's module interface (``ldm.modules.attention``,
``ldm.modules.diffusionmodules.openaimodel``, ``ldm.models.diffusion.ddim``).

The directory is named ``adaface-dev_amd``; import it as ``adaface_dev_amd``
(the repo-root shim ``adaface_dev_amd.py`` maps the name onto this directory).
"""

__version__ = "0.1.0"

SD15_UNET_CONFIG = dict(
    in_channels=4,
    model_channels=320,
    out_channels=4,
    num_res_blocks=2,
    attention_resolutions=[4, 2, 1],
    channel_mult=[1, 2, 4, 4],
    num_heads=8,
    use_spatial_transformer=True,
    transformer_depth=1,
    context_dim=768,
    legacy=False,
)

# Same topology at 1/10 width: used for full-tensor parity fixtures.
TINY_UNET_CONFIG = dict(
    in_channels=4,
    model_channels=32,
    out_channels=4,
    num_res_blocks=2,
    attention_resolutions=[4, 2, 1],
    channel_mult=[1, 2, 4, 4],
    num_heads=8,
    use_spatial_transformer=True,
    transformer_depth=1,
    context_dim=64,
    legacy=False,
)
