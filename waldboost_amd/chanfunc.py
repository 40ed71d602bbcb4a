"""Channel functions the native layer has kernels for (include/waldboost_hip.h: WB_CHN_*)."""
import numpy as np

from . import _native as nat


class ChannelSpec:
    """What the native layer needs to know about one channel function."""

    def __init__(self, key, func_id, n_channels, dtype, reference_name):
        self.key, self.func_id, self.n_channels = key, func_id, n_channels
        self.dtype = np.dtype(dtype)
        self.reference_name = reference_name        # module.qualname the reference writes into .pb files
        self.func = None

    @property
    def wb_dtype(self):
        return nat.WB_DTYPE_U8 if self.dtype == np.uint8 else nat.WB_DTYPE_F32

    def __repr__(self):
        return f"ChannelSpec({self.key})"


SPECS = {
    "grad_hist": ChannelSpec("grad_hist", nat.WB_CHN_GRAD_HIST, 4, np.float32, "waldboost.channels.grad_hist"),
    "grad_hist_4_u1": ChannelSpec("grad_hist_4_u1", nat.WB_CHN_GRAD_HIST_4_U1, 4, np.uint8,
                                  "waldboost.fpga.channels.grad_hist_4_u1"),
    "grad_mag_u1": ChannelSpec("grad_mag_u1", nat.WB_CHN_GRAD_MAG_U1, 1, np.uint8,
                               "waldboost.fpga.channels.grad_mag_u1"),
    "grad_mag": ChannelSpec("grad_mag", nat.WB_CHN_GRAD_MAG, 1, np.float32, "waldboost.channels.grad_mag"),
}
