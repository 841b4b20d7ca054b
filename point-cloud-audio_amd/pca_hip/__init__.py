"""pca_hip: MI355X (gfx950) kernels for the point-cloud-audio hot path, bound over the C ABI
of libpca_hip.so (include/pca_hip.h)."""
from ._lib import LIB_PATH, PcaHipError, lib  # noqa: F401
from .ops import (MAXK, RANDK, cross_entropy, get_mode, importance_kernel,  # noqa: F401
                  importance_points, linear, mab, mab_infer, pack_points_2d, pack_points_2d_seq, pack_points_2d_ss, pack_points_3d,
                  pack_points_3d_seq, resample, set_mode, stft_logmag, stft_logmag_batch,
                  subsample_points)
