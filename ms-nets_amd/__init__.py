"""ms-nets_amd -- MI355X-native cost-volume forward pass of MS-Nets (ccj5351/MS-Nets).

Only the hot path lives here: matchers + likelihood features + volume assembly,
the GCNet / PSMNet 3D-conv aggregators and the soft-argmin tails, all as hand-written HIP kernels for
gfx950 behind the C ABI in include/msnet_hip.h.  The directory name carries a hyphen, so import it as

    import importlib; msnets = importlib.import_module("ms-nets_amd")      # or:  import msnets_amd

Sub-modules mirror the reference's surfaces:
    gcnet_3dcnn.GCNet_CostVolumeAggre, psmnet_3dcnn.PSMNet_CostVolumeAggre   (src/models/*)
    libmatchers, libfeatextract                                             (src/cpp/lib/*)
    cbmv_generator.get_costs / extract_features_left / build_ms_volume       (src/dataloader/cbmv_generator.py)
    dist.shard_indices / gather_disparities                                    (replaces nn.DataParallel)
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
