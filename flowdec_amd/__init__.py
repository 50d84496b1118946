"""flowdec_amd -- MI355X-native (gfx950) inference path for the FlowDec postfilter.

Public surface mirrors the reference (`FlowModel.enhance / forward`, `NCSNpp.forward`, checkpoint
state_dict layout); compute lives in libflowdec_hip.so (C ABI: include/flowdec_hip.h).
"""
from .model import (AmplitudeCompressedComplexSTFT, FlowModel, NCSNpp, OUVESDE, PRESETS, RegressionModel, ScoreModel,  # noqa: F401
                    from_preset, sigma_y_from_file)

__all__ = ["FlowModel", "ScoreModel", "RegressionModel", "OUVESDE", "NCSNpp", "AmplitudeCompressedComplexSTFT", "from_preset", "PRESETS",
           "sigma_y_from_file"]
