"""ps_slm_amd -- MI355X-native TASU alignment hot path (SenseVoiceSmall -> projector -> Qwen2.5) behind the
reference's ``model_factory`` plugin surface.  All device arithmetic goes through libtasu_hip.so (C-ABI in
include/tasu_hip.h); torch is used for device memory, streams and torch.distributed only."""
__version__ = "0.1.0"
