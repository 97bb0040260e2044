"""MI355X-native hot path of itskalvik/skeleton-action-recognition: HIP kernels behind a C ABI
(csrc/, include/sar_hip.h) + the Python host that mirrors the reference's model interfaces."""
