"""dss_amd: MI355X-native hot path of cronelab/delayed-speech-synthesis (LPCNet vocoder + HGA extractor).

Python host side above the C-ABI library ``libdss_hip.so`` (see include/dss_hip.h).
"""
