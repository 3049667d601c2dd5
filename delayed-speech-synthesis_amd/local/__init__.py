"""MI355X-native mirror of the reference's ``local`` package, restricted to what the online synthesis path
(decode_online.py) imports: units, models and the feature transforms of common.  Same class names, argument
meaning and error behaviour as the reference; the compute goes through libdss_hip.so."""
