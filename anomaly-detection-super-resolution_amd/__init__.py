"""srad_amd - MI355X-native engine for the SR forward / scorer hot path of
Benedict3007/anomaly-detection-super-resolution."""
__version__ = "0.1.0"
