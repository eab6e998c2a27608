"""velocity_amd: MI355X-native KLT + NLS hot path of ultralytics/velocity."""
__version__ = "0.1.0"
