"""pympc_amd: MI355X-native batched solver for pympc's pnnls hot path."""
__version__ = "0.1.0"
