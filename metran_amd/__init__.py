"""metran_amd -- MI355X-native batched Kalman filter / smoother for Metran's
dynamic-factor model (hot path of pastas/metran)."""
__version__ = "0.1.0"
