"""Import shim (fixture generation only)."""
__version__ = "6.0.0-stub"
