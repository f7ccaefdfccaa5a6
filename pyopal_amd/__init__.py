"""MI355X-native implementation of PyOpal's database-search hot path."""
__version__ = "0.1.0"
