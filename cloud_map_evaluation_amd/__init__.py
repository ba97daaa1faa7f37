"""cloud_map_evaluation_amd — MI355X-native engine for MapEval's metric hot path."""
__version__ = "0.1.0"
