"""wide_deep_amd -- MI355X-native Wide&Deep CTR train-step hot path."""
__all__ = ["capi", "plan", "engine", "synth"]
