"""MI355X-native hot path of nsu-ai-team/conv1d-text-vae."""
__all__ = []
