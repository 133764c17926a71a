"""parler_tts_amd — MI355X-native Parler-TTS generation path (drop-in for `parler_tts` on that path)."""
__version__ = "0.1.0"
