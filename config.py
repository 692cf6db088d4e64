"""Project-level constants (same two names the reference's config.py exports, used by voicemap_amd.librispeech)."""
import os

LIBRISPEECH_SAMPLING_RATE = 16000  # Hz, every LibriSpeech file
PATH = os.path.dirname(os.path.realpath(__file__))  # repository root: data/, logs/ and models/ live under it
