"""tts_amd: MI355X-native Griffin-Lim vocoder (mel -> waveform), a drop-in for the
``mel_to_audio_griffin_lim`` function of the reference's vocoder/mel_audio.py."""
from .vocoder import (GriffinLimPlan, HaloChain, audio_to_mel_db, get_plan, griffin_lim_batch, mel_to_audio,
                      mel_to_audio_griffin_lim)

__all__ = ["GriffinLimPlan", "HaloChain", "audio_to_mel_db", "get_plan", "griffin_lim_batch", "mel_to_audio",
           "mel_to_audio_griffin_lim"]
__version__ = "0.1.0"
