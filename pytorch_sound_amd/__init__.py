"""pytorch_sound_amd - MI355X (gfx950) native hot path behind pytorch_sound's API.

waveform -> framing / Hann / rFFT -> magnitude -> mel -> model forward/backward -> optimizer step,
driven by `Trainer`, behind the reference's `register_model` / `build_model` registry.  Numbers on
the feature path come from libpsnd_hip.so (hand-written HIP, C ABI in include/psnd.h); there is no
CPU or eager fallback for it.
"""
__version__ = '0.1.0'
