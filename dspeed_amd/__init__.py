"""dspeed_amd -- MI355X-native engine for dspeed's waveform-DSP hot path.

Public surface (mirrors the slice of ``dspeed`` that the hot path needs):

* ``dspeed_amd.processors``        -- processor registry, same names as ``dspeed.processors``
* ``dspeed_amd.build_processing_chain`` / ``ProcessingChain`` -- JSON recipe -> fused device chain
* ``dspeed_amd.build_dsp``         -- the table loop around it (channels, per-channel recipes and database, row selection, output file)
* ``dspeed_amd.errors``            -- ``DSPFatal`` / ``ProcessingChainError``
* ``dspeed_amd.device``            -- device arrays, streams, events

The compute path is the HIP library ``libdspeed_hip.so`` (``python -m dspeed_amd.build``); nothing here
falls back to the CPU.
"""
from .errors import DSPError, DSPFatal, ProcessingChainError  # noqa: F401
from .build_dsp import build_dsp  # noqa: F401,E402  (function and submodule share the name, as in the reference: the function wins)

__version__ = "0.1.0"


def __getattr__(name):
    if name in ("build_processing_chain", "ProcessingChain", "GroupedProcessingChain", "WaveformInput"):
        from . import processing_chain

        return getattr(processing_chain, name)
    raise AttributeError(name)
