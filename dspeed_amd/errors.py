"""Error types mirroring the reference (src/dspeed/errors.py:4-40): same names, same attributes, same
string formatting, so code written against ``dspeed.errors`` keeps working."""
from __future__ import annotations


class DSPError(Exception):
    """Base class for signal processors."""


class DSPFatal(DSPError):
    """Fatal error thrown by DSP processors that halts production.

    ``wf_range`` (range of waveform indices) and ``processor`` (processor + arguments string) are set by the
    chain after the exception is caught and are appended to the message, as in the reference
    (errors.py:10-34, processing_chain.py:1154-1159).
    """

    def __init__(self, *args) -> None:
        super().__init__(*args)
        self.wf_range = None
        self.processor = None

    def __str__(self) -> str:
        suffix = ""
        if self.wf_range:
            suffix += "\nThrown while processing entries " + str(self.wf_range)
        if self.processor:
            suffix += "\nThrown by " + self.processor
        return super().__str__() + suffix


class ProcessingChainError(DSPError):
    """Error thrown when there is a problem setting up a processing chain."""
