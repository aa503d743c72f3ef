"""The error types of the path, with the names, attributes and message layout that code written against ``dspeed.errors`` relies on
(reference src/dspeed/errors.py:4-40; the chain fills ``wf_range`` / ``processor`` in, processing_chain.py:1154-1159)."""
from __future__ import annotations


class DSPError(Exception):
    """Root of everything a processor or a chain raises on purpose."""


class DSPFatal(DSPError):
    """A configuration-level failure inside a processor: it stops the run (a failure of one waveform is a NaN result instead).

    Whoever catches it on the way up may attach ``wf_range`` (the rows being processed, a ``range``) and ``processor`` (the
    processor call as text); the message then carries one more line for each.
    """

    wf_range = None
    processor = None

    def __str__(self) -> str:
        lines = [Exception.__str__(self)]
        if self.wf_range:
            lines.append(f"Thrown while processing entries {self.wf_range}")
        if self.processor:
            lines.append(f"Thrown by {self.processor}")
        return "\n".join(lines)


class ProcessingChainError(DSPError):
    """A chain could not be set up (unknown variable, shape or unit mismatch, unsupported recipe syntax)."""
