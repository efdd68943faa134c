"""Run the reference application UNMODIFIED on top of libpss.so:

    python -m pyspecsdr_amd.run [--fix-classify] [--gpu-decoders] /path/to/PySpecSDR/pyspecsdr.py [its own arguments]

--fix-classify: classify_signal runs (on the GPU) instead of raising the reference's NameError for its missing `welch`
import — a deliberate deviation from the reference's present behaviour, hence opt-in.

--gpu-decoders: `import decoders` resolves to pyspecsdr_amd.decoders as well (decode_morse / decode_aprs with their sample-rate halves on the
GPU and the per-message halves in the library's host code) instead of the application's own decoders.py on top of the GPU band-pass.

The reference reaches its hot path through ONE module name — `from signal_processing import *` (pyspecsdr.py:98; its
decoders.py:3 imports `bandpass_filter` from `signal_processing` again).  `install()` registers the drop-in module under
exactly that name in `sys.modules` before the script starts, so every such import resolves to the GPU-backed
implementation and no file of the reference is edited.  The reference's own decoders.py is left alone: its bookkeeping
runs as it is, on top of the GPU band-pass.
"""
import os
import runpy
import sys


def install():
    """Make `signal_processing` resolve to the drop-in module for the rest of this process."""
    from . import signal_processing
    sys.modules["signal_processing"] = signal_processing
    return signal_processing


def main(argv=None):
    argv = list(sys.argv if argv is None else argv)
    flags = set()
    while len(argv) > 1 and argv[1] in ("--fix-classify", "--gpu-decoders"):
        flags.add(argv.pop(1))
    fix_classify = "--fix-classify" in flags
    if len(argv) < 2:
        print(__doc__)
        return 2
    script = os.path.abspath(argv[1])
    sp = install()
    if fix_classify:
        sp.CLASSIFY_RAISES_NAMEERROR = False
    if "--gpu-decoders" in flags:
        from . import decoders
        sys.modules["decoders"] = decoders
    sys.argv = [script] + argv[2:]
    sys.path.insert(0, os.path.dirname(script))          # what `python script.py` would put first
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
